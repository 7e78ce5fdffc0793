"""Parallel training strategies: sync replicas (reduce-scatter / all-gather) and the
asynchronous parameter-server mode with bounded staleness."""
from .comm import Communicator, init_process_group  # noqa: F401
from .sync_replicas import SyncReplicasStrategy  # noqa: F401

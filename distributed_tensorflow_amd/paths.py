"""Data / log path helpers with the call signatures of the ``clusterone`` package used at
/root/reference/example.py:7,84-89,97.

Off the hosted platform those helpers return the local paths; on it they return fixed
mount points (``/logs`` and ``/data/<dataset>/<path>``).  Here the "platform" switch is the
environment: ``DTF_LOGS_ROOT`` / ``DTF_DATA_ROOT`` override the roots (a job launcher can
point every task at a shared directory without code changes); otherwise the local
arguments are used.
"""
from __future__ import annotations

import os


def get_logs_path(root: str = "./logs") -> str:
    """Directory for logs and checkpoints: ``$DTF_LOGS_ROOT`` if set, else ``root``."""
    override = os.environ.get("DTF_LOGS_ROOT")
    path = override if override else root
    return os.path.expanduser(path)


def get_data_path(dataset_name: str = "", local_root: str = "./data", local_repo: str = "", path: str = "") -> str:
    """Dataset location: ``$DTF_DATA_ROOT/<dataset_name>/<path>`` if the override is set,
    else ``<local_root>/<local_repo>/<path>``."""
    override = os.environ.get("DTF_DATA_ROOT")
    if override:
        return os.path.join(override, dataset_name, path)
    return os.path.join(os.path.expanduser(local_root), local_repo, path)

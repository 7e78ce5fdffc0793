"""Failure detection for a ps/worker cluster.

TensorFlow's distributed session notices a lost task when an RPC to it fails (``UnavailableError`` /
``AbortedError``) and ``MonitoredTrainingSession`` (/root/reference/example.py:189-192) then re-creates the
session from the last checkpoint.  Where the data plane is a static RCCL / gloo process group, it cannot
re-admit a restarted member and - on RCCL - blocks inside a collective when a peer disappears.  Loss of a task
is therefore detected out of band and answered by restarting the whole gang from the last checkpoint (the exception:
workers of an elastic asynchronous cluster, see :func:`elastic_workers`):

* every task runs a :class:`FailureDetector` thread that heartbeats to the control plane of the first
  parameter server (the chief in ps-less clusters) and asks it which members have gone silent;
* a task that leaves normally announces it first, so finishing early is not a failure;
* the training thread sees the loss as :class:`ClusterLostError` from :class:`HeartbeatHook` at its next
  ``session.run``; if the training thread is stuck in a transfer with the dead peer, the detector ends the
  process itself after ``abort_after_s`` with exit status ``EXIT_CLUSTER_LOST``;
* ``python -m distributed_tensorflow_amd.launch`` restarts all tasks when one ends abnormally, and the chief of the
  new gang restores the latest checkpoint, so training continues from the saved ``global_step``.
"""
from __future__ import annotations

import os
import sys
import threading
import time
from typing import Callable, List, Optional, Sequence, Tuple

from .control import ControlClient, ControlPlaneError, ControlTimeout
from .session import SessionRunHook

EXIT_CLUSTER_LOST = 75      # EX_TEMPFAIL: "try again later" - the launcher restarts the gang


class ClusterLostError(RuntimeError):
    """A task of the cluster stopped responding; the process group cannot continue and must be restarted."""


def member_name(job: str, index: int) -> str:
    return "%s:%d" % (job, index)


class FailureDetector:
    def __init__(self, address: Tuple[str, int], me: str, peers: Sequence[str], interval_s: float = 0.5,
                 max_age_s: float = 5.0, startup_grace_s: float = 120.0, abort_after_s: Optional[float] = None,
                 on_abort: Optional[Callable[[List[str]], None]] = None, tolerated: Sequence[str] = ()):
        self.address = address
        # members whose silence is not the loss of the cluster: stateless workers of an elastic asynchronous cluster
        # (their variables and optimizer state live on the parameter servers; the launcher restarts them one by one)
        self.tolerated = set(tolerated)
        self._absent: List[str] = []
        self.me = me
        self.peers = [p for p in peers if p != me]
        self.interval_s = interval_s
        self.max_age_s = max_age_s
        self.startup_grace_s = startup_grace_s
        self.abort_after_s = abort_after_s
        self.on_abort = on_abort
        self._lost: List[str] = []
        self._lock = threading.Lock()
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self._client: Optional[ControlClient] = None

    # -- queries (any thread)
    def lost(self) -> List[str]:
        with self._lock:
            return list(self._lost)

    def absent(self) -> List[str]:
        """Tolerated members that are silent right now."""
        with self._lock:
            return list(self._absent)

    def check(self):
        lost = self.lost()
        if lost:
            raise ClusterLostError("lost contact with %s" % ", ".join(lost))

    # -- lifecycle
    def start(self) -> "FailureDetector":
        if self._thread is None:
            self._client = ControlClient(self.address[0], self.address[1], timeout_s=30.0)
            self._client.heartbeat(self.me)
            self._thread = threading.Thread(target=self._loop, name="dtf-heartbeat", daemon=True)
            self._thread.start()
        return self

    def stop(self, clean: bool = True):
        """``clean``: tell the others that this task is leaving on purpose."""
        self._stop.set()
        t, self._thread = self._thread, None
        if t is not None:
            t.join(timeout=5.0)
        c, self._client = self._client, None
        if c is not None:
            if clean:
                try:
                    c.set("left/" + self.me, b"1")
                except ControlPlaneError:
                    pass
            c.close()

    def _silent_members(self) -> List[str]:
        c = self._client
        c.heartbeat(self.me)
        seen = set(c.dead_members(-0.001))                     # every member that ever sent a heartbeat
        silent = set(c.dead_members(self.max_age_s))
        if time.time() - self._t0 > self.startup_grace_s:
            silent |= set(self.peers) - seen                   # never showed up at all
        out = []
        for p in self.peers:
            if p in silent:
                try:
                    c.get("left/" + p, timeout_s=0.0)          # it said goodbye: not a failure
                except ControlTimeout:
                    out.append(p)
        return out

    def _loop(self):
        self._t0 = time.time()
        lost_since = None
        while not self._stop.wait(self.interval_s):
            try:
                lost = self._silent_members()
            except ControlPlaneError:
                if self._stop.is_set():
                    return
                lost = ["control plane at %s:%d" % self.address]
            absent = [m for m in lost if m in self.tolerated]
            lost = [m for m in lost if m not in self.tolerated]
            with self._lock:
                self._lost = lost
                self._absent = absent
            if not lost:
                lost_since = None
                continue
            lost_since = lost_since or time.time()
            if self.abort_after_s is not None and time.time() - lost_since >= self.abort_after_s:
                if self._stop.is_set():
                    return
                sys.stderr.write("ERROR: %s lost contact with %s; ending this task so the cluster can be restarted\n"
                                 % (self.me, ", ".join(lost)))
                sys.stderr.flush()
                if self.on_abort is not None:
                    self.on_abort(lost)
                    return
                os._exit(EXIT_CLUSTER_LOST)


def detector_for_server(server, **kw) -> Optional[FailureDetector]:
    """A detector for the task that owns ``server`` (None for a one-task cluster)."""
    if server is None or server.world_size <= 1:
        return None
    peers = [member_name("ps", j) for j in range(server.num_ps)] + \
            [member_name("worker", i) for i in range(server.num_workers)]
    if elastic_workers(server) and "tolerated" not in kw:
        kw["tolerated"] = [member_name("worker", i) for i in range(server.num_workers)]
    return FailureDetector(server.control_address(0), member_name(server.job_name, server.task_index), peers, **kw)


def elastic_workers(server=None) -> bool:
    """``DTF_ELASTIC_WORKERS=1`` in a cluster with parameter-server tasks: a worker that dies is restarted alone and
    re-admitted by the running cluster (asynchronous mode on the one-sided data plane, where a worker holds no state
    anybody else needs) - what ``MonitoredTrainingSession`` does for a TensorFlow worker whose variables live on the
    ps tasks (/root/reference/example.py:136-139,189-192)."""
    return os.environ.get("DTF_ELASTIC_WORKERS", "0") == "1" and (server is None or server.num_ps > 0)


class HeartbeatHook(SessionRunHook):
    """Keeps this task's heartbeat going for the lifetime of the session and turns the loss of another task into a
    :class:`ClusterLostError` at the next ``run``.  ``MonitoredTrainingSession`` installs it for cluster runs."""

    def __init__(self, server=None, interval_s: Optional[float] = None, max_age_s: Optional[float] = None,
                 abort_after_s: Optional[float] = None):
        env = os.environ
        self.server = server
        self.interval_s = interval_s if interval_s is not None else float(env.get("DTF_HEARTBEAT_SECS", "0.5"))
        self.max_age_s = max_age_s if max_age_s is not None else float(env.get("DTF_HEARTBEAT_MAX_AGE_SECS", "5"))
        self.abort_after_s = abort_after_s if abort_after_s is not None \
            else float(env.get("DTF_HEARTBEAT_ABORT_SECS", "10"))
        self.detector: Optional[FailureDetector] = None

    def after_create_session(self, session, coord):
        if self.detector is None:
            from .cluster import current_server
            self.detector = detector_for_server(self.server or current_server(), interval_s=self.interval_s,
                                                max_age_s=self.max_age_s, abort_after_s=self.abort_after_s)
            if self.detector is not None:
                self.detector.start()
        # a strategy that waits on shared-memory words (the asynchronous mode's doorbells) polls the detector while it
        # waits: the loss of the task it is waiting for surfaces as ClusterLostError instead of a timeout
        strategy = getattr(getattr(session, "engine", None), "strategy", None)
        if self.detector is not None and strategy is not None and hasattr(strategy, "external_health_check"):
            strategy.external_health_check = self.detector.check
            if hasattr(strategy, "external_absent"):
                strategy.external_absent = self.detector.absent

    def before_run(self, run_context):
        if self.detector is not None:
            self.detector.check()
        return None

    def end(self, session):
        self.close(clean=True)

    def close(self, clean: bool = True):
        d, self.detector = self.detector, None
        if d is not None:
            d.stop(clean=clean)

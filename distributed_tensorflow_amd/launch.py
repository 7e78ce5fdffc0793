"""Single-node cluster launcher with gang restart.

    python -m distributed_tensorflow_amd.launch --num_ps 1 --num_workers 8 examples/example.py --log_dir /tmp/run

starts ``num_ps`` parameter-server tasks and ``num_workers`` worker tasks of the given script on this machine, each
with the environment the reference scripts read (``JOB_NAME``, ``TASK_INDEX``, ``PS_HOSTS``, ``WORKER_HOSTS``;
/root/reference/example.py:59-63) and ``LOCAL_RANK`` = the worker's GPU.  When a task ends abnormally (killed,
crashed, or ended by its failure detector with status 75) the remaining tasks are stopped and the whole gang is
started again on fresh ports, up to ``--max_restarts`` times; the chief restores the latest checkpoint from the
script's checkpoint directory, so training continues from the saved ``global_step``.

With ``--elastic_workers`` (or ``DTF_ELASTIC_WORKERS=1``) a *worker* that ends abnormally is started again alone, on
the same address, while the parameter servers and the other workers keep running: in the asynchronous mode on the
one-sided data plane a worker holds no state anybody else needs, and its successor is re-admitted by the running
cluster (``AsyncPSStrategy``) - TensorFlow's behaviour for a worker whose variables live on the ps tasks
(/root/reference/example.py:136-139,189-192).  A lost ps task still restarts the gang.

``--run_dir`` receives one log per task and attempt plus ``pids.json`` (task name -> process id of the current
attempt), rewritten at every (re)start.
"""
from __future__ import annotations

import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Tuple


def free_ports(n: int) -> List[int]:
    socks, ports = [], []
    for _ in range(n):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        socks.append(s)
        ports.append(s.getsockname()[1])
    for s in socks:
        s.close()
    return ports


class Gang:
    """One attempt: the processes of every task of the cluster."""

    def __init__(self, script_argv: List[str], num_ps: int, num_workers: int, attempt: int, run_dir: Optional[str],
                 env: Optional[Dict[str, str]] = None):
        self.procs: Dict[Tuple[str, int], subprocess.Popen] = {}
        self.logs = []
        self.attempt, self.run_dir, self.script_argv = attempt, run_dir, list(script_argv)
        self.envs: Dict[Tuple[str, int], Dict[str, str]] = {}
        self.task_restarts: Dict[str, int] = {}
        ports = free_ports(num_ps + num_workers + 1)
        ps_hosts = ",".join("127.0.0.1:%d" % p for p in ports[:num_ps])
        worker_hosts = ",".join("127.0.0.1:%d" % p for p in ports[num_ps:num_ps + num_workers])
        base = dict(os.environ if env is None else env)
        base.update({"PS_HOSTS": ps_hosts, "WORKER_HOSTS": worker_hosts, "DTF_RENDEZVOUS_PORT": str(ports[-1]),
                     "DTF_RESTART_COUNT": str(attempt), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        tasks = [("ps", j) for j in range(num_ps)] + [("worker", i) for i in range(num_workers)]
        for job, idx in tasks:
            e = dict(base, JOB_NAME=job, TASK_INDEX=str(idx))
            if job == "worker":
                e["LOCAL_RANK"] = str(idx)
            self.envs[(job, idx)] = e
            self._spawn((job, idx), "")
        self._write_pids()

    def _spawn(self, key: Tuple[str, int], suffix: str):
        out = None
        if self.run_dir:
            out = open(os.path.join(self.run_dir, "%s_%d.attempt%d%s.log" % (key[0], key[1], self.attempt, suffix)), "w")
            self.logs.append(out)
        self.procs[key] = subprocess.Popen([sys.executable] + self.script_argv, env=self.envs[key], stdout=out,
                                           stderr=subprocess.STDOUT if out else None)

    def _write_pids(self):
        if self.run_dir:
            tmp = os.path.join(self.run_dir, "pids.json.tmp")
            with open(tmp, "w") as f:
                json.dump({"attempt": self.attempt, "pids": {"%s:%d" % k: p.pid for k, p in self.procs.items()},
                           "task_restarts": self.task_restarts}, f)
            os.replace(tmp, os.path.join(self.run_dir, "pids.json"))

    def restart_task(self, key: Tuple[str, int]):
        """Start one task again in place (same addresses, same attempt): an elastic worker's successor."""
        name = "%s:%d" % key
        n = self.task_restarts[name] = self.task_restarts.get(name, 0) + 1
        self.envs[key]["DTF_TASK_RESTART_COUNT"] = str(n)
        self._spawn(key, ".restart%d" % n)
        self._write_pids()

    def failed(self) -> Optional[Tuple[Tuple[str, int], int]]:
        for k, p in self.procs.items():
            rc = p.poll()
            if rc is not None and rc != 0:
                return k, rc
        return None

    def workers_done(self) -> bool:
        return all(p.poll() == 0 for (job, _), p in self.procs.items() if job == "worker")

    def all_done(self) -> bool:
        return all(p.poll() is not None for p in self.procs.values())

    def stop(self, grace_s: float = 5.0):
        """End exactly the processes this gang started."""
        live = [p for p in self.procs.values() if p.poll() is None]
        for p in live:
            p.send_signal(signal.SIGTERM)
        deadline = time.time() + grace_s
        for p in live:
            try:
                p.wait(timeout=max(0.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for f in self.logs:
            f.close()
        self.logs = []


def run(script_argv: List[str], num_ps: int, num_workers: int, max_restarts: int = 3, run_dir: Optional[str] = None,
        ps_exit_grace_s: float = 30.0, env: Optional[Dict[str, str]] = None, log=print,
        elastic_workers: Optional[bool] = None) -> int:
    if run_dir:
        os.makedirs(run_dir, exist_ok=True)
    env = dict(os.environ if env is None else env)
    if elastic_workers is None:
        elastic_workers = env.get("DTF_ELASTIC_WORKERS", "0") == "1"
    elastic_workers = bool(elastic_workers) and num_ps > 0
    env["DTF_ELASTIC_WORKERS"] = "1" if elastic_workers else "0"
    attempt = 0
    restarts = 0
    while True:
        gang = Gang(script_argv, num_ps, num_workers, attempt, run_dir, env)
        log("[launch] attempt %d: started %d ps + %d workers" % (attempt, num_ps, num_workers))
        failure = None
        done_at = None
        try:
            while True:
                failure = gang.failed()
                if failure is not None and elastic_workers and failure[0][0] == "worker" and restarts < max_restarts:
                    (job, idx), rc = failure
                    restarts += 1
                    log("[launch] task %s:%d ended with %s; restarting that task alone (restart %d of %d)"
                        % (job, idx, "signal %d" % -rc if rc < 0 else "status %d" % rc, restarts, max_restarts))
                    gang.restart_task((job, idx))
                    failure = None
                    continue
                if failure is not None or gang.all_done():
                    break
                if gang.workers_done():
                    # the chief asks the ps tasks to leave join(); give them a moment, then end them
                    done_at = done_at or time.time()
                    if time.time() - done_at > ps_exit_grace_s:
                        break
                time.sleep(0.1)
        finally:
            gang.stop()
        if failure is None:
            log("[launch] all workers finished")
            return 0
        (job, idx), rc = failure
        why = "signal %d" % -rc if rc < 0 else "status %d" % rc
        if restarts >= max_restarts:
            log("[launch] task %s:%d ended with %s; no restarts left" % (job, idx, why))
            return rc if rc > 0 else 1
        attempt += 1
        restarts += 1
        log("[launch] task %s:%d ended with %s; restarting the cluster (restart %d of %d)"
            % (job, idx, why, restarts, max_restarts))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m distributed_tensorflow_amd.launch", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--num_ps", type=int, default=1)
    ap.add_argument("--num_workers", type=int, default=2)
    ap.add_argument("--max_restarts", type=int, default=3)
    ap.add_argument("--run_dir", default=None, help="directory for per-task logs and pids.json")
    ap.add_argument("--elastic_workers", action="store_true",
                    help="restart a failed worker alone and let the running cluster re-admit it (asynchronous mode)")
    ap.add_argument("script", help="training script; everything after it is passed to the script")
    ap.add_argument("script_args", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    return run([a.script] + a.script_args, a.num_ps, a.num_workers, a.max_restarts, a.run_dir,
               elastic_workers=True if a.elastic_workers else None)


if __name__ == "__main__":
    sys.exit(main())

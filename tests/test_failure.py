"""Failure detection and gang restart: heartbeats over the control plane, the session hook that turns a silent task
into ClusterLostError, and the launcher restarting 1 ps + 2 workers after one worker is killed (the counterpart of
TensorFlow's recoverable session for the cluster of /root/reference/example.py:124-141,189-192)."""
import json
import os
import signal
import subprocess
import sys
import time

import pytest

from distributed_tensorflow_amd.control import ControlServer
from distributed_tensorflow_amd.failure import ClusterLostError, FailureDetector
from distributed_tensorflow_amd.saver import BundleReader, latest_checkpoint

from .test_example_cluster import ROOT, SCRIPT, _env


def _wait(pred, timeout=10.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if pred():
            return True
        time.sleep(0.05)
    return False


def test_silent_member_is_reported_and_goodbye_is_not():
    srv = ControlServer("127.0.0.1", 0)
    addr = ("127.0.0.1", srv.port)
    names = ["ps:0", "worker:0", "worker:1"]
    det = {n: FailureDetector(addr, n, names, interval_s=0.05, max_age_s=0.4).start() for n in names}
    try:
        time.sleep(0.6)
        assert all(d.lost() == [] for d in det.values())
        det["worker:1"].stop(clean=True)           # finished its work: nobody minds
        time.sleep(0.8)
        assert det["worker:0"].lost() == [] and det["ps:0"].lost() == []
        det["ps:0"].stop(clean=False)              # vanished without a word
        assert _wait(lambda: det["worker:0"].lost() == ["ps:0"])
        with pytest.raises(ClusterLostError, match="ps:0"):
            det["worker:0"].check()
    finally:
        for d in det.values():
            d.stop(clean=False)
        srv.stop()


def test_detector_ends_a_stuck_task_and_reports_a_dead_control_plane():
    srv = ControlServer("127.0.0.1", 0)
    addr = ("127.0.0.1", srv.port)
    aborted = []
    d = FailureDetector(addr, "worker:0", ["worker:0", "ps:0"], interval_s=0.05, max_age_s=0.3, abort_after_s=0.2,
                        on_abort=aborted.append).start()
    try:
        time.sleep(0.3)
        assert d.lost() == []          # ps:0 has not shown up yet, but the start-up grace period is running
        srv.stop()                     # the task that hosts the control plane is gone
        assert _wait(lambda: bool(aborted))
        assert "control plane" in aborted[0][0]
    finally:
        d.stop(clean=False)


def test_member_that_never_appears_is_lost_after_the_grace_period():
    srv = ControlServer("127.0.0.1", 0)
    d = FailureDetector(("127.0.0.1", srv.port), "worker:0", ["worker:0", "worker:1"], interval_s=0.05, max_age_s=0.3,
                        startup_grace_s=0.3).start()
    try:
        assert _wait(lambda: d.lost() == ["worker:1"], timeout=5.0)
    finally:
        d.stop(clean=False)
        srv.stop()


@pytest.mark.parametrize("victim", ["worker:1", "ps:0"])
def test_gang_restart_after_a_task_is_killed(tmp_path, victim):
    log_dir, run_dir = str(tmp_path / "logs"), str(tmp_path / "run")
    last_step = 6000
    env = _env({"DTF_HEARTBEAT_MAX_AGE_SECS": "2", "DTF_HEARTBEAT_ABORT_SECS": "2", "PYTHONPATH": ROOT})
    launcher = subprocess.Popen(
        [sys.executable, "-m", "distributed_tensorflow_amd.launch", "--num_ps", "1", "--num_workers", "2",
         "--max_restarts", "2", "--run_dir", run_dir, SCRIPT, "--log_dir", log_dir, "--seed", "2",
         "--last_step", str(last_step), "--save_checkpoint_secs", "0.3"],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT)
    try:
        # wait until the first gang has written a checkpoint well into the run, then kill exactly one task: a worker,
        # or the parameter server (which also hosts the control plane the others heartbeat to)
        def progressed():
            ck = latest_checkpoint(log_dir) if os.path.isdir(log_dir) else None
            return ck is not None and int(ck.rsplit("-", 1)[1]) >= 300
        assert _wait(progressed, timeout=180), "no checkpoint from the first attempt"
        assert launcher.poll() is None, launcher.stdout.read()[-3000:]
        with open(os.path.join(run_dir, "pids.json")) as f:
            info = json.load(f)
        assert info["attempt"] == 0
        killed_at = int(latest_checkpoint(log_dir).rsplit("-", 1)[1])
        assert killed_at < last_step, "run finished before the fault could be injected"
        os.kill(info["pids"][victim], signal.SIGKILL)
        out, _ = launcher.communicate(timeout=300)
    finally:
        if launcher.poll() is None:
            launcher.kill()
    assert launcher.returncode == 0, out[-3000:]
    assert "restarting the cluster (restart 1 of 2)" in out and "all workers finished" in out
    with open(os.path.join(run_dir, "pids.json")) as f:
        assert json.load(f)["attempt"] == 1
    chief2 = open(os.path.join(run_dir, "worker_0.attempt1.log")).read()
    assert "Training complete!" in chief2
    restored = [ln for ln in chief2.splitlines() if "Restoring parameters from" in ln]
    assert restored and int(restored[0].rsplit("-", 1)[1]) >= killed_at     # picked up where the first gang stopped
    ck = latest_checkpoint(log_dir)
    step = int(ck.rsplit("-", 1)[1])
    assert last_step <= step <= last_step + 4          # continued to the absolute step budget, not a fresh run
    with BundleReader(ck) as r:
        assert step <= int(r.get_tensor("global_step").item()) <= step + 1


def test_example2_cluster_surfaces_a_lost_worker(tmp_path):
    """The Keras-style script in a cluster: it opens its session with ``master=server.target``, so heartbeats run, and a
    killed worker ends the survivor with ClusterLostError (from the epoch-boundary health check) or through the failure
    detector's abort (status 75) - never a hang."""
    from .test_example_cluster import _ports
    script = os.path.join(ROOT, "examples", "example2.py")
    ps_port, w0, w1, rdv = _ports(4)
    hosts = {"PS_HOSTS": "127.0.0.1:%d" % ps_port, "WORKER_HOSTS": "127.0.0.1:%d,127.0.0.1:%d" % (w0, w1),
             "DTF_RENDEZVOUS_PORT": str(rdv), "DTF_HEARTBEAT_MAX_AGE_SECS": "1.5", "DTF_HEARTBEAT_ABORT_SECS": "3",
             "DTF_HEARTBEAT_SECS": "0.2"}
    args = [sys.executable, script, "--log_dir", str(tmp_path / "logs"), "--epochs", "100000", "--mode", "async", "--seed", "1"]
    procs, logs = {}, {}
    try:
        for job, idx in (("ps", 0), ("worker", 0), ("worker", 1)):
            logs[(job, idx)] = open(str(tmp_path / ("%s%d.log" % (job, idx))), "w+")
            procs[(job, idx)] = subprocess.Popen(args, env=_env(dict(hosts, JOB_NAME=job, TASK_INDEX=str(idx))),
                                                 stdout=logs[(job, idx)], stderr=subprocess.STDOUT)

        def training():
            logs[("worker", 0)].flush()
            logs[("worker", 0)].seek(0)
            return "Epoch 3/" in logs[("worker", 0)].read()
        assert _wait(training, timeout=120), "worker 0 never started training"
        procs[("worker", 1)].kill()
        rc = procs[("worker", 0)].wait(timeout=60)
        logs[("worker", 0)].seek(0)
        text = logs[("worker", 0)].read()
        assert rc != 0
        assert "ClusterLostError" in text or rc == 75, text[-2000:]
    finally:
        for p in procs.values():
            if p.poll() is None:
                p.kill()
        for f in logs.values():
            f.close()


@pytest.mark.parametrize("victim,staleness", [("worker:1", "-1"), ("worker:0", "3")])
def test_elastic_worker_is_restarted_alone_and_readmitted(tmp_path, victim, staleness):
    """``launch --elastic_workers``: asynchronous training with the variables on a ps task (one-sided data plane).  A
    killed worker - the chief included - is started again alone; the ps task and the other worker keep their processes
    and keep training; the successor continues the dead worker's lanes, does not roll the owners back to a checkpoint,
    and the cluster reaches its step budget.  With a staleness bound the dead worker's clock does not hold the
    survivor back."""
    assert "control words: shared memory" in elastic_run(tmp_path, victim, staleness, _env({}))


def elastic_run(tmp_path, victim, staleness, base_env, last_step=9000):
    log_dir, run_dir = str(tmp_path / "logs"), str(tmp_path / "run")
    env = dict(base_env, DTF_HEARTBEAT_MAX_AGE_SECS="1.5", DTF_HEARTBEAT_ABORT_SECS="2", DTF_HEARTBEAT_SECS="0.2",
               PYTHONPATH=ROOT)
    launcher = subprocess.Popen(
        [sys.executable, "-m", "distributed_tensorflow_amd.launch", "--num_ps", "1", "--num_workers", "2",
         "--max_restarts", "2", "--elastic_workers", "--run_dir", run_dir, SCRIPT, "--log_dir", log_dir, "--seed", "2",
         "--last_step", str(last_step), "--save_checkpoint_secs", "0.3", "--max_staleness", staleness],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT)
    try:
        def progressed():
            ck = latest_checkpoint(log_dir) if os.path.isdir(log_dir) else None
            return ck is not None and int(ck.rsplit("-", 1)[1]) >= 300
        assert _wait(progressed, timeout=180), "no checkpoint from the cluster"
        assert launcher.poll() is None, launcher.stdout.read()[-3000:]
        with open(os.path.join(run_dir, "pids.json")) as f:
            before = json.load(f)
        killed_at = int(latest_checkpoint(log_dir).rsplit("-", 1)[1])
        assert killed_at < last_step, "run finished before the fault could be injected"
        os.kill(before["pids"][victim], signal.SIGKILL)
        out, _ = launcher.communicate(timeout=300)
    finally:
        if launcher.poll() is None:
            launcher.kill()
    logs = {f: open(os.path.join(run_dir, f)).read() for f in sorted(os.listdir(run_dir)) if f.endswith(".log")}
    report = out[-2000:] + "\n" + "\n".join("== %s\n%s" % (k, v[-1500:]) for k, v in logs.items())
    assert launcher.returncode == 0, report
    assert "restarting that task alone (restart 1 of 2)" in out and "restarting the cluster" not in out, report
    with open(os.path.join(run_dir, "pids.json")) as f:
        after = json.load(f)
    assert after["attempt"] == 0 and after["task_restarts"] == {victim: 1}
    survivors = [k for k in before["pids"] if k != victim]
    assert all(after["pids"][k] == before["pids"][k] for k in survivors)          # nobody else was restarted
    assert after["pids"][victim] != before["pids"][victim]
    job, idx = victim.split(":")
    successor = logs["%s_%s.attempt0.restart1.log" % (job, idx)]
    other = logs["worker_%d.attempt0.log" % (1 - int(idx))]
    assert "Training complete!" in successor and "Training complete!" in other, report
    assert "ClusterLostError" not in other
    ck = latest_checkpoint(log_dir)
    step = int(ck.rsplit("-", 1)[1])
    assert last_step <= step <= last_step + 6, report     # the shared step went on counting: no rollback, no fresh run
    assert "re-admitted at gradient number" in successor, report
    return successor

"""Several ranks time-sharing the one GPU of the test box: every kernel runs on the GPU, the ranks talk through gloo
(RCCL refuses two ranks on one device).  This rehearses the multi-process paths - the reference's 1 ps + 2 workers
launch and the torchrun launch of bench.py - with the HIP kernels in the loop."""
import json
import os
import signal
import subprocess
import sys

import pytest

from distributed_tensorflow_amd.saver import latest_checkpoint

from .test_example_cluster import SCRIPT, _ports

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WAIT = int(os.environ.get("DTF_TEST_WAIT", "100"))     # per process; a stuck cluster fails with its logs instead of hanging


def _env(extra=None):
    env = dict(os.environ)
    for k in ("JOB_NAME", "TASK_INDEX", "PS_HOSTS", "WORKER_HOSTS", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env["DTF_COMM_BACKEND"] = "gloo"
    env["PYTHONFAULTHANDLER"] = "1"          # a process ended with SIGABRT prints the Python stack of every thread
    env["OMP_NUM_THREADS"] = "2"
    env.update(extra or {})
    return env


@pytest.mark.parametrize("mode", ["async", "sync"])
def test_one_ps_two_workers_on_one_gpu(tmp_path, mode):
    ps_port, w0, w1, rdv = _ports(4)
    hosts = {"PS_HOSTS": "127.0.0.1:%d" % ps_port, "WORKER_HOSTS": "127.0.0.1:%d,127.0.0.1:%d" % (w0, w1),
             "DTF_RENDEZVOUS_PORT": str(rdv)}
    args = [sys.executable, SCRIPT, "--log_dir", str(tmp_path), "--last_step", "200", "--seed", "5"] + \
        (["--sync"] if mode == "sync" else ["--max_staleness", "3"])
    procs, logs = {}, {}
    try:
        for job, idx in (("ps", 0), ("worker", 0), ("worker", 1)):
            logs[(job, idx)] = open(str(tmp_path / ("%s%d.log" % (job, idx))), "w+")
            procs[(job, idx)] = subprocess.Popen(args, env=_env(dict(hosts, JOB_NAME=job, TASK_INDEX=str(idx))),
                                                 stdout=logs[(job, idx)], stderr=subprocess.STDOUT)
        for key in (("worker", 0), ("worker", 1), ("ps", 0)):
            try:
                rc = procs[key].wait(timeout=WAIT)
            except subprocess.TimeoutExpired:
                rc = None
            if rc != 0:
                # stuck or failed: end exactly the processes this test started, with their stacks in the logs
                report = []
                for k2, p2 in procs.items():
                    if p2.poll() is None:
                        p2.send_signal(signal.SIGABRT)
                        try:
                            p2.wait(timeout=20)
                        except subprocess.TimeoutExpired:
                            p2.kill()
                    logs[k2].seek(0)
                    report.append("==== %s%d rc=%s\n%s" % (k2[0], k2[1], p2.poll(), logs[k2].read()[-4000:]))
                raise AssertionError("%s%d rc=%s\n%s" % (key[0], key[1], rc, "\n".join(report)))
            logs[key].seek(0)
            text = logs[key].read()
            if key[0] == "worker":
                assert "Training complete!" in text
    finally:
        for p in procs.values():
            if p.poll() is None:
                p.kill()
        for f in logs.values():
            f.close()
    ck = latest_checkpoint(str(tmp_path))
    assert ck is not None and 200 <= int(ck.rsplit("-", 1)[1]) <= 204


def test_elastic_worker_is_readmitted_on_the_gpu(tmp_path):
    """1 ps + 2 workers on the one GPU, variables in the ps task's hipIpc-mapped region: a killed worker is restarted
    alone, maps the owner's region again and continues its doorbell lane while the others keep training."""
    from .test_failure import elastic_run
    # (a long budget: on the GPU the cluster takes ten thousand steps a second, and the fault has to land mid-run)
    log = elastic_run(tmp_path, "worker:1", "3", _env({"HSA_ENABLE_IPC_MODE_LEGACY": "0"}), last_step=120000)
    from distributed_tensorflow_amd.parallel.doorbell import ControlBlocks
    if ControlBlocks.replica_mode_available("cuda:0"):
        # the owners mapped the successor's new region and brought its replica of the control words up to date
        assert "control words: replicas in device memory" in log


@pytest.mark.parametrize("allreduce", ["rccl", "p2p"])
def test_bench_two_ranks_share_the_gpu(allreduce):
    """bench.py launched exactly as the multi-GPU driver launches it (torch.distributed.run, 2 ranks); "rccl" is the
    default gradient all-reduce (gloo stands in here), "p2p" the one-shot peer-pointer kernel."""
    port = _ports(1)[0]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--batch-per-gpu", "4096", "--allreduce", allreduce]
    out = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=WAIT, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["steps"] == 6 and d["config"]["allreduce"] == allreduce
    assert d["config"]["train_loss"] == d["config"]["train_loss"]      # finite, not NaN


def test_bench_self_launch_two_ranks_share_the_gpu():
    """Plain ``python bench.py --gpus 2`` (no torchrun): the script starts both ranks itself."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--batch-per-gpu", "4096", "--steady", "0.1", "--eval", "10"]
    out = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=WAIT, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["parallelism"] == "dp2" and c["group_ranks"] == 2 and c["self_launched"]
    assert c["rccl_ranks"] is None and c["backend"] == "gloo" and c["launch_variant"] == "forced-gloo"
    assert sorted(r["rank"] for r in c["ranks"]) == [0, 1] and len({r["pid"] for r in c["ranks"]}) == 2
    assert c["test_acc"] is not None and c["test_acc"] > 0.5


def test_bench_more_ranks_than_gpus_over_rccl_is_refused():
    """RCCL needs one GPU per rank: ``--gpus 8`` on the one-GPU box exits non-zero with a message that says why, and
    prints no result."""
    env = _env()
    env.pop("DTF_COMM_BACKEND")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=WAIT, cwd=ROOT)
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this machine really has 8 GPUs")
    assert out.returncode != 0
    assert "exposes" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]

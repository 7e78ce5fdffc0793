"""examples/example.py launched the way the reference is launched: no arguments (single machine), and as a
localhost cluster of 1 ps + 2 workers (CPU, gloo) driven by JOB_NAME / TASK_INDEX / PS_HOSTS / WORKER_HOSTS."""
import glob
import os
import socket
import subprocess
import sys

import pytest

from distributed_tensorflow_amd.saver import BundleReader, latest_checkpoint, list_variables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "examples", "example.py")


def _ports(n):
    socks, ports = [], []
    for _ in range(n):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        socks.append(s)
        ports.append(s.getsockname()[1])
    for s in socks:
        s.close()
    return ports


def _env(extra=None):
    env = dict(os.environ)
    for k in ("JOB_NAME", "TASK_INDEX", "PS_HOSTS", "WORKER_HOSTS"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    env.update(extra or {})
    return env


def test_single_machine(tmp_path):
    out = subprocess.run([sys.executable, SCRIPT, "--log_dir", str(tmp_path), "--last_step", "150", "--seed", "1"],
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Running single-machine training" in out.stdout
    assert "Training complete!" in out.stdout
    ck = latest_checkpoint(str(tmp_path))
    assert ck is not None and ck.endswith("model.ckpt-150")


@pytest.mark.parametrize("mode", ["async", "async-sendrecv", "bounded", "sync"])
def test_one_ps_two_workers(tmp_path, mode):
    ps_port, w0, w1, rdv = _ports(4)
    hosts = {"PS_HOSTS": "127.0.0.1:%d" % ps_port, "WORKER_HOSTS": "127.0.0.1:%d,127.0.0.1:%d" % (w0, w1),
             "DTF_RENDEZVOUS_PORT": str(rdv)}
    extra = {"async": [], "async-sendrecv": [], "bounded": ["--max_staleness", "2"], "sync": ["--sync"]}[mode]
    if mode == "async-sendrecv":
        hosts["DTF_PS_DATA_PLANE"] = "p2p"      # the plane of multi-node clusters (default here: peer regions)
    args = [sys.executable, SCRIPT, "--log_dir", str(tmp_path), "--last_step", "120", "--seed", "5"] + extra
    procs = {}
    logs = {}
    try:
        for job, idx in (("ps", 0), ("worker", 0), ("worker", 1)):
            logs[(job, idx)] = open(str(tmp_path / ("%s%d.log" % (job, idx))), "w+")
            procs[(job, idx)] = subprocess.Popen(args, env=_env(dict(hosts, JOB_NAME=job, TASK_INDEX=str(idx))),
                                                 stdout=logs[(job, idx)], stderr=subprocess.STDOUT)
        for key in (("worker", 0), ("worker", 1), ("ps", 0)):
            try:
                rc = procs[key].wait(timeout=240)
            except subprocess.TimeoutExpired:
                rc = None
            logs[key].seek(0)
            text = logs[key].read()
            assert rc == 0, "%s%d rc=%s\n%s" % (key[0], key[1], rc, text[-3000:])
            if key[0] == "worker":
                assert "Running distributed training" in text and "Training complete!" in text
    finally:
        for p in procs.values():
            if p.poll() is None:
                p.kill()
        for f in logs.values():
            f.close()
    ck = latest_checkpoint(str(tmp_path))
    assert ck is not None
    step = int(ck.rsplit("-", 1)[1])
    # shared global_step: the budget is cluster-wide; each worker may have two gradients in the owners' inbox slots and
    # one more computed against a counter it read before those were applied
    assert 120 <= step <= 126
    names = dict(list_variables(ck))
    assert names["dense_1/kernel"] == (64, 128) and names["global_step"] == ()
    assert "dense_3/kernel/Adam_1" in names
    with BundleReader(ck) as r:
        # asynchronous mode: the other worker's pushes (two inbox slots) may land between the chief reading global_step for the file name
        # and the owner snapshotting its shard (TensorFlow's SaveV2 on a live parameter server has the same window)
        assert step <= int(r.get_tensor("global_step").item()) <= step + 2
        assert r.get_tensor("dense_1/kernel").abs().sum() > 0
    assert glob.glob(os.path.join(str(tmp_path), "events.out.tfevents.*"))


def test_supervisor_style_replica_script(tmp_path):
    """``examples/supervisor_replica.py``: 1 ps + 2 workers of a script written with ``tf.train.Supervisor`` and raw
    ``tf.Variable`` / ``tf.matmul`` ops; shared global step, chief's checkpoint in ``--train_dir`` with the script's own
    variable names."""
    script = os.path.join(ROOT, "examples", "supervisor_replica.py")
    ps_port, w0, w1, rdv = _ports(4)
    args = [sys.executable, script, "--ps_hosts", "127.0.0.1:%d" % ps_port,
            "--worker_hosts", "127.0.0.1:%d,127.0.0.1:%d" % (w0, w1), "--train_dir", str(tmp_path), "--train_steps", "80"]
    procs, logs = {}, {}
    try:
        for job, idx in (("ps", 0), ("worker", 0), ("worker", 1)):
            logs[(job, idx)] = open(str(tmp_path / ("%s%d.log" % (job, idx))), "w+")
            procs[(job, idx)] = subprocess.Popen(args + ["--job_name", job, "--task_index", str(idx)],
                                                 env=_env({"DTF_RENDEZVOUS_PORT": str(rdv)}), stdout=logs[(job, idx)],
                                                 stderr=subprocess.STDOUT)
        texts = {}
        for key in (("worker", 0), ("worker", 1), ("ps", 0)):
            try:
                rc = procs[key].wait(timeout=240)
            except subprocess.TimeoutExpired:
                rc = None
            logs[key].seek(0)
            texts[key] = logs[key].read()
            assert rc == 0, "%s%d rc=%s\n%s" % (key[0], key[1], rc, texts[key][-3000:])
    finally:
        for p in procs.values():
            if p.poll() is None:
                p.kill()
        for f in logs.values():
            f.close()
    for idx in (0, 1):
        assert "Worker %d: session ready" % idx in texts[("worker", idx)] and "done after" in texts[("worker", idx)]
    ck = latest_checkpoint(str(tmp_path))
    assert ck is not None and 80 <= int(ck.rsplit("-", 1)[1]) <= 86
    names = dict(list_variables(ck))
    assert names["hid_w"] == (784, 100) and names["sm_b"] == (10,) and "hid_w/Adam_1" in names

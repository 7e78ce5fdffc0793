"""Resuming an asynchronous parameter-server job restores the OWNERS' optimizer state, not only the variables: after
a restart the owners reload variables, Adam slots, beta powers and the step from the checkpoint files, so a checkpoint
written right after the restore equals the one restored from.  Also: the reference's launch shape at world 8
(1 ps + 8 workers, /root/reference/example.py:124-131) through the single-node launcher, with ps-task owners and with
the ps task reduced to the control plane (owners inside the workers - what an 8-GPU node runs)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from distributed_tensorflow_amd import launch
from distributed_tensorflow_amd.saver import BundleReader, latest_checkpoint, list_variables

from .test_example_cluster import SCRIPT, _env, _ports

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, torch
sys.path.insert(0, %(root)r)
torch.set_num_threads(1)
from distributed_tensorflow_amd import session as ss
from distributed_tensorflow_amd.cluster import ClusterSpec, Server
from distributed_tensorflow_amd.data import mnist_like
from distributed_tensorflow_amd.engine import ModelEngine, TRAIN_OP
from distributed_tensorflow_amd.layers import Dense, Dropout
from distributed_tensorflow_amd.models import Sequential
from distributed_tensorflow_amd.parallel.async_ps import AsyncPSStrategy

index, hosts, ps_hosts, steps, ckdir = int(sys.argv[1]), sys.argv[2].split(","), sys.argv[3], int(sys.argv[4]), sys.argv[5]
spec = {"worker": hosts}
if ps_hosts:
    spec["ps"] = ps_hosts.split(",")
srv = Server(ClusterSpec(spec), "worker", index, config={"device": "cpu"})
strat = AsyncPSStrategy(srv)
m = Sequential([Dense(24, input_dim=784, activation="relu"), Dropout(0.3), Dense(10, activation="softmax")])
m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], strategy=strat, device="cpu",
          seed=10 + index)
eng = ModelEngine(m)
x, y = mnist_like(32 * 64, seed=index)
n = len(hosts)
with ss.MonitoredTrainingSession(master=srv.target, is_chief=srv.is_chief, checkpoint_dir=ckdir, engine=eng,
                                 save_checkpoint_secs=None, save_summaries_steps=None, log_step_count_steps=None,
                                 hooks=[ss.StopAtStepHook(last_step=steps)]) as sess:
    start = sess.global_step_value()
    i = 0
    while not sess.should_stop():
        sess.run(TRAIN_OP, {"x": x[i * 32:(i + 1) * 32], "y": y[i * 32:(i + 1) * 32]})
        i += 1
    strat.worker_done()
    strat.cp0.barrier("done", n, timeout_s=120)
    if srv.is_chief:
        # every worker has stopped pushing: what the owners hold now is final
        prefix = sess.save_checkpoint(os.path.join(ckdir, "model.ckpt"), strat.global_step())
    strat.cp0.barrier("saved", n, timeout_s=120)
if srv.is_chief:
    strat.shutdown_owners()
print("RESULT " + json.dumps({"index": index, "start": start, "ran": i}))
'''


def _run_cluster(tmp_path, script, n, steps, ckdir, ps=False):
    ports = _ports(n + 1)
    hosts = ",".join("127.0.0.1:%d" % p for p in ports[:n])
    ps_hosts = "127.0.0.1:%d" % ports[n] if ps else ""
    procs = [subprocess.Popen([sys.executable, str(script), str(i), hosts, ps_hosts, str(steps), ckdir], env=_env(),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for i in range(n)]
    res = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=300)
            assert p.returncode == 0, out[-3000:]
            res.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return sorted(res, key=lambda r: r["index"])


def test_owners_reload_slots_and_beta_powers_on_resume(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    ckdir = str(tmp_path / "ck")
    first = _run_cluster(tmp_path, script, 2, 12, ckdir)
    assert all(r["start"] == 0 for r in first)
    ck1 = latest_checkpoint(ckdir)
    step1 = int(ck1.rsplit("-", 1)[1])
    assert step1 >= 12
    with BundleReader(ck1) as r:
        saved = {k: r.get_tensor(k) for k in r.entries}
    assert saved["dense_1/kernel/Adam_1"].abs().sum() > 0

    # second job on the same directory whose budget is already spent: restore, train nothing, save again
    second = _run_cluster(tmp_path, script, 2, step1, ckdir)
    assert all(r["start"] == step1 and r["ran"] == 0 for r in second)
    ck2 = latest_checkpoint(ckdir)
    assert ck2 == ck1 or int(ck2.rsplit("-", 1)[1]) == step1
    with BundleReader(ck2) as r:
        again = {k: r.get_tensor(k) for k in r.entries}
    assert set(again) == set(saved)
    for k in saved:                      # the owners' state after the restore IS the checkpoint: slots, powers, step
        assert torch.equal(saved[k], again[k]), k


@pytest.mark.parametrize("colocated", ["0", "1"], ids=["ps_task_owners", "ps_task_control_plane_only"])
def test_launch_one_ps_eight_workers_async(tmp_path, colocated):
    """python -m distributed_tensorflow_amd.launch --num_ps 1 --num_workers 8 examples/example.py (CPU, gloo)."""
    log_dir = str(tmp_path / "logs")
    env = _env({"DTF_COLOCATED_OWNERS": colocated})
    rc = launch.run([SCRIPT, "--log_dir", log_dir, "--last_step", "160", "--seed", "3", "--max_staleness", "2"],
                    num_ps=1, num_workers=8, max_restarts=0, run_dir=str(tmp_path / "run"), env=env, log=lambda *_: None)
    logs = {f: open(os.path.join(str(tmp_path / "run"), f)).read() for f in os.listdir(str(tmp_path / "run")) if f.endswith(".log")}
    assert rc == 0, "\n".join("== %s\n%s" % (k, v[-1500:]) for k, v in logs.items())
    assert sum("Training complete!" in v for v in logs.values()) == 8
    ck = latest_checkpoint(log_dir)
    step = int(ck.rsplit("-", 1)[1])
    # every worker checks the shared step before it starts a step, and may have one more gradient in flight per spare
    # inbox slot (two slots here): the budget is overshot by at most workers * slots - 1
    assert 160 <= step <= 160 + 8 * 2
    names = dict(list_variables(ck))
    assert names["dense_1/kernel"] == (64, 128) and "dense_3/kernel/Adam_1" in names
    with BundleReader(ck) as r:
        assert r.num_shards == (8 if colocated == "1" else 1)

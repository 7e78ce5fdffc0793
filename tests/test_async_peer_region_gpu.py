"""The asynchronous parameter-server data plane on a GPU: workers with colocated owners exchange gradient and
variable shards through hipIpc-mapped peer regions, and doorbell words replace every per-step message - replicas in
every rank's device memory written by kernels where the CPU can address that memory, else (and in the last case here)
shared host memory written by stream host functions.  2 and 4 self-launched ranks time-share the one GPU of the test box; on an
8-GPU node the same code runs with one rank per GPU over xGMI.

Checked: training converges, a step makes no control-plane round trip, the staleness bound holds, the owners hold the
only optimizer state (1/N each), and the checkpoint they write restores bit-exactly into a single process."""
import json
import os
import subprocess
import sys

import pytest
import torch

from distributed_tensorflow_amd.saver import BundleReader

from .test_example_cluster import _ports

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WAIT = int(os.environ.get("DTF_TEST_WAIT", "100"))

WORKER = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
from distributed_tensorflow_amd.cluster import ClusterSpec, Server
from distributed_tensorflow_amd.data import mnist_like
from distributed_tensorflow_amd.engine import ModelEngine
from distributed_tensorflow_amd.layers import Dense, Dropout
from distributed_tensorflow_amd.models import Sequential
from distributed_tensorflow_amd.parallel.async_ps import AsyncPSStrategy

index, hosts, steps, bound, ckpt = int(sys.argv[1]), sys.argv[2].split(","), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
n = len(hosts)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
srv = Server(ClusterSpec({"worker": hosts}), "worker", index, config={"device": "cuda:0"})
strat = AsyncPSStrategy(srv, max_staleness=bound, timeout_s=60)
assert strat.colocated and strat.num_owners == n
m = Sequential([Dense(96, input_dim=784, activation="relu"), Dropout(0.3), Dense(10, activation="softmax")])
m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], strategy=strat, device=dev,
          seed=10 + index)
assert strat.region is not None and strat.region.local.is_cuda           # the hipIpc path, not shared host memory
B = 256
x, y = mnist_like(B * steps, seed=index, device=dev)
calls, lead, losses = [], [], []
for i in range(steps):
    c0 = strat.control_calls()
    m.train_step(x[i * B:(i + 1) * B], y[i * B:(i + 1) * B])
    calls.append(strat.control_calls() - c0)
    lead.append(strat.clock - 1 - strat.min_clock())
    if i in (0, steps - 1):
        losses.append(m.read_stats()["loss"])
    elif i == steps - 2:
        m.stats.zero_()
strat.worker_done()
strat.cp0.barrier("all_done", n, timeout_s=60)
eng = ModelEngine(m)
gs = strat.global_step()
if srv.is_chief:
    eng.save_checkpoint(eng.make_saver(), ckpt, gs)
strat.cp0.barrier("saved", n, timeout_s=60)
strat.pull(m)
torch.cuda.synchronize()
stats = strat.owner_stats() if srv.is_chief else None
report = strat.hbm_report(m)
strat.cp0.barrier("pulled", n, timeout_s=60)
if srv.is_chief:
    strat.shutdown_owners()
print("RESULT " + json.dumps({"index": index, "global_step": gs, "sum": float(m.store.flat.double().sum()),
                              "calls": calls, "lead": lead, "losses": losses, "hist": strat.staleness_hist,
                              "owner_stats": stats, "hbm": report, "depth": strat.depth,
                              "replica_words": strat.blocks.replicas is not None,
                              "kernel": m.store.variables["dense_1/kernel"].tensor.double().sum().item()}))
strat.close()
'''


def _run(tmp_path, n, steps, bound, words="replica"):
    ports = _ports(n)
    hosts = ",".join("127.0.0.1:%d" % p for p in ports)
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    ckpt = str(tmp_path / "ck" / "model.ckpt")
    env = dict(os.environ)
    for k in ("JOB_NAME", "TASK_INDEX", "PS_HOSTS", "WORKER_HOSTS", "RANK", "WORLD_SIZE", "LOCAL_RANK", "DTF_COMM_BACKEND"):
        env.pop(k, None)
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", PYTHONFAULTHANDLER="1",
               DTF_DOORBELL_KERNELS="1" if words == "replica" else "0")
    procs = [subprocess.Popen([sys.executable, str(script), str(i), hosts, str(steps), str(bound), ckpt], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for i in range(n)]
    results = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=WAIT)
            assert p.returncode == 0, out[-3000:]
            line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
            results.append(json.loads(line[len("RESULT "):]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    results.sort(key=lambda r: r["index"])
    return results, ckpt


@pytest.mark.parametrize("n,bound,words", [(2, 2, "replica"), (4, 3, "replica"), (2, 0, "replica"), (3, 2, "shm")])
def test_ranks_sharing_the_gpu_train_through_peer_regions(tmp_path, n, bound, words):
    steps = 30
    results, ckpt = _run(tmp_path, n, steps, bound, words)
    from distributed_tensorflow_amd.parallel.doorbell import ControlBlocks
    assert all(r["replica_words"] == (words == "replica" and ControlBlocks.replica_mode_available("cuda:0")) for r in results)
    assert all(r["global_step"] == n * steps for r in results)
    assert len({round(r["sum"], 5) for r in results}) == 1               # everybody pulled the same variables
    # steady state: no control-plane round trip inside a step
    assert all(c == 0 for r in results for c in r["calls"]), results[0]["calls"]
    # the bound: a worker starts a step only while its clock leads the slowest by at most `bound`
    assert all(l <= bound for r in results for l in r["lead"]), [max(r["lead"]) for r in results]
    assert all(r["losses"][-1] < r["losses"][0] for r in results), [r["losses"] for r in results]
    assert all(r["depth"] == (1 if bound == 0 else 2) for r in results)
    if bound == 0:
        # one inbox slot, no run-ahead: with a single other worker a push sees at most that worker's updates
        assert all(int(k) <= n for r in results for k in r["hist"])
    owners = results[0]["owner_stats"]
    assert len(owners) == n and all(o["version"] == n * steps for o in owners)
    assert sum(sum(o["staleness"].values()) for o in owners) == n * n * steps
    # sharded storage: a worker holds no optimizer slots, an owner exactly 1/N of each
    for r in results:
        h = r["hbm"]
        assert h["worker_slots"] == 0 and h["owner_slots"] * n == 2 * h["worker_variables"]
        assert h["owner_variables"] * n == h["worker_variables"]
    # the owners' checkpoint: whole variables although their elements are spread over owners; bit-exact restore
    prefix = "%s-%d" % (ckpt, n * steps)
    from distributed_tensorflow_amd.layers import Dense, Dropout
    from distributed_tensorflow_amd.models import Sequential
    from distributed_tensorflow_amd.saver import Saver
    m = Sequential([Dense(96, input_dim=784, activation="relu"), Dropout(0.3), Dense(10, activation="softmax")])
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", device="cpu", seed=99)
    assert Saver(m.store, m.optimizer).restore(prefix) == n * steps
    with BundleReader(prefix) as r:
        assert r.num_shards == n
        for v in m.store.trainable:
            assert torch.equal(v.tensor, r.get_tensor(v.name))
            assert torch.equal(m.store.slot_view("m", v), r.get_tensor(v.name + "/Adam"))
            assert torch.equal(m.store.slot_view("v", v), r.get_tensor(v.name + "/Adam_1"))
        assert r.get_tensor("dense_1/kernel/Adam_1").abs().sum() > 0
        assert abs(r.get_tensor("dense_1/kernel").double().sum().item() - results[0]["kernel"]) < 1e-6

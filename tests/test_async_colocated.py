"""Asynchronous training with colocated owners: a cluster of workers only, every worker also owning 1/N of the flat
variable buffer (the layout meant for 8 workers on 8 GPUs), here as 3 CPU processes over gloo."""
import json
import os
import subprocess
import sys

from .test_example_cluster import _env, _ports

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
torch.set_num_threads(1)
from distributed_tensorflow_amd.cluster import ClusterSpec, Server
from distributed_tensorflow_amd.data import mnist_like
from distributed_tensorflow_amd.layers import Dense, Dropout
from distributed_tensorflow_amd.models import Sequential
from distributed_tensorflow_amd.parallel.async_ps import AsyncPSStrategy

index, hosts, steps = int(sys.argv[1]), sys.argv[2].split(","), int(sys.argv[3])
srv = Server(ClusterSpec({"worker": hosts}), "worker", index, config={"device": "cpu"})
strat = AsyncPSStrategy(srv, max_staleness=2)
assert strat.colocated and strat.num_owners == len(hosts) and strat.placement_name == "flat"
m = Sequential([Dense(32, input_dim=784, activation="relu"), Dropout(0.3), Dense(10, activation="softmax")])
m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], strategy=strat, device="cpu",
          seed=10 + index)
first = m.store.flat.clone()
x, y = mnist_like(64 * steps, seed=index)
losses = []
for i in range(steps):
    m.stats.zero_()
    m.train_step(x[i * 64:(i + 1) * 64], y[i * 64:(i + 1) * 64])
    losses.append(m.read_stats()["loss"])
strat.worker_done()
strat.cp0.barrier("all_done", len(hosts), timeout_s=120)
strat.pull(m)
gs = strat.global_step()
stats = strat.owner_stats() if srv.is_chief else None
strat.cp0.barrier("pulled", len(hosts), timeout_s=120)
if srv.is_chief:
    strat.shutdown_owners()
print("RESULT " + json.dumps({"index": index, "global_step": gs, "sum": float(m.store.flat.double().sum()),
                              "first_sum": float(first.double().sum()), "loss0": losses[0], "loss_last": losses[-1],
                              "stale_max": max(strat.staleness_hist), "owner_stats": stats}))
'''


def test_three_workers_colocated_owners(tmp_path):
    n, steps = 3, 12
    ports = _ports(n + 1)
    hosts = ",".join("127.0.0.1:%d" % p for p in ports[:n])
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    procs = [subprocess.Popen([sys.executable, str(script), str(i), hosts, str(steps)],
                              env=_env({"DTF_RENDEZVOUS_PORT": str(ports[n])}), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for i in range(n)]
    results = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=240)
            assert p.returncode == 0, out[-3000:]
            line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
            results.append(json.loads(line[len("RESULT "):]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    results.sort(key=lambda r: r["index"])
    # one shared global_step: every push of every worker counted once
    assert all(r["global_step"] == n * steps for r in results)
    # after the final pull every worker holds the owners' variables: identical everywhere, and trained
    assert len({round(r["sum"], 6) for r in results}) == 1
    assert results[0]["sum"] != results[0]["first_sum"]
    # the chief's initial values were the ones training started from (chief-initialises / others-wait)
    assert all(r["loss_last"] < r["loss0"] for r in results)
    # between a worker's pull at clock c and the apply of the gradient it then computes, any other worker moves from
    # at least c - s to at most c + s + 1 pushes (s = max_staleness = 2); the worker's own previous gradient may
    # still be in flight as well
    assert all(r["stale_max"] <= (n - 1) * (2 * 2 + 1) + 1 for r in results)
    owners = results[0]["owner_stats"]
    assert len(owners) == n and all(o["version"] == n * steps for o in owners)

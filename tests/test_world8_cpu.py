"""World-size-8 rehearsals on CPU: the shapes an 8-GPU node runs (8 flat owners, 8 staleness clocks, 8 replicas
writing one checkpoint), with gloo / shared memory standing in for RCCL / hipIpc.

* asynchronous training, 8 workers with colocated owners, flat sharding, ``max_staleness=2``: no control-plane
  traffic inside a step (round trips per step counted), an owner-side sharded checkpoint whose variables straddle owners, and a restore of
  that checkpoint into a single process, bit-exact, optimizer slots included;
* synchronous replicas, world 4, bucketed: every replica writes its own data shard; restored into a single process.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from distributed_tensorflow_amd.saver import BundleReader, list_variables

from .test_example_cluster import _env, _ports

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ASYNC_WORKER = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
torch.set_num_threads(1)
from distributed_tensorflow_amd.cluster import ClusterSpec, Server
from distributed_tensorflow_amd.data import mnist_like
from distributed_tensorflow_amd.engine import ModelEngine
from distributed_tensorflow_amd.layers import Dense, Dropout
from distributed_tensorflow_amd.models import Sequential
from distributed_tensorflow_amd.parallel.async_ps import AsyncPSStrategy

index, hosts, steps, ckpt = int(sys.argv[1]), sys.argv[2].split(","), int(sys.argv[3]), sys.argv[4]
n = len(hosts)
srv = Server(ClusterSpec({"worker": hosts}), "worker", index, config={"device": "cpu"})
strat = AsyncPSStrategy(srv, max_staleness=2)
assert strat.colocated and strat.num_owners == n and strat.placement_name == "flat"
m = Sequential([Dense(40, input_dim=784, activation="relu"), Dropout(0.3), Dense(10, activation="softmax")])
m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], strategy=strat, device="cpu",
          seed=10 + index)
assert strat.gather          # 784x40 floats over 8 owners: the kernel matrix straddles owners
x, y = mnist_like(32 * steps, seed=index)
calls = []
for i in range(steps):
    c0 = strat.control_calls()
    m.train_step(x[i * 32:(i + 1) * 32], y[i * 32:(i + 1) * 32])
    calls.append(strat.control_calls() - c0)
strat.worker_done()
strat.cp0.barrier("all_done", n, timeout_s=120)
eng = ModelEngine(m)
if srv.is_chief:
    prefix = eng.save_checkpoint(eng.make_saver(), ckpt, strat.global_step())
strat.cp0.barrier("saved", n, timeout_s=120)
strat.pull(m)
gs = strat.global_step()
stats = strat.owner_stats() if srv.is_chief else None
strat.cp0.barrier("pulled", n, timeout_s=120)
if srv.is_chief:
    strat.shutdown_owners()
print("RESULT " + json.dumps({"index": index, "global_step": gs, "sum": float(m.store.flat.double().sum()),
                              "calls": calls, "stale_max": max(strat.staleness_hist), "owner_stats": stats,
                              "hbm": strat.hbm_report(m),
                              "kernel": m.store.variables["dense_1/kernel"].tensor.double().sum().item()}))
'''


def test_eight_async_workers_flat_owners_checkpoint_round_trip(tmp_path):
    n, steps = 8, 6
    ports = _ports(n)
    hosts = ",".join("127.0.0.1:%d" % p for p in ports)
    script = tmp_path / "worker.py"
    script.write_text(ASYNC_WORKER % {"root": ROOT})
    ckpt = str(tmp_path / "ck" / "model.ckpt")
    procs = [subprocess.Popen([sys.executable, str(script), str(i), hosts, str(steps), ckpt], env=_env(),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for i in range(n)]
    results = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=300)
            assert p.returncode == 0, out[-3000:]
            line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
            results.append(json.loads(line[len("RESULT "):]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    results.sort(key=lambda r: r["index"])
    assert all(r["global_step"] == n * steps for r in results)
    assert len({round(r["sum"], 6) for r in results}) == 1           # everybody pulled the same variables
    # a training step makes NO control-plane round trip: gradients, parameters, versions, clocks and the shared step
    # all travel through the peer regions and the doorbell words
    assert all(c == 0 for r in results for c in r["calls"]), results[0]["calls"]
    # between a worker's pull at clock c and the apply of the gradient it then computes, any other worker moves from
    # at least c - s to at most c + s + 1 pushes (s = max_staleness = 2); the worker's own previous gradient may
    # still be in flight as well
    assert all(r["stale_max"] <= (n - 1) * (2 * 2 + 1) + 1 for r in results)
    owners = results[0]["owner_stats"]
    assert len(owners) == n and all(o["version"] == n * steps for o in owners)
    # sharded storage: no optimizer slots on a worker, exactly 1/8 of the variables and of each slot on an owner
    for r in results:
        h = r["hbm"]
        assert h["worker_slots"] == 0
        assert h["owner_variables"] * n == h["worker_variables"] and h["owner_slots"] * n == 2 * h["worker_variables"]

    # the checkpoint: one data shard per owner, TF names, whole variables although their elements were spread over owners
    prefix = "%s-%d" % (ckpt, n * steps)
    names = dict(list_variables(prefix))
    assert names["dense_1/kernel"] == (784, 40) and names["dense_2/bias"] == (10,) and names["global_step"] == ()
    assert {"dense_1/kernel/Adam", "dense_1/kernel/Adam_1", "beta1_power", "beta2_power"} <= set(names)
    with BundleReader(prefix) as r:
        assert r.num_shards == n
        assert len({e["shard"] for e in r.entries.values()}) > 1       # several owners wrote
        kernel = r.get_tensor("dense_1/kernel")
        assert abs(kernel.double().sum().item() - results[0]["kernel"]) < 1e-9     # what the workers pulled afterwards
        assert int(r.get_tensor("global_step").item()) == n * steps
        assert r.get_tensor("dense_1/kernel/Adam_1").abs().sum() > 0
        assert abs(r.get_tensor("beta1_power").item() - 0.9 ** (n * steps + 1)) < 1e-6

    # restore into a single-process model: variables and slots bit-exact
    from distributed_tensorflow_amd.layers import Dense, Dropout
    from distributed_tensorflow_amd.models import Sequential
    from distributed_tensorflow_amd.saver import Saver
    m = Sequential([Dense(40, input_dim=784, activation="relu"), Dropout(0.3), Dense(10, activation="softmax")])
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", device="cpu", seed=99)
    assert Saver(m.store, m.optimizer).restore(prefix) == n * steps
    with BundleReader(prefix) as r:
        for v in m.store.trainable:
            assert torch.equal(v.tensor, r.get_tensor(v.name))
            assert torch.equal(m.store.slot_view("m", v), r.get_tensor(v.name + "/Adam"))
            assert torch.equal(m.store.slot_view("v", v), r.get_tensor(v.name + "/Adam_1"))


SYNC_WORKER = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
torch.set_num_threads(1)
from distributed_tensorflow_amd.data import mnist_like
from distributed_tensorflow_amd.engine import ModelEngine
from distributed_tensorflow_amd.layers import Dense
from distributed_tensorflow_amd.models import Sequential
from distributed_tensorflow_amd.parallel import Communicator, SyncReplicasStrategy, init_process_group

rank, world, port, ckpt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
init_process_group("gloo", rank=rank, world_size=world, master_addr="127.0.0.1", master_port=port)
m = Sequential([Dense(48, input_dim=784, activation="relu"), Dense(48, activation="relu"), Dense(10, activation="softmax")])
strat = SyncReplicasStrategy(Communicator(), bucket_bytes=4096, allreduce_below_bytes=0)        # sharded, several buckets
m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], strategy=strat, device="cpu",
          seed=100 + rank)
assert not strat.replicated and len(strat.buckets) > 1
x, y = mnist_like(32 * world, seed=3)
for _ in range(4):
    m.train_step(x[rank * 32:(rank + 1) * 32], y[rank * 32:(rank + 1) * 32])
eng = ModelEngine(m)
saver = eng.make_saver()
assert saver.num_shards == world
prefix = eng.save_checkpoint(saver, ckpt, 4)            # collective: every replica writes its shard
assert not m.store.slots                                 # no full-length optimizer state anywhere ...
slot_bytes = strat.slot_bytes_per_rank()                 # ... a rank holds 1/world of each slot
full = strat.full_slots(m)                               # (collective, for the comparison below only)
torch.save({"flat": m.store.flat, "m": full["m"], "v": full["v"]}, ckpt + ".rank%%d.pt" %% rank)
print("RESULT " + json.dumps({"rank": rank, "prefix": prefix, "slot_bytes": slot_bytes, "flat_bytes": m.store.flat_len * 4}))
import torch.distributed as dist
dist.barrier()
dist.destroy_process_group()
'''


def test_sync_replicas_write_one_shard_each(tmp_path):
    world = 4
    port = _ports(1)[0]
    script = tmp_path / "worker.py"
    script.write_text(SYNC_WORKER % {"root": ROOT})
    ckpt = str(tmp_path / "ck" / "model.ckpt")
    os.makedirs(os.path.dirname(ckpt))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port), ckpt], env=_env(),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    try:
        for p in procs:
            out, _ = p.communicate(timeout=300)
            assert p.returncode == 0, out[-3000:]
            res = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
            assert res["slot_bytes"] * world == 2 * res["flat_bytes"]        # Adam's m and v: total / world per rank
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    prefix = ckpt + "-4"
    files = sorted(os.listdir(os.path.dirname(ckpt)))
    assert ["model.ckpt-4.data-0000%d-of-00004" % i for i in range(world)] == [f for f in files if ".data-" in f]
    assert not [f for f in files if ".index.part-" in f]                 # merged by the chief
    names = dict(list_variables(prefix))
    assert names["dense_1/kernel"] == (784, 48) and "dense_3/bias/Adam_1" in names and names["global_step"] == ()
    with BundleReader(prefix) as r:
        # the strategy's buckets (two here: the two upper layers, the first layer) are dealt out to the replicas
        assert r.num_shards == world and len({e["shard"] for e in r.entries.values()}) == 2
    # restore into a single process: bit-exact against what the replicas held, slots included
    from distributed_tensorflow_amd.layers import Dense
    from distributed_tensorflow_amd.models import Sequential
    from distributed_tensorflow_amd.saver import Saver
    m = Sequential([Dense(48, input_dim=784, activation="relu"), Dense(48, activation="relu"), Dense(10, activation="softmax")])
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", device="cpu", seed=7)
    assert Saver(m.store, m.optimizer).restore(prefix) == 4
    held = torch.load(ckpt + ".rank0.pt", weights_only=True)
    from distributed_tensorflow_amd.variables import _round_up
    off = 0
    for v in m.store.trainable:
        for name, mine in (("flat", v.tensor), ("m", m.store.slot_view("m", v)), ("v", m.store.slot_view("v", v))):
            assert torch.equal(mine.reshape(-1), held[name][off:off + v.numel]), (v.name, name)
        off += _round_up(v.numel, 64 * world)

"""ps tasks as owners on the one-sided data plane: 2 ps tasks + 2 workers of one node (CPU here: shared-memory regions).
Whole variables are placed round-robin, so an owner holds several flat ranges; workers put / get those ranges straight
into / out of the ps tasks' regions, doorbell words carry versions, acknowledgements, the shared step and the clocks,
and no process group exists at all.  ``DTF_PS_DATA_PLANE=p2p`` keeps the send/recv plane (other nodes / mixed
devices)."""
import json
import os
import subprocess
import sys

import pytest

from distributed_tensorflow_amd.saver import BundleReader

from .test_example_cluster import _env, _ports

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TASK = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
torch.set_num_threads(1)
import torch.distributed as dist
from distributed_tensorflow_amd.cluster import ClusterSpec, Server
from distributed_tensorflow_amd.data import mnist_like
from distributed_tensorflow_amd.layers import Dense, Dropout
from distributed_tensorflow_amd.models import Sequential
from distributed_tensorflow_amd.parallel.async_ps import AsyncPSStrategy

job, index, ps_hosts, w_hosts, steps = sys.argv[1], int(sys.argv[2]), sys.argv[3].split(","), sys.argv[4].split(","), int(sys.argv[5])
import os
device = os.environ.get("DTF_TEST_DEVICE", "cpu")
srv = Server(ClusterSpec({"ps": ps_hosts, "worker": w_hosts}), job, index, config={"device": device})
if job == "ps":
    srv.join()
    sys.exit(0)
strat = AsyncPSStrategy(srv, max_staleness=2)
assert not strat.colocated and strat.num_owners == len(ps_hosts)
reference_net = os.environ.get("DTF_TEST_MODEL") == "xor"
if reference_net:
    # the reference network at its own batch size: on a GPU a worker's whole step is one kernel that reads the ps
    # tasks' memory and writes into their inboxes
    from distributed_tensorflow_amd.data import get_data
    from distributed_tensorflow_amd.models import xor_mlp
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], strategy=strat,
              device=device, seed=10 + index)
    x, y = get_data(64 * steps, val_set_size=0, seed=index)[:2]
else:
    m = Sequential([Dense(32, input_dim=784, activation="relu"), Dropout(0.3), Dense(24, activation="relu"),
                    Dense(10, activation="softmax")])
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], strategy=strat, device=device,
              seed=10 + index)
    x, y = mnist_like(64 * steps, seed=index)
assert not m.store.slots                      # nobody but the owners holds optimizer state
x, y = x.to(device), y.to(device)
losses = []
calls0 = strat.control_calls()
for i in range(steps):
    m.stats.zero_()
    m.train_step(x[i * 64:(i + 1) * 64], y[i * 64:(i + 1) * 64])
    losses.append(m.read_stats()["loss"])
calls = strat.control_calls() - calls0
strat.worker_done()
strat.cp0.barrier("all_done", len(w_hosts), timeout_s=120)
strat.pull(m)
gs = strat.global_step()
stats = strat.owner_stats() if srv.is_chief else None
ck = None
if srv.is_chief:
    import uuid
    strat.save(m, sys.argv[6], uuid.uuid4().hex)
strat.cp0.barrier("pulled", len(w_hosts), timeout_s=120)
if srv.is_chief:
    strat.shutdown_owners()
print("RESULT " + json.dumps({"index": index, "global_step": gs, "sum": float(m.store.flat.double().sum()),
                              "loss0": losses[0], "loss_last": losses[-1], "plane": strat.plane, "calls": calls,
                              "ranges": [len(r) for r in strat.ranges], "process_group": dist.is_initialized(),
                              "direct": any(p is not None for p in strat._direct_plans.values()),
                              "replica_words": strat.blocks is not None and strat.blocks.replicas is not None,
                              "stale_max": max(strat.staleness_hist), "owner_stats": stats,
                              "flat": m.store.flat.tolist() if srv.is_chief else None,
                              "names": [[v.name, v.offset, v.numel] for v in m.store.trainable]}))
'''


@pytest.mark.parametrize("plane", ["region", "p2p"])
def test_two_ps_tasks_two_workers(tmp_path, plane):
    _run(tmp_path, plane, "cpu", _env({}))


@pytest.mark.gpu
def test_two_ps_tasks_two_workers_share_the_gpu(tmp_path):
    """The same cluster with all four tasks on one MI355X: hipIpc-mapped regions, the pump's apply kernels on the ps
    tasks' streams, doorbells rung by kernels behind the workers' copies."""
    env = dict(os.environ)
    for k in ("JOB_NAME", "TASK_INDEX", "PS_HOSTS", "WORKER_HOSTS"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    results = _run(tmp_path, "region", "cuda:0", env)
    assert not any(r["direct"] for r in results)          # not the reference network: push / pull through the link


@pytest.mark.gpu
def test_reference_network_one_kernel_per_worker_step(tmp_path):
    """The reference network at batch 64 with its variables on two ps tasks, everything on one MI355X: a worker's step is
    ONE kernel - variables read from the ps tasks' variable planes, gradients written into the worker's inbox slots there,
    doorbells rung by the same launch - and training behaves as on the general path (shared step, versions, equal
    variables everywhere after the final pull, owner-written checkpoint, falling loss)."""
    env = dict(os.environ)
    for k in ("JOB_NAME", "TASK_INDEX", "PS_HOSTS", "WORKER_HOSTS"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["DTF_TEST_MODEL"] = "xor"
    results = _run(tmp_path, "region", "cuda:0", env, steps=60)
    assert all(r["direct"] for r in results)
    # where the CPU can address the GPU's memory the control words live as replicas in every rank's own device memory
    # and the kernel rang the doorbells itself
    from distributed_tensorflow_amd.parallel.doorbell import ControlBlocks
    assert all(r["replica_words"] == ControlBlocks.replica_mode_available("cuda:0") for r in results)


def _run(tmp_path, plane, device, base_env, steps=12):
    n_ps, n_w = 2, 2
    ports = _ports(n_ps + n_w + 1)
    ps_hosts = ",".join("127.0.0.1:%d" % p for p in ports[:n_ps])
    w_hosts = ",".join("127.0.0.1:%d" % p for p in ports[n_ps:n_ps + n_w])
    script = tmp_path / "task.py"
    script.write_text(TASK % {"root": ROOT})
    prefix = str(tmp_path / "model.ckpt-final")
    env = dict(base_env, DTF_RENDEZVOUS_PORT=str(ports[-1]), DTF_PS_DATA_PLANE="" if plane == "region" else "p2p",
               DTF_TEST_DEVICE=device)
    tasks = [("ps", j) for j in range(n_ps)] + [("worker", i) for i in range(n_w)]
    procs = [subprocess.Popen([sys.executable, str(script), job, str(i), ps_hosts, w_hosts, str(steps), prefix], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for job, i in tasks]
    results, outs = [], []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240)[0])
        report = "\n".join("== %s %d rc=%s\n%s" % (job, i, p.returncode, out[-2500:])
                           for (job, i), p, out in zip(tasks, procs, outs))
        assert all(p.returncode == 0 for p in procs), report
        for (job, i), out in zip(tasks, outs):
            if job == "worker":
                line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
                results.append(json.loads(line[len("RESULT "):]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    results.sort(key=lambda r: r["index"])
    assert all(r["plane"] == plane for r in results)
    assert all(r["global_step"] == n_w * steps for r in results)
    assert len({round(r["sum"], 6) for r in results}) == 1 and all(r["loss_last"] < r["loss0"] for r in results)
    assert results[0]["ranges"] == [3, 3]              # six variables round-robin over two ps tasks: w, b alternate
    owners = results[0]["owner_stats"]
    assert len(owners) == n_ps and all(o["version"] == n_w * steps for o in owners)
    if plane == "region":
        # the step path touches no socket and no process group was ever created
        assert all(r["calls"] == 0 and not r["process_group"] for r in results)
    else:
        assert all(r["calls"] >= 2 * steps and r["process_group"] for r in results)
    # the owners wrote the checkpoint (one data shard each); it holds what the chief pulled afterwards
    import torch
    flat = torch.tensor(results[0]["flat"])
    with BundleReader(prefix) as r:
        assert r.num_shards == n_ps
        for name, off, numel in results[0]["names"]:
            assert torch.equal(r.get_tensor(name).reshape(-1), flat[off:off + numel]), name
            assert r.has_tensor(name + "/Adam") and r.has_tensor(name + "/Adam_1")
        assert int(r.get_tensor("global_step").item()) == n_w * steps
    return results

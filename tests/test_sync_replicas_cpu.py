"""Sync-replica strategy on CPU: world sizes 2 and 8 over gloo must match a single process fed the union batch
(the 64*world alignment of the flat buffer and the bucket split are the same arithmetic an 8-GPU node runs)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bucket_bytes, allreduce_below, out_q):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from distributed_tensorflow_amd.data import mnist_like
    from distributed_tensorflow_amd.layers import Dense, Dropout
    from distributed_tensorflow_amd.models import Sequential
    from distributed_tensorflow_amd.parallel import Communicator, SyncReplicasStrategy, init_process_group
    init_process_group("gloo", rank=rank, world_size=world, master_addr="127.0.0.1", master_port=port)
    m = Sequential([Dense(32, input_dim=784, activation="relu"), Dense(32, activation="relu"),
                    Dense(10, activation="softmax")])
    strat = SyncReplicasStrategy(Communicator(), bucket_bytes=bucket_bytes, allreduce_below_bytes=allreduce_below)
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], strategy=strat,
              device="cpu", seed=100 + rank)  # different seeds: rank 0's init must win
    x, y = mnist_like(64 * world, seed=3)
    xs, ys = x[rank * 64:(rank + 1) * 64], y[rank * 64:(rank + 1) * 64]
    for _ in range(5):
        m.train_step(xs, ys)
    logs = m.read_stats()
    # by value (numpy), not as a shared-memory tensor: the parent may unpickle it after this process has exited
    out_q.put((rank, m.store.flat.clone().numpy(), m.global_step, logs, len(strat.buckets), strat.replicated))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("bucket_bytes,allreduce_below", [(64 << 20, 0), (4096, 0), (64 << 20, 8 << 20)],
                         ids=["one_bucket", "many_buckets", "replicated_allreduce"])
def test_sync_replicas_matches_single_process(bucket_bytes, allreduce_below, world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_bytes, allreduce_below, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    res = [(r[0], torch.from_numpy(r[1])) + tuple(r[2:]) for r in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # replicas agree bit for bit and advanced global_step once per aggregated update
    assert all(torch.equal(res[0][1], r[1]) for r in res[1:])
    assert all(r[2] == 5 for r in res)
    if bucket_bytes == 4096:
        assert res[0][4] > 1
    assert res[0][5] == (allreduce_below > 0) and (not res[0][5] or res[0][4] == 0)

    # single-process reference on the union batch with rank 0's initial values
    sys.path.insert(0, ROOT)
    from distributed_tensorflow_amd.data import mnist_like
    from distributed_tensorflow_amd.layers import Dense
    from distributed_tensorflow_amd.models import Sequential
    m = Sequential([Dense(32, input_dim=784, activation="relu"), Dense(32, activation="relu"),
                    Dense(10, activation="softmax")])
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device="cpu", seed=100)
    x, y = mnist_like(64 * world, seed=3)
    for _ in range(5):
        m.train_step(x, y)
    ref = m.store.flat
    got = res[0][1]
    # layouts differ only by alignment padding: compare variable by variable
    from distributed_tensorflow_amd.variables import _round_up
    off_ref = 0
    off_got = 0
    for v in m.store.trainable:
        a = ref[v.offset:v.offset + v.numel]
        b = got[off_got:off_got + v.numel]
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)
        off_got += _round_up(v.numel, 64 * world)
    logs = res[0][3]
    ref_logs = m.read_stats()
    assert abs(logs["loss"] - ref_logs["loss"]) < 1e-4


def _resume_worker(rank, world, port, ckpt, phase, out_q):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from distributed_tensorflow_amd.data import mnist_like
    from distributed_tensorflow_amd.engine import ModelEngine
    from distributed_tensorflow_amd.layers import Dense
    from distributed_tensorflow_amd.models import Sequential
    from distributed_tensorflow_amd.parallel import Communicator, SyncReplicasStrategy, init_process_group
    init_process_group("gloo", rank=rank, world_size=world, master_addr="127.0.0.1", master_port=port)
    m = Sequential([Dense(32, input_dim=784, activation="relu"), Dense(32, activation="relu"),
                    Dense(10, activation="softmax")])
    strat = SyncReplicasStrategy(Communicator(), bucket_bytes=4096, allreduce_below_bytes=0)       # sharded apply
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", strategy=strat, device="cpu", seed=5 + rank + 10 * phase)
    assert not m.store.slots and strat.slot_bytes_per_rank() * world == 2 * m.store.flat_len * 4
    eng = ModelEngine(m)
    saver = eng.make_saver()
    x, y = mnist_like(64 * world, seed=3)
    xs, ys = x[rank * 64:(rank + 1) * 64], y[rank * 64:(rank + 1) * 64]
    if phase == 0:
        for _ in range(3):
            m.train_step(xs, ys)
        eng.save_checkpoint(saver, ckpt, 3)
    else:
        if rank == 0:                       # the chief restores; everybody else receives / reads its own slot shards
            assert eng.restore_checkpoint(saver, ckpt + "-3") == 3
        eng.synchronize_state(rank == 0)
    for _ in range(3):
        m.train_step(xs, ys)
    slots = strat.full_slots(m)
    out_q.put((rank, m.store.flat.clone().numpy(), slots["v"].clone().numpy(), m.global_step))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_slots_resume_bit_exact(tmp_path):
    """Sharded apply, world 2: a run that is checkpointed after 3 steps and continued equals, bit for bit, a run by
    fresh processes that restore that checkpoint - the variables broadcast from the chief, the optimizer-slot shards
    read by every replica for itself from the files the bucket writers produced."""
    world = 2
    ckpt = str(tmp_path / "model.ckpt")
    finals = []
    for phase in (0, 1):
        port = _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_resume_worker, args=(r, world, port, ckpt, phase, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert all(r[3] == 6 for r in res)
        finals.append(res[0])
        assert all((res[0][1] == r[1]).all() for r in res[1:])
    assert (finals[0][1] == finals[1][1]).all()          # variables
    assert (finals[0][2] == finals[1][2]).all() and float(abs(finals[0][2]).sum()) > 0       # Adam's v


def _out_of_step_worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from distributed_tensorflow_amd.data import mnist_like
    from distributed_tensorflow_amd.failure import ClusterLostError
    from distributed_tensorflow_amd.layers import Dense
    from distributed_tensorflow_amd.models import Sequential
    from distributed_tensorflow_amd.parallel import Communicator, SyncReplicasStrategy, init_process_group
    init_process_group("gloo", rank=rank, world_size=world, master_addr="127.0.0.1", master_port=port)
    m = Sequential([Dense(16, input_dim=784, activation="relu"), Dense(10, activation="softmax")])
    strat = SyncReplicasStrategy(Communicator())
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", strategy=strat, device="cpu", seed=1)
    x, y = mnist_like(32, seed=rank)
    seen = []
    for i in range(4):
        m.train_step(x, y)
        if i == 2 and rank == 1:
            m.store.global_step += 1          # this replica's counter leaves the group's (a stale or repeated update)
        try:
            seen.append(strat.agree(i == 1, int(m.store.global_step.item())))
        except ClusterLostError as e:
            seen.append(str(e))
            break
    out_q.put((rank, seen))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_replicas_out_of_step_are_detected():
    """One aggregated update per collective keeps every replica's ``global_step`` equal; the periodic agreement
    exchange (the chief's checkpoint decision) doubles as the check, and a replica whose counter has left the group
    makes EVERY replica raise - the lock-step counterpart of SyncReplicasOptimizer dropping stale gradients."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_out_of_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        assert res[rank][:2] == [False, True]                  # in step: the chief's decision reaches everybody
        assert "out of step" in res[rank][2] and "from 3 to 4" in res[rank][2]


def _bn_worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from distributed_tensorflow_amd.data import mnist_like
    from distributed_tensorflow_amd.layers import BatchNormalization, Dense
    from distributed_tensorflow_amd.models import Sequential
    from distributed_tensorflow_amd.parallel import Communicator, SyncReplicasStrategy, init_process_group
    init_process_group("gloo", rank=rank, world_size=world, master_addr="127.0.0.1", master_port=port)
    m = Sequential([Dense(16, input_dim=784), BatchNormalization(momentum=0.5), Dense(10, activation="softmax")])
    strat = SyncReplicasStrategy(Communicator())
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", strategy=strat, device="cpu", seed=1 + rank)
    mm = m.store.variables["batch_normalization_1/moving_mean"].tensor
    if rank == 0:
        mm.fill_(3.0)                      # as if the chief had restored a checkpoint
    strat.broadcast_state(m)
    received = mm.clone()
    x, y = mnist_like(64, seed=rank)
    for _ in range(3):
        m.train_step(x, y)
    out_q.put((rank, received.tolist(), mm.tolist(), m.store.flat.sum().item()))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_batchnorm_under_synchronous_replicas():
    """The non-trained moving statistics travel with the chief's state (after a restore they exist only there); during
    training they stay each replica's own - its batches' statistics - while gamma / beta are averaged like any variable."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bn_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == [3.0] * 16                 # broadcast from the chief
    assert res[0][2] != res[1][2]                               # then local: different batches
    assert abs(res[0][3] - res[1][3]) < 1e-4                    # the trained variables stay identical

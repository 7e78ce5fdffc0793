"""PeerRegion on its own: one-sided put / get, the one-launch gather / scatter of equal shards, range tables, regions of
different sizes.  CPU: two ranks as threads over POSIX shared memory.  GPU: a one-rank world (a process cannot map its
own hipIpc handle, and the multi-process path is covered by tests/test_async_peer_region_gpu.py and
tests/test_async_ps_region.py) - it exercises the kernels' indexing against the rank's own region."""
import threading

import pytest
import torch

from distributed_tensorflow_amd.control import ControlClient, ControlServer
from distributed_tensorflow_amd.parallel.peer_region import PeerRegion


@pytest.fixture()
def server():
    s = ControlServer("127.0.0.1", 0)
    yield s
    s.stop()


def _ranks(server, world, fn):
    out, errs = [None] * world, []

    def run(r):
        try:
            out[r] = fn(r, ControlClient("127.0.0.1", server.port))
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not errs, errs
    return out


def test_two_ranks_put_get_gather_scatter_and_ranges_cpu(server):
    n = 64

    def body(rank, cp):
        reg = PeerRegion(rank, 2, 4 * n, "cpu", cp, name="t/region")
        reg.local[:n] = torch.arange(n, dtype=torch.float32) + 1000 * rank          # "variable plane" of this rank
        cp.barrier("filled", 2)
        flat = torch.empty(2 * n)
        reg.gather(0, flat)
        assert torch.equal(flat, torch.cat([torch.arange(n) + 0.0, torch.arange(n) + 1000.0]))
        grad = torch.arange(2 * n, dtype=torch.float32) * (rank + 1)
        reg.scatter(n * (1 + rank), grad)                 # my slot in everybody's inbox
        cp.barrier("scattered", 2)
        for w in range(2):                                # what worker w sent me
            assert torch.equal(reg.local[n * (1 + w):n * (2 + w)], torch.arange(rank * n, (rank + 1) * n) * (w + 1.0))
        # range table: interleaved pieces of a flat buffer against packed positions in both peers' regions
        table = reg.make_ranges([(0, 3 * n, 0, 8), (1, 3 * n, 8, 16), (0, 3 * n + 8, 24, 4), (1, 3 * n + 16, 28, 12)])
        cp.barrier("tables", 2)
        if rank == 0:
            reg.move_ranges(table, False, 0, torch.arange(40, dtype=torch.float32) + 7)
        cp.barrier("moved", 2)
        back = torch.zeros(40)
        reg.move_ranges(table, True, 0, back)
        assert torch.equal(back, torch.arange(40) + 7.0)
        with pytest.raises(ValueError):
            reg.make_ranges([(1, 4 * n - 2, 0, 4)])       # past the end of rank 1's region
        cp.barrier("done", 2)
        reg.close()
        return True

    assert _ranks(server, 2, body) == [True, True]


def test_regions_of_different_sizes_cpu(server):
    """An owner's large region and a worker's token one: bounds are those of the rank that is addressed."""
    def body(rank, cp):
        reg = PeerRegion(rank, 2, 0, "cpu", cp, name="t/asym", sizes=[256, 8])
        assert reg.local.numel() == (256 if rank == 0 else 8)
        if rank == 1:
            reg.put(0, 200, torch.full((56,), 3.0))
            with pytest.raises(ValueError):
                reg.put(0, 250, torch.zeros(8))
        else:
            with pytest.raises(ValueError):
                reg.put(1, 4, torch.zeros(8))             # fits my region, not rank 1's
        cp.barrier("put", 2)
        if rank == 0:
            assert float(reg.local[200:].sum()) == 56 * 3.0
        with pytest.raises(ValueError):
            reg.gather(0, torch.empty(32))                # equal-shard collectives are bounded by the smallest region
        cp.barrier("done", 2)
        reg.close()
        return True

    assert _ranks(server, 2, body) == [True, True]


@pytest.mark.gpu
def test_kernels_against_the_own_region_gpu(server):
    dev = torch.device("cuda:0")
    cp = ControlClient("127.0.0.1", server.port)
    n = 4096 + 64
    reg = PeerRegion(0, 1, 4 * n, dev, cp, name="t/gpu")
    src = torch.randn(n, device=dev)
    reg.scatter(n, src)                                   # one shard: the whole tensor at offset n
    out = torch.empty(n, device=dev)
    reg.gather(n, out)
    assert torch.equal(out, src)
    assert torch.equal(reg.local[n:2 * n], src) and float(reg.local[:n].abs().sum()) == 0.0
    # interleaved ranges of a flat buffer <-> packed positions, with a base offset (which inbox slot)
    flat = torch.randn(3000, device=dev)
    ranges = [(0, 0, 64, 512), (0, 512, 1024, 256), (0, 768, 2000, 1000), (0, 1768, 0, 64)]
    table = reg.make_ranges(ranges)
    assert table.handle                                   # the one-launch path, not per-range copies
    reg.move_ranges(table, False, 2 * n, flat)
    packed = torch.cat([flat[64:576], flat[1024:1280], flat[2000:3000], flat[0:64]])
    assert torch.equal(reg.local[2 * n:2 * n + 1832], packed)
    back = torch.zeros(3000, device=dev)
    reg.move_ranges(table, True, 2 * n, back)
    mask = torch.zeros(3000, dtype=torch.bool, device=dev)
    for _, _, lo, k in ranges:
        mask[lo:lo + k] = True
    assert torch.equal(back[mask], flat[mask]) and float(back[~mask].abs().sum()) == 0.0
    with pytest.raises(RuntimeError):
        reg.move_ranges(table, False, 4 * n - 1000, flat)             # base + extent past the region
    with pytest.raises(RuntimeError):
        reg.move_ranges(table, True, 0, torch.zeros(2000, device=dev))    # local buffer shorter than the ranges reach
    unaligned = reg.make_ranges([(0, 1, 3, 10)])
    assert not unaligned.handle                           # falls back to runtime copies
    reg.move_ranges(unaligned, False, 0, flat)
    assert torch.equal(reg.local[1:11], flat[3:13])
    reg.close()

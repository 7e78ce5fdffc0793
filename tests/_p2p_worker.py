"""One rank of the peer-pointer all-reduce test (launched by test_p2p_allreduce_gpu.py; ranks share cuda:0)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.parallel.p2p_allreduce import P2PAllReduce  # noqa: E402


def delayed_rank(rank, world, dev):
    """A rank that shows up later than the timeout: the waiting ranks' result is NaN (never the local gradient) and
    their error check raises; the late rank itself still finds everybody's contribution."""
    import time
    from distributed_tensorflow_amd.parallel.p2p_allreduce import P2PTimeoutError
    ar = P2PAllReduce(4096, dev, timeout_s=0.3)
    t = torch.full((4096,), float(rank + 1), device=dev)
    ar.all_reduce_sum(t)                        # a normal call first
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.full((4096,), world * (world + 1) / 2.0))
    dist.barrier()
    if rank == world - 1:
        time.sleep(1.5)
    t = torch.full((4096,), float(rank + 1), device=dev)
    ar.all_reduce_sum(t)
    torch.cuda.synchronize()
    if rank == world - 1:
        assert torch.equal(t.cpu(), torch.full((4096,), world * (world + 1) / 2.0))
        ar.check()
    else:
        assert torch.isnan(t).all(), t[:4]
        try:
            ar.check()
            raise AssertionError("check() did not report the missed reduction")
        except P2PTimeoutError:
            pass
    dist.barrier()
    print("P2P_OK rank %d delayed" % rank, flush=True)
    ar.close()
    dist.destroy_process_group()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if os.environ.get("P2P_DELAY_TEST") == "1":
        return delayed_rank(rank, world, dev)
    cap = 120000
    ar = P2PAllReduce(cap, dev)
    g = torch.Generator().manual_seed(1234)             # same stream of sizes on every rank
    for it in range(40):
        n = cap if it % 5 == 0 else int(torch.randint(1, cap, (1,), generator=g))
        parts = [torch.randn(n, generator=torch.Generator().manual_seed(1000 * it + r)) for r in range(world)]
        want = parts[0].clone()
        for r in range(1, world):
            want += parts[r]                             # the kernel adds in rank order too: bitwise comparable
        t = parts[rank].to(dev)
        ar.all_reduce_sum(t)
        got = t.cpu()
        assert torch.equal(got, want), "iteration %d n=%d: max diff %g" % (it, n, (got - want).abs().max())
    # back-to-back calls without a host synchronisation in between (double buffering + one flag round per call)
    t = torch.full((cap,), float(rank + 1), device=dev)
    for _ in range(25):
        ar.all_reduce_sum(t)
        t.mul_(1.0 / world)
    torch.cuda.synchronize()
    mean = (world + 1) / 2.0
    assert torch.allclose(t.cpu(), torch.full((cap,), mean), rtol=1e-5), t[:4]
    ar.check()
    dist.barrier()
    print("P2P_OK rank %d fine_grained=%s" % (rank, ar.fine_grained), flush=True)
    ar.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

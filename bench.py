#!/usr/bin/env python
"""Headline benchmark: examples/sec of sync-replica data-parallel training of the
MNIST-shaped MLP (784 -> 128 relu -> dropout .3 -> 128 relu -> dropout .3 -> 10 softmax,
Adam, softmax cross-entropy) in fp32 on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 50 --warmup 10

Every timed step is a complete training step on fresh device-resident synthetic data:
forward, loss, backward, gradient exchange, Adam apply, global_step increment.  The
gradient exchange is one RCCL all-reduce of the flat gradient plus a replicated apply when
the model is small (latency-bound; the default model is), and bucketed reduce-scatter ->
sharded Adam -> all-gather overlapped with backward when it is large (--model wide_mlp).
Weak scaling: the per-GPU batch is fixed.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# must be in place before the first GPU call of the process: RCCL's intra-node transport needs dmabuf IPC here
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# CPU proxy of the reference workload (examples/s, 1 thread, PyTorch CPU): the denominator of "vs_baseline"
BASELINE_EXAMPLES_PER_SEC = 37700.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="mnist_mlp", choices=["mnist_mlp", "xor_mlp", "mnist_cnn", "wide_mlp"])
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="0 = the model's default")
    ap.add_argument("--pool", type=int, default=4, help="distinct device-resident synthetic batches cycled through")
    ap.add_argument("--graph", type=int, default=-1, help="replay the step from a hipGraph (-1 = auto: single GPU only)")
    ap.add_argument("--fused", type=int, default=-1,
                    help="one-launch whole-step kernel for the small-batch reference MLP (-1 = when available, 0 = off)")
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    ap.add_argument("--width", type=int, default=16384, help="wide_mlp hidden width")
    ap.add_argument("--depth", type=int, default=5, help="wide_mlp hidden depth")
    ap.add_argument("--mode", default="sync", choices=["sync", "async"],
                    help="sync: synchronous replicas (headline); async: parameter-server mode, owners colocated with workers")
    ap.add_argument("--max-staleness", type=int, default=4, help="async mode: bound on worker clock skew (-1 = unbounded)")
    ap.add_argument("--single-rank-collectives", type=int, default=0,
                    help="testing aid: run the multi-GPU code path (RCCL calls included) on a 1-rank process group")
    ap.add_argument("--allreduce", default="rccl", choices=["rccl", "p2p"],
                    help="small-model gradient exchange: RCCL all-reduce, or the one-shot peer-pointer kernel (one node)")
    ap.add_argument("--matmul", default="fp32", choices=["fp32", "fp32_split", "bf16"],
                    help="fp32 (default): fp32 MFMA.  fp32_split: fp32 operands split exactly into three bf16 limbs, six "
                         "limb products accumulated in fp32 on the bf16 matrix cores (fp32-level accuracy).  bf16: opt-in "
                         "mixed precision for wide dense layers (bf16 operands, fp32 accumulation)")
    ap.add_argument("--eval", type=int, default=0, help="also train for this many extra steps and report test accuracy")
    return ap.parse_args()


DEFAULT_BATCH = {"mnist_mlp": 262144, "xor_mlp": 32768, "mnist_cnn": 4096, "wide_mlp": 8192}


def main():
    args = parse_args()
    if args.allreduce == "p2p":
        os.environ["DTF_ALLREDUCE"] = "p2p"
    import torch
    import torch.distributed as dist

    from distributed_tensorflow_amd import data, models
    from distributed_tensorflow_amd.ops import _native
    from distributed_tensorflow_amd.parallel import Communicator, SyncReplicasStrategy, init_process_group

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    _native.kernels()  # fail loudly if the HIP library is missing

    strategy = None
    multi = world > 1 or bool(args.single_rank_collectives)
    if args.mode == "async":
        # asynchronous parameter-server mode with colocated owners: every rank is a worker AND owns 1/N of the
        # flat variable buffer in its HBM; packed push / pull per owner per step, bounded staleness
        from distributed_tensorflow_amd.cluster import ClusterSpec, Server
        from distributed_tensorflow_amd.parallel.async_ps import AsyncPSStrategy
        base = int(os.environ.get("MASTER_PORT", "29500")) + 16
        if world > 1:
            init_process_group("nccl", device=dev)
        hosts = ["127.0.0.1:%d" % (base + i) for i in range(world)]
        server = Server(ClusterSpec({"worker": hosts}), "worker", rank, config={"device": str(dev)})
        strategy = AsyncPSStrategy(server, max_staleness=args.max_staleness if args.max_staleness >= 0 else None)
        multi = world > 1
    elif multi:
        if world == 1:
            os.environ["DTF_FORCE_COLLECTIVES"] = "1"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            init_process_group("nccl", rank=0, world_size=1, device=dev)
        else:
            init_process_group("nccl", device=dev)
        strategy = SyncReplicasStrategy(Communicator(), bucket_bytes=int(args.bucket_mb * (1 << 20)))

    B = args.batch_per_gpu or DEFAULT_BATCH[args.model]
    if args.model == "mnist_mlp":
        model = models.mnist_mlp()
        loss = "sparse_categorical_crossentropy"
        make = lambda n, seed: data.mnist_like(n, seed=seed, device=dev)
    elif args.model == "xor_mlp":
        model = models.xor_mlp()
        loss = "mean_squared_error"
        make = lambda n, seed: data.get_data(n, val_set_size=0, seed=seed, device=dev)[:2]
    elif args.model == "mnist_cnn":
        model = models.mnist_cnn()
        loss = "sparse_categorical_crossentropy"
        def make(n, seed):
            x, y = data.mnist_like(n, seed=seed, device=dev)
            return x.view(n, 28, 28, 1), y
    else:
        model = models.wide_mlp(width=args.width, depth=args.depth)
        loss = "sparse_categorical_crossentropy"
        make = lambda n, seed: data.mnist_like(n, seed=seed, device=dev)
    model.compile(loss=loss, optimizer="adam", metrics=["accuracy"], strategy=strategy, device=dev, seed=0,
                  matmul_dtype=args.matmul)

    pool = max(1, args.pool)
    # small-batch reference network on one GPU: the whole step (and runs of consecutive steps) is one launch
    plan = model.fused_step_plan(B) if (args.fused != 0 and not multi) else None
    if args.fused == 1 and plan is None:
        raise SystemExit("--fused 1: no fused step kernel for this model / batch / world size")
    if plan is not None:
        pool = max(pool, 128)
    x_all, y_all = make(B * pool, 1000 + rank)
    batches = [(x_all[i * B:(i + 1) * B], y_all[i * B:(i + 1) * B]) for i in range(pool)]

    use_graph = plan is None and args.mode == "sync" and ((args.graph == 1) or (args.graph == -1 and not multi))
    graphs = grad_graphs = None
    if use_graph and not multi:
        graphs = [model._graphed_steps(bx, by, B, 1) for bx, by in batches]
    elif plan is None and args.graph == 1 and multi and getattr(strategy, "replicated", False):
        # small per-GPU batches (host-bound eager launches): forward+loss+backward replays from a hipGraph, the
        # gradient all-reduce and the replicated apply follow eagerly (RCCL stays outside the graph).  At the
        # default batch the GPU is the bottleneck and eager launches measured 4 % faster, so this is opt-in.
        grad_graphs = [model.graphed_gradients(bx, by) for bx, by in batches]
        use_graph = True

    def run(first, count):
        """Training steps first .. first+count-1; step i consumes batch i % pool."""
        if plan is not None:
            i = first
            while i < first + count:
                k = min(pool - i % pool, first + count - i)
                off = (i % pool) * B
                model.train_steps_fused(x_all[off:], y_all[off:], B, k, plan)
                i += k
            return
        for i in range(first, first + count):
            if graphs is not None:
                graphs[i % pool].replay()
            elif grad_graphs is not None:
                grad_graphs[i % pool].replay()
                model.strategy.apply_gradients(model)
                model._host_step += 1
            else:
                bx, by = batches[i % pool]
                model.train_step(bx, by)

    def step(i):
        run(i, 1)

    def barrier():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[dev.index])
        else:
            dist.barrier()       # DTF_COMM_BACKEND=gloo: ranks time-sharing one GPU (rehearsal on a one-GPU box)

    def fence():
        if multi:
            barrier()
        torch.cuda.synchronize()

    run(0, args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # diagnostic, outside the timed region: what one gradient exchange of the latency-bound path costs on this node
    # (back-to-back all-reduces of the flat gradient buffer; every rank takes part)
    exchange_us = None
    if multi and args.mode == "sync" and getattr(strategy, "replicated", False) and dist.get_backend() == "nccl":
        buf = torch.zeros_like(model.store.flat_grad)
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        exchange_us = e0.elapsed_time(e1) * 1000.0 / 20

    logs = model.read_stats()
    test_acc = None
    if args.eval > 0:
        for i in range(args.eval):
            step(i)
        xt, yt = make(8192, 99)
        test_acc = model.evaluate(xt, yt, batch_size=8192)[1]

    staleness = None
    if args.mode == "async":
        # observed staleness of this rank's pushes: owner updates applied between its pull and its push
        staleness = {str(k): v for k, v in sorted(strategy.staleness_hist.items())}
        strategy.worker_done()
        strategy.cp0.barrier("bench_done", world, timeout_s=300)
        if rank == 0:
            strategy.shutdown_owners()
    global_batch = B * world
    value = global_batch * args.steps / elapsed
    if rank == 0:
        out = {
            "metric": "examples/sec (whole node) %s MNIST-shape model" % ("sync-DP" if args.mode == "sync" else "async-PS"),
            "value": value,
            "unit": "examples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": value / BASELINE_EXAMPLES_PER_SEC,
            "dtype": {"fp32": "fp32", "fp32_split": "fp32 (operands as three bf16 limbs, fp32 accumulate, fp32-level accuracy)",
                      "bf16": "bf16 operands / fp32 accumulate (wide dense layers), fp32 elsewhere"}[args.matmul],
            "data": "synthetic (device-resident MNIST-shaped inputs, random-init weights)",
            "config": {"model": args.model, "global_batch": global_batch, "batch_per_gpu": B, "seq_len": None,
                       "parallelism": ("dp%d" if args.mode == "sync" else "async-ps%d") % world, "params": model.store.num_params,
                       "optimizer": "adam", "hip_graph": bool(use_graph), "fused_step_kernel": plan is not None, "allreduce": args.allreduce if multi else None,
                       "allreduce_us": exchange_us,
                       "train_loss": logs.get("loss"), "train_acc": logs.get("acc"), "test_acc": test_acc,
                       **({"max_staleness": args.max_staleness, "staleness_histogram_rank0": staleness}
                          if staleness is not None else {})},
        }
        print(json.dumps(out), flush=True)
    if multi:
        barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

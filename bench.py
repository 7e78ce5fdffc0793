#!/usr/bin/env python
"""Headline benchmark: examples/sec of sync-replica data-parallel training of the
MNIST-shaped MLP (784 -> 128 relu -> dropout .3 -> 128 relu -> dropout .3 -> 10 softmax,
Adam, softmax cross-entropy) in fp32 on N MI355X GPUs of one node, plus test accuracy.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus 8 --steps 50 --warmup 10          # starts 8 ranks itself, one per GPU, over RCCL
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 50 --warmup 10

Launching.  Under torchrun (RANK / WORLD_SIZE in the environment) the process is one rank.  Started plainly with
``--gpus N > 1`` it is a launcher: before anything touches a GPU it starts N rank processes of itself (one per GPU,
the reference's "N worker processes", /root/reference/example.py:124-141), relays rank 0's JSON line and exits
non-zero if any rank fails.  A machine with fewer than N GPUs is an error, never a silent smaller run: the printed
``n_gpus`` always equals ``--gpus``.

Every timed step is a complete training step on fresh device-resident synthetic data:
forward, loss, backward, gradient exchange, Adam apply, global_step increment.  The
gradient exchange is one RCCL all-reduce of the flat gradient plus a replicated apply when
the model is small (latency-bound; the default model is), and bucketed reduce-scatter ->
sharded Adam -> all-gather overlapped with backward when it is large (--model wide_mlp).
Weak scaling: the per-GPU batch is fixed.  Rank 0 prints ONE JSON line.

Outside the timed region: the ``--warmup`` steps, then further untimed steps until the step time is steady (a box
that has just started has not reached its sustained clocks), and - after the timed steps - ``--eval`` more training
steps followed by the accuracy on a held-out synthetic test set (``config.test_acc``; the reference prints its
validation accuracy, /root/reference/example.py:222-226).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# must be in place before the first GPU call of the process: RCCL's intra-node transport needs dmabuf IPC here
if os.environ.get("DTF_KEEP_IPC_MODE") != "1":
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# CPU proxy of the reference workload (examples/s, 1 thread, PyTorch CPU): the denominator of "vs_baseline"
BASELINE_EXAMPLES_PER_SEC = 37700.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="mnist_mlp", choices=["mnist_mlp", "xor_mlp", "mnist_cnn", "wide_mlp"])
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="0 = the model's default")
    ap.add_argument("--pool", type=int, default=4, help="distinct device-resident synthetic batches cycled through")
    ap.add_argument("--graph", type=int, default=-1,
                    help="replay the step from a hipGraph (-1 = auto: the whole step on one GPU; forward+backward of a "
                         "small-batch replica whose launches would be host-bound; 0 = eager launches)")
    ap.add_argument("--fused", type=int, default=-1,
                    help="one-launch whole-step kernel for the small-batch reference MLP (-1 = when available, 0 = off)")
    ap.add_argument("--workers", type=int, default=1,
                    help="asynchronous workers inside the GPU for the one-launch step kernel of the small-batch reference "
                         "MLP: lock-free updates of shared variables, one shared global_step (the reference's mode)")
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
    ap.add_argument("--eval", type=int, default=-1,
                    help="after the timed steps, train this many more steps (untimed) and report the accuracy on a held-out "
                         "test set (-1 = auto: 100 steps, 60 for wide_mlp; 0 = skip)")
    ap.add_argument("--steady", type=float, default=0.5,
                    help="after --warmup, keep stepping (untimed) for at least this many seconds and until two consecutive "
                         "10-step windows agree within 2 %% (0 = off)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu: plumbing rehearsal of the launcher and the collectives over gloo (not a benchmark)")
    return ap.parse_args(argv)


DEFAULT_BATCH = {"mnist_mlp": 262144, "xor_mlp": 32768, "mnist_cnn": 16384, "wide_mlp": 8192}


# ----------------------------------------------------------------------------------------------- launcher
def _under_launcher() -> bool:
    """True when some launcher (torchrun, or this script's own) has already made this process one rank."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# How the rank processes of a multi-GPU run bring up their process group, in the order tried.  A value of None
# removes the variable from the rank's environment.  The first set whose every rank passes the pre-flight check
# (group initialised, one all-reduce verified against its analytic sum) runs the benchmark; its name is printed as
# ``config.launch_variant``.  The last one stages the collectives through host memory over gloo: slow, but a
# labelled measurement instead of none.
LAUNCH_VARIANTS = {
    "cuda": [
        ("rccl-eager-init", {}),
        ("rccl-lazy-init", {"DTF_PG_LAZY_INIT": "1"}),
        ("rccl-lazy-init-default-ipc", {"DTF_PG_LAZY_INIT": "1", "DTF_KEEP_IPC_MODE": "1",
                                        "HSA_ENABLE_IPC_MODE_LEGACY": None}),
        ("gloo-host-staged", {"DTF_COMM_BACKEND": "gloo"}),
    ],
    "cpu": [("gloo-cpu", {}), ("gloo-cpu-second-try", {})],
}
INIT_TIMEOUT_S = float(os.environ.get("DTF_BENCH_INIT_TIMEOUT", "420"))


class _LocalCoordinator:
    """One supervisor owns every rank (``python bench.py --gpus N`` started plainly)."""

    def __init__(self, world):
        self.world = world
        self._fail = set()
        self._ok = {}

    def port(self, v):
        return _free_port()

    def report_fail(self, v):
        self._fail.add(v)

    def failed(self, v):
        return v in self._fail

    def report_ok(self, v):
        self._ok[v] = self._ok.get(v, 0) + 1

    def all_ok(self, v):
        return self._ok.get(v, 0) >= self.world

    def all_down(self, v, timeout_s=60.0):
        return True


class _StoreCoordinator:
    """One supervisor per rank (the driver's ``python -m torch.distributed.run ... bench.py``): the supervisors agree
    through the launcher's own TCP store (they are clients of it; no GPU call is involved)."""

    def __init__(self, rank, world):
        import datetime
        from torch.distributed import TCPStore
        self.rank, self.world = rank, world
        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() in ("1", "true")
        self.store = TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world_size=None,
                              is_master=(rank == 0 and not agent), timeout=datetime.timedelta(seconds=600),
                              wait_for_workers=False)
        self.ns = "dtf_bench/%s/" % os.environ.get("TORCHELASTIC_RUN_ID", "run")

    def port(self, v):
        key = self.ns + "v%d/port" % v
        if self.rank == 0:
            self.store.set(key, str(_free_port()))
        return int(self.store.get(key).decode())

    def report_fail(self, v):
        self.store.set(self.ns + "v%d/fail" % v, "1")

    def failed(self, v):
        return bool(self.store.check([self.ns + "v%d/fail" % v]))

    def report_ok(self, v):
        self.store.add(self.ns + "v%d/ok" % v, 1)

    def all_ok(self, v):
        return self.store.add(self.ns + "v%d/ok" % v, 0) >= self.world

    def all_down(self, v, timeout_s=60.0):
        key = self.ns + "v%d/down" % v
        self.store.add(key, 1)
        deadline = time.time() + timeout_s
        while time.time() < deadline:
            if self.store.add(key, 0) >= self.world:
                return True
            time.sleep(0.05)
        return False


def _stop(procs):
    """End exactly the processes started here (a rank that failed leaves its peers blocked in a collective)."""
    for p in procs:
        if p.poll() is None:
            p.terminate()
    deadline = time.time() + 10
    for p in procs:
        try:
            p.wait(timeout=max(0.1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


def supervise(args, argv, my_ranks, coord, self_launched) -> int:
    """Run the rank processes ``my_ranks`` of a ``coord.world``-rank job as fresh child processes and relay rank 0's
    JSON line.  The supervisor never touches a GPU, and a rank that has is never re-executed: when any rank of the job
    fails before the whole group has passed its pre-flight check, every supervisor ends its children and starts a new
    set under the next entry of ``LAUNCH_VARIANTS``.  Returns the exit code: 0 only if every rank exited 0 and rank 0
    printed a line (the reference starts N worker processes by hand, /root/reference/example.py:124-141)."""
    import tempfile
    import threading
    import signal
    world = coord.world
    variants = LAUNCH_VARIANTS[args.device]
    if args.device == "cuda" and os.environ.get("DTF_COMM_BACKEND"):
        variants = [("forced-%s" % os.environ["DTF_COMM_BACKEND"], {})]     # the caller chose the data plane

    def _terminated(signum, frame):      # a launcher above ends its ranks with SIGTERM: take the children along
        raise SystemExit(128 + signum)

    signal.signal(signal.SIGTERM, _terminated)
    local_rank = {r: (int(os.environ.get("LOCAL_RANK", r)) if len(my_ranks) == 1 else r) for r in my_ranks}
    only = os.environ.get("DTF_BENCH_LAUNCH_VARIANT")
    status_dir = tempfile.mkdtemp(prefix="dtf_bench_")
    last_rc = 1
    for v, (name, overrides) in enumerate(variants):
        if only and name != only:
            continue
        port = coord.port(v)
        procs, lines = [], []
        for r in my_ranks:
            env = dict(os.environ)
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)    # the children's rank 0 serves their own store on ``port``
            env.update({"RANK": str(r), "LOCAL_RANK": str(local_rank[r]), "WORLD_SIZE": str(world),
                        "LOCAL_WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "DTF_BENCH_CHILD": "1",
                        "DTF_BENCH_SELF_LAUNCHED": "1" if self_launched else "0", "DTF_BENCH_VARIANT": name,
                        "DTF_BENCH_VARIANT_INDEX": str(v), "DTF_BENCH_STATUS": os.path.join(status_dir, "v%d_r%d" % (v, r))})
            for k, val in overrides.items():
                if val is None:
                    env.pop(k, None)
                else:
                    env[k] = val
            env.setdefault("OMP_NUM_THREADS", "2")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
        pump = None
        if 0 in my_ranks:
            out0 = procs[my_ranks.index(0)].stdout

            def _pump(stream=out0, sink=lines):
                for line in stream:
                    sink.append(line.rstrip("\n"))

            pump = threading.Thread(target=_pump, daemon=True)
            pump.start()
        reported, group_ok, failed = set(), False, None
        t_start = time.time()
        try:
            pending = set(range(len(procs)))
            while pending:
                for i in sorted(pending):
                    rc = procs[i].poll()
                    if rc is None:
                        continue
                    pending.discard(i)
                    if rc != 0 and failed is None:
                        failed = (my_ranks[i], rc)
                if not group_ok:
                    for i, r in enumerate(my_ranks):
                        if i not in reported and os.path.exists(os.path.join(status_dir, "v%d_r%d" % (v, r))):
                            reported.add(i)
                            coord.report_ok(v)
                    group_ok = coord.all_ok(v)
                if failed is None and (coord.failed(v) or (not group_ok and time.time() - t_start > INIT_TIMEOUT_S)):
                    failed = (-1, 1)
                if failed is not None:
                    # the pre-flight check ends with a collective, so one rank past it means every rank got through it
                    group_ok = group_ok or len(reported) > 0
                    coord.report_fail(v)
                    break
                time.sleep(0.05)
        finally:
            _stop(procs)
        if pump is not None:
            pump.join(timeout=5)
        if failed is None:
            result = [l for l in lines if l.startswith("{")]
            for l in lines:
                if not l.startswith("{"):
                    sys.stderr.write(l + "\n")
            if 0 in my_ranks:
                if not result:
                    sys.stderr.write("bench.py: rank 0 printed no result line\n")
                    return 1
                print(result[-1], flush=True)
            return 0
        last_rc = failed[1] if failed[1] > 0 else 1
        if last_rc == 78:        # EXIT_CONFIG
            return last_rc
        if group_ok:
            # every rank had passed the pre-flight check: this is a failure of the run itself, not of its bring-up
            sys.stderr.write("bench.py: rank %d of %d exited with status %d; no result\n" % (failed[0], world, failed[1]))
            return last_rc
        sys.stderr.write("bench.py: launch variant '%s' did not bring up all %d ranks (%s)%s\n"
                         % (name, world, "rank %d exited with status %d" % failed if failed[0] >= 0 else "a peer failed or timed out",
                            "; trying the next one" if v + 1 < len(variants) and not only else ""))
        coord.all_down(v)
    sys.stderr.write("bench.py: no launch variant brought up %d ranks; no result\n" % world)
    return last_rc


# ----------------------------------------------------------------------------------------------- one rank
def _preflight(torch, dist, dev, on_gpu, backend, rank, world):
    """Before anything is timed: one all-reduce of the flat-gradient size of the headline model (118 282 floats,
    473 KB) checked against its analytic sum in every rank, the peer-access matrix and the RCCL version.  Raises if
    the sum is wrong, so a broken transport ends the bring-up (and the supervisor tries its next variant)."""
    n = 118282
    idx = torch.arange(n, dtype=torch.float32, device=dev)
    t = (idx % 97.0) * float(rank + 1)
    if backend == "gloo" and on_gpu:
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    want = (idx % 97.0) * float(world * (world + 1) // 2)
    err = float((t - want).abs().max().item())
    if err != 0.0:
        raise SystemExit("bench.py: pre-flight all-reduce over %s is wrong by %g in rank %d" % (backend, err, rank))
    info = {"allreduce_473KB_max_abs_err": err, "backend": backend}
    if on_gpu:
        ndev = torch.cuda.device_count()
        me = dev.index or 0
        row = [bool(j == me or torch.cuda.can_device_access_peer(me, j)) for j in range(min(ndev, world))]
        rows = [None] * world
        dist.all_gather_object(rows, row)
        info["peer_access"] = rows
        if backend == "nccl":
            try:
                info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception as e:      # reporting only
                info["rccl_version"] = "unknown (%s)" % type(e).__name__
        info["hsa_enable_ipc_mode_legacy"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    if backend == "nccl":
        dist.barrier(device_ids=[dev.index])
    else:
        dist.barrier()
    return info


EXIT_CONFIG = 78     # a rank ended because the request cannot be met on this machine: no other launch variant will help


def _config_error(msg):
    sys.stderr.write(msg + "\n")
    sys.exit(EXIT_CONFIG)


def run_rank(args) -> None:
    if args.allreduce == "p2p":
        os.environ["DTF_ALLREDUCE"] = "p2p"
    import torch
    import torch.distributed as dist

    from distributed_tensorflow_amd import data, models
    from distributed_tensorflow_amd.ops import _native
    from distributed_tensorflow_amd.parallel import Communicator, SyncReplicasStrategy, init_process_group
    from distributed_tensorflow_amd.parallel.comm import pick_backend

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree"
                         % (args.gpus, world))
    on_gpu = args.device == "cuda"
    if on_gpu:
        # counting devices does not initialise the GPU runtime, so this check is safe in every rank before rendezvous
        ndev = torch.cuda.device_count()
        if ndev < 1:
            _config_error("bench.py needs a GPU (none visible)")
        backend = pick_backend(torch.device("cuda", 0))
        if world > ndev and backend == "nccl":
            _config_error("bench.py: --gpus %d but this machine exposes %d GPU(s); RCCL needs one GPU per rank "
                          "(DTF_COMM_BACKEND=gloo rehearses several ranks on one GPU)" % (world, ndev))
        dev = torch.device("cuda", local_rank % ndev)
        torch.cuda.set_device(dev)
        _native.kernels()  # fail loudly if the HIP library is missing
    else:
        dev = torch.device("cpu")
        backend = "gloo"
        torch.set_num_threads(max(1, int(os.environ.get("OMP_NUM_THREADS", "2"))))

    def sync_device():
        if on_gpu:
            torch.cuda.synchronize()

    # test hook: "v:r[,v:r...]" makes rank r of launch variant v die while the group comes up
    inject = os.environ.get("DTF_BENCH_INJECT_INIT_FAIL", "")
    if inject and "%s:%d" % (os.environ.get("DTF_BENCH_VARIANT_INDEX", "-"), rank) in inject.split(","):
        sys.stderr.write("bench.py: injected bring-up failure in rank %d\n" % rank)
        os._exit(86)

    strategy = None
    multi = world > 1 or bool(args.single_rank_collectives)
    if args.mode == "async":
        # asynchronous parameter-server mode with colocated owners: every rank is a worker AND owns 1/N of the
        # flat variable buffer in its HBM; packed push / pull per owner per step, bounded staleness
        from distributed_tensorflow_amd.cluster import ClusterSpec, Server
        from distributed_tensorflow_amd.parallel.async_ps import AsyncPSStrategy
        base = int(os.environ.get("MASTER_PORT", "29500")) + 16
        if world > 1:
            init_process_group(backend, device=dev if on_gpu else None)
        hosts = ["127.0.0.1:%d" % (base + i) for i in range(world)]
        server = Server(ClusterSpec({"worker": hosts}), "worker", rank, config={"device": str(dev)})
        strategy = AsyncPSStrategy(server, max_staleness=args.max_staleness if args.max_staleness >= 0 else None)
        multi = world > 1
    elif multi:
        if world == 1:
            os.environ["DTF_FORCE_COLLECTIVES"] = "1"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            init_process_group(backend, rank=0, world_size=1, device=dev if on_gpu else None)
        else:
            init_process_group(backend, device=dev if on_gpu else None)
        strategy = SyncReplicasStrategy(Communicator(), bucket_bytes=int(args.bucket_mb * (1 << 20)))
    data_plane = dist.get_backend() if dist.is_initialized() else None      # what barriers / timing reductions use
    preflight = _preflight(torch, dist, dev, on_gpu, data_plane, rank, world) if world > 1 else None
    if os.environ.get("DTF_BENCH_STATUS"):
        open(os.environ["DTF_BENCH_STATUS"], "w").write("ok\n")     # tells the supervisor that this rank is up

    B = args.batch_per_gpu or DEFAULT_BATCH[args.model]
    metrics = ["accuracy"]
    if args.model == "mnist_mlp":
        model = models.mnist_mlp()
        loss = "sparse_categorical_crossentropy"
        make = lambda n, seed: data.mnist_like(n, seed=seed, device=dev)
    elif args.model == "xor_mlp":
        model = models.xor_mlp()
        loss = "mean_squared_error"
        metrics = ["binary_rounded_accuracy"]       # the reference's accuracy (/root/reference/example.py:157-160)
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
    optimizer = "adam"
    if args.model == "wide_mlp":
        from distributed_tensorflow_amd.optimizers import Adam
        optimizer = Adam(lr=models.wide_mlp_learning_rate(args.width))     # Adam's default rate diverges at this width
    t_compile = time.perf_counter()
    model.compile(loss=loss, optimizer=optimizer, metrics=metrics, strategy=strategy, device=dev, seed=0,
                  matmul_dtype=args.matmul)
    sync_device()
    t_compile = time.perf_counter() - t_compile

    pool = max(1, args.pool)
    # small-batch reference network on one GPU: the whole step (and runs of consecutive steps) is one launch
    plan = model.fused_step_plan(B) if (args.fused != 0 and not multi) else None
    if args.fused == 1 and plan is None:
        raise SystemExit("--fused 1: no fused step kernel for this model / batch / world size")
    if plan is not None:
        pool = max(pool, 128 if args.workers <= 1 else 4096)     # mini-batches per launch
        model.async_workers = max(1, args.workers)
    x_all, y_all = make(B * pool, 1000 + rank)
    batches = [(x_all[i * B:(i + 1) * B], y_all[i * B:(i + 1) * B]) for i in range(pool)]

    graphs = grad_graphs = None
    replicated = bool(getattr(strategy, "replicated", False))
    if on_gpu and plan is None and args.mode == "sync" and args.graph != 0:
        if not multi:
            graphs = [model._graphed_steps(bx, by, B, 1) for bx, by in batches]
        elif replicated and (args.graph == 1 or B * model.store.num_params < (1 << 34)):
            # a replica whose step is short enough to be bound by the host's eager launches: forward+loss+backward
            # replays from a hipGraph, the gradient all-reduce and the replicated apply follow eagerly (RCCL stays
            # outside the graph).  At large batches the GPU is the bottleneck and eager launches are as fast.
            grad_graphs = [model.graphed_gradients(bx, by) for bx, by in batches]
    use_graph = graphs is not None or grad_graphs is not None

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

    def barrier():
        if data_plane == "nccl":
            dist.barrier(device_ids=[dev.index])
        elif data_plane is not None:
            dist.barrier()       # gloo: CPU ranks, or ranks time-sharing one GPU (rehearsal on a one-GPU box)

    def fence():
        if multi:
            barrier()
        sync_device()

    def max_over_ranks(seconds: float) -> float:
        if world > 1:
            t = torch.tensor([seconds], dtype=torch.float64, device=dev if data_plane == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return seconds

    # ---- untimed: the requested warm-up, then until the step time is steady (every rank takes the same decisions:
    # they are made on the maximum over ranks)
    run(0, args.warmup)
    fence()
    done = args.warmup
    steady_windows = 0
    if args.steady > 0:
        window, tol, cap = 10, 0.02, max(4.0, 8 * args.steady)
        t_begin = time.perf_counter()
        prev = None
        while True:
            t0 = time.perf_counter()
            run(done, window)
            fence()
            dt = max_over_ranks(time.perf_counter() - t0)
            done += window
            steady_windows += 1
            total = max_over_ranks(time.perf_counter() - t_begin)
            agree = prev is not None and abs(dt - prev) <= tol * max(dt, prev)
            if (agree and total >= args.steady) or total >= cap:
                break
            prev = dt

    # ---- timed region: exactly --steps steps between two fences
    fence()
    t0 = time.perf_counter()
    run(done, args.steps)
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    done += args.steps

    # diagnostic, outside the timed region: what one gradient exchange of the latency-bound path costs on this node
    # (back-to-back all-reduces of the flat gradient buffer; every rank takes part: the condition is rank-independent)
    exchange_us = None
    if multi and args.mode == "sync" and replicated and data_plane == "nccl":
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
    # ---- test accuracy (untimed): a fixed budget of further training steps, then a held-out set every rank evaluates
    # a slice of (the statistics are summed over ranks)
    test_acc = None
    n_eval = args.eval if args.eval >= 0 else (60 if args.model == "wide_mlp" else 100)
    if n_eval > 0:
        run(done, n_eval)
        done += n_eval
        if args.mode == "async":
            strategy.pull(model)         # the owners' current variables (no collective: every worker evaluates alone)
        n_test = 8192
        xt, yt = make(n_test * world, 99)
        res = model.evaluate(xt[rank * n_test:(rank + 1) * n_test], yt[rank * n_test:(rank + 1) * n_test],
                             batch_size=n_test)
        test_acc = res[1] if len(res) > 1 else None

    staleness = None
    if args.mode == "async":
        # observed staleness of this rank's pushes: owner updates applied between its pull and its push
        staleness = {str(k): v for k, v in sorted(strategy.staleness_hist.items())}
        strategy.worker_done()
        strategy.cp0.barrier("bench_done", world, timeout_s=300)
        if rank == 0:
            strategy.shutdown_owners()

    # what the data plane itself reports: its size and the device every rank computes on
    me = {"rank": rank, "device": str(dev), "pid": os.getpid()}
    if on_gpu:
        props = torch.cuda.get_device_properties(dev)
        me["gpu"] = props.name
        me["uuid"] = str(getattr(props, "uuid", ""))
    ranks = [me]
    group_size = 1
    if dist.is_initialized():
        group_size = dist.get_world_size()
        if world > 1:
            ranks = [None] * world
            dist.all_gather_object(ranks, me)
    if on_gpu and data_plane == "nccl" and world > 1:
        distinct = {(r["device"], r.get("uuid")) for r in ranks}
        if len(distinct) != world:
            raise SystemExit("bench.py: %d ranks but only %d distinct GPUs: %r" % (world, len(distinct), ranks))

    # device memory this rank holds: PyTorch's allocator (variables, gradients, activations, slots) plus the
    # peer-mapped region of the asynchronous mode, which is allocated outside it
    hbm = None
    if on_gpu:
        hbm = int(torch.cuda.max_memory_allocated(dev))
        if args.mode == "async" and getattr(strategy, "region_floats", None):
            hbm += 4 * int(strategy.region_floats)
    hbm_detail = strategy.hbm_report(model) if hasattr(strategy, "hbm_report") else None

    global_batch = B * world
    value = global_batch * args.steps / elapsed
    if rank == 0:
        out = {
            "metric": "examples/sec (whole node) %s %s" % (
                "sync-DP" if args.mode == "sync" else "async-PS",
                "reference XOR MLP (64-128-128-32)" if args.model == "xor_mlp" else "MNIST-shape model"),
            "value": value,
            "unit": "examples/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": value / BASELINE_EXAMPLES_PER_SEC,
            "dtype": {"fp32": "fp32", "fp32_split": "fp32 (operands as three bf16 limbs, fp32 accumulate, fp32-level accuracy)",
                      "bf16": "bf16 operands / fp32 accumulate (wide dense layers), fp32 elsewhere"}[args.matmul],
            "data": "synthetic (device-resident %s inputs, random-init weights)"
                    % ("XOR-task" if args.model == "xor_mlp" else "MNIST-shaped"),
            "config": {"model": args.model, "global_batch": global_batch, "batch_per_gpu": B, "seq_len": None,
                       "parallelism": ("dp%d" if args.mode == "sync" else "async-ps%d") % world, "params": model.store.num_params,
                       "optimizer": "adam", "learning_rate": model.optimizer.lr,
                       "compile_seconds": t_compile, "device_initialised": bool(getattr(model.store, "device_initialised", False)),
                       "hip_graph": bool(use_graph), "fused_step_kernel": plan is not None,
                       "intra_gpu_async_workers": (model.async_workers if plan is not None else None),
                       "allreduce": args.allreduce if (multi and args.mode == "sync") else None, "allreduce_us": exchange_us,
                       "train_loss": logs.get("loss"), "train_acc": logs.get("acc"), "test_acc": test_acc,
                       "eval_steps": n_eval if test_acc is not None else 0,
                       "steady_warmup_steps": steady_windows * 10,
                       "hbm_bytes_per_rank": hbm, "hbm_detail": hbm_detail,
                       "data_plane": ("peer regions (%s) + doorbell words; no control-plane call inside a step"
                                      % ("hipIpc over xGMI" if on_gpu else "shared memory")
                                      if args.mode == "async" else {"nccl": "rccl"}.get(data_plane, data_plane)),
                       # the size of the RCCL communicator; null whenever the collectives did not go over RCCL
                       "rccl_ranks": group_size if data_plane == "nccl" else None,
                       "backend": data_plane, "group_ranks": group_size,
                       "launch_variant": os.environ.get("DTF_BENCH_VARIANT"), "preflight": preflight,
                       "self_launched": os.environ.get("DTF_BENCH_SELF_LAUNCHED") == "1",
                       "ranks": ranks,
                       **({"device": "cpu (plumbing rehearsal, not a benchmark)"} if not on_gpu else {}),
                       **({"max_staleness": args.max_staleness, "staleness_histogram_rank0": staleness,
                           "inbox_depth": strategy.depth, "wait_seconds_rank0": strategy.wait_seconds,
                           "control_words": ("replicas in device memory, written by kernels"
                                             if getattr(strategy.blocks, "replicas", None) is not None
                                             else "shared memory, written by host code"),
                           "host_comm_seconds_rank0": strategy.comm_seconds}
                          if staleness is not None else {})},
        }
        print(json.dumps(out), flush=True)
    if multi:
        barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and os.environ.get("DTF_BENCH_CHILD") != "1":
        # this process only supervises: the ranks are fresh children, so a failed bring-up can be tried again another
        # way without ever re-using a process that has touched a GPU
        if _under_launcher():
            rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
            if world != args.gpus:
                raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree"
                                 % (args.gpus, world))
            sys.exit(supervise(args, argv, [rank], _StoreCoordinator(rank, world), self_launched=False))
        sys.exit(supervise(args, argv, list(range(args.gpus)), _LocalCoordinator(args.gpus), self_launched=True))
    if os.environ.get("DTF_BENCH_CHILD") == "1":
        import ctypes
        import signal
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)    # PR_SET_PDEATHSIG: never outlive the supervisor
    run_rank(args)


if __name__ == "__main__":
    main()

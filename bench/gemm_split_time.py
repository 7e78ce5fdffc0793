"""Time of the split-limb fp32 GEMMs (env MODES, SHAPES) - tuning aid."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402

dev = torch.device("cuda")
modes = [int(v) for v in os.environ.get("MODES", "0,8").split(",")]
shapes = [tuple(int(v) for v in sh.split(",")) for sh in os.environ.get("SHAPES", "131072,784,128;2048,16384,16384").split(";")]


def timeit(fn, rounds=12):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1000)
    return statistics.median(ts)


for (m, k, n) in shapes:
    xs = [torch.randn(m, k, device=dev) for _ in range(3)]
    w = torch.randn(k, n, device=dev) * 0.05
    b = torch.randn(n, device=dev)
    dz = torch.randn(m, n, device=dev)
    y = torch.empty(m, n, device=dev)
    dx = torch.empty(m, k, device=dev)
    dw = torch.empty(k, n, device=dev)
    db = torch.empty(n, device=dev)
    flop = 2.0 * m * k * n
    for mode in modes:
        F.set_gemm_split_products(mode)
        ws = torch.empty(max(F.dense_dw_workspace_floats(m, n, k), 1), device=dev)
        it = [0]

        def fwd():
            it[0] += 1
            F.dense_forward(xs[it[0] % 3], w, b, 1, out=y)

        def bdx():
            F.dense_backward_dx(dz, w, xs[0], 1, 1.0, out=dx)

        def bdw():
            it[0] += 1
            F.dense_backward_dw(xs[it[0] % 3], dz, dw, db, workspace=ws)

        r = ["%s %.1f us %.0f TF" % (nm, t, flop / t / 1e6) for nm, t in (("fwd", timeit(fwd)), ("dx", timeit(bdx)), ("dw", timeit(bdw)))]
        print("M=%d K=%d N=%d products=%d  %s" % (m, k, n, mode, "  ".join(r)), flush=True)
F.set_gemm_split_products(0)

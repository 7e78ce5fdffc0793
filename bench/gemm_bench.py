"""Time the dense-layer GEMM kernels in isolation (interleaved rounds, median + min)."""
import os
import sys
import statistics

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402

dev = torch.device("cuda")
M = int(os.environ.get("M", 32768))
shapes = [(M, 784, 128), (M, 128, 128), (M, 128, 10)]
if os.environ.get("SHAPE"):                      # SHAPE=m,k,n times that single product
    shapes = [tuple(int(v) for v in os.environ["SHAPE"].split(","))]
rounds = 20
BF16 = os.environ.get("BF16", "0") == "1"     # bf16-operand kernels (wide shapes only)


def timeit(fn):
    ts = []
    for _ in range(3):
        fn()
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1000)
    return statistics.median(ts), min(ts)


for (m, k, n) in shapes:
    xs = [(torch.rand(m, k, device=dev) * 2 - 1) for _ in range(3)]   # rotate inputs: > Infinity Cache at large m
    w = torch.rand(k, n, device=dev) * 2 - 1
    b = torch.rand(n, device=dev)
    dz = torch.rand(m, n, device=dev) * 2 - 1
    y = torch.empty(m, n, device=dev)
    dx = torch.empty(m, k, device=dev)
    dw = torch.empty(k, n, device=dev)
    db = torch.empty(n, device=dev)
    ws = torch.empty(max(F.dense_dw_workspace_floats(m, n, k), 1), device=dev)
    it = [0]

    def fwd():
        it[0] += 1
        F.dense_forward(xs[it[0] % 3], w, b, 1, out=y, bf16=BF16)

    def bdx():
        F.dense_backward_dx(dz, w, xs[0], 1, 1.0, out=dx, bf16=BF16)

    def bdw():
        it[0] += 1
        F.dense_backward_dw(xs[it[0] % 3], dz, dw, db, workspace=ws, bf16=BF16)

    flop = 2.0 * m * k * n
    for name, fn in (("fwd", fwd), ("dx", bdx), ("dw", bdw)):
        med, mn = timeit(fn)
        print("%-4s M=%d K=%d N=%d  median %.1f us  min %.1f us  %.1f TF (median)" % (name, m, k, n, med, mn, flop / med / 1e6), flush=True)

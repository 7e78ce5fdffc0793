"""fp32 GEMMs on the bf16 matrix cores (three-limb split): error against an fp64 product next to the native fp32 MFMA
kernels' error, and the time of both on the headline and wide-layer shapes."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)


def products(x, w, b, dz, mode):
    F.set_gemm_split_products(mode)
    m, k = x.shape
    n = w.shape[1]
    y = torch.empty(m, n, device=dev)
    dx = torch.empty(m, k, device=dev)
    dw = torch.empty(k, n, device=dev)
    db = torch.empty(n, device=dev)
    ws = torch.empty(max(F.dense_dw_workspace_floats(m, n, k), 1), device=dev)
    F.dense_forward(x, w, b, 0, out=y)
    F.dense_backward_dx(dz, w, None, 0, 1.0, out=dx)
    F.dense_backward_dw(x, dz, dw, db, workspace=ws)
    torch.cuda.synchronize()
    return y, dx, dw, db


def err(a, ref):
    d = (a.double() - ref)
    return d.abs().max().item() / ref.abs().max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


print("== accuracy: (max |err| / max |ref|, rms err / rms ref) against fp64", flush=True)
for (m, k, n) in [(4096, 784, 128), (2048, 4096, 2048), (8192, 128, 512)]:
    x = torch.randn(m, k, device=dev)
    w = torch.randn(k, n, device=dev) * 0.05
    b = torch.randn(n, device=dev)
    dz = torch.randn(m, n, device=dev)
    ry = x.double() @ w.double() + b.double()
    rdx = dz.double() @ w.double().t()
    rdw = x.double().t() @ dz.double()
    rdb = dz.double().sum(0)
    for mode in (0, 6, 8, 9):
        y, dx, dw, db = products(x, w, b, dz, mode)
        print("M=%d K=%d N=%d products=%d  fwd %.2e %.2e  dx %.2e %.2e  dw %.2e %.2e  db %.2e" % (
            (m, k, n, mode) + err(y, ry) + err(dx, rdx) + err(dw, rdw) + (err(db, rdb)[0],)), flush=True)


def timeit(fn, rounds=20):
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


print("== time (median us, fp32-equivalent TFLOP/s)", flush=True)
for (m, k, n) in [(131072, 784, 128), (2048, 16384, 16384)]:
    xs = [torch.randn(m, k, device=dev) for _ in range(3)]
    w = torch.randn(k, n, device=dev) * 0.05
    b = torch.randn(n, device=dev)
    dz = torch.randn(m, n, device=dev)
    y = torch.empty(m, n, device=dev)
    dx = torch.empty(m, k, device=dev)
    dw = torch.empty(k, n, device=dev)
    db = torch.empty(n, device=dev)
    flop = 2.0 * m * k * n
    for mode in (0, 6, 8, 9):
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

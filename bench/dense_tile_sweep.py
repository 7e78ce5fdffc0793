"""Times one dense product (SHAPE=m,k,n) forward / dgrad / wgrad under the tuning overrides given in the environment
(DTF_GEMM_TILE, DTF_FWD_SPLITS, DTF_DW_PLAN).  Back-to-back launches inside one timed region, rotating inputs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402

m, k, n = (int(v) for v in os.environ.get("SHAPE", "4096,3136,512").split(","))
dev = torch.device("cuda")
xs = [(torch.rand(m, k, device=dev) * 2 - 1) for _ in range(3)]
w = torch.rand(k, n, device=dev) * 2 - 1
b = torch.rand(n, device=dev)
dz = torch.rand(m, n, device=dev) * 2 - 1
y = torch.empty(m, n, device=dev)
dx = torch.empty(m, k, device=dev)
dw = torch.empty(k, n, device=dev)
db = torch.empty(n, device=dev)
ws = torch.empty(max(F.dense_dw_workspace_floats(m, n, k), 1), device=dev)
fs = int(os.environ.get("DTF_FWD_SPLITS", "1"))
fws = torch.empty(fs * m * n, device=dev) if fs > 1 else None
it = [0]


def fwd():
    it[0] += 1
    F.dense_forward(xs[it[0] % 3], w, b, 1, out=y, workspace=fws)


def bdx():
    F.dense_backward_dx(dz, w, xs[0], 1, 1.0, out=dx)


def bdw():
    it[0] += 1
    F.dense_backward_dw(xs[it[0] % 3], dz, dw, db, workspace=ws)


tag = " ".join("%s=%s" % (e, os.environ[e]) for e in ("DTF_GEMM_TILE", "DTF_FWD_SPLITS", "DTF_DW_PLAN") if e in os.environ)
for name, fn in (("fwd", fwd), ("dx", bdx), ("dw", bdw)):
    for _ in range(20):
        fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(50):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) * 1000 / 50
    print("%-3s %dx%dx%d %-40s %.1f us  %.1f TF" % (name, m, k, n, tag, us, 2.0 * m * k * n / us / 1e6), flush=True)

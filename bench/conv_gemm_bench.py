"""The three implicit-GEMM products of the MNIST CNN's second convolution (batch 4096, 14x14x32 -> 64, 5x5 'same') timed
one by one with device events; DTF_GEMM_TILE / DTF_DW_PLAN / DTF_GEMM_SINGLE select variants (read once per process)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402

dev = torch.device("cuda")
B, H, W, C, N, KH, KW = int(os.environ.get("B", 4096)), 14, 14, 32, 64, 5, 5
x = torch.rand(B, H, W, C, device=dev) - 0.5
w = (torch.rand(KH, KW, C, N, device=dev) - 0.5) * 0.1
b = torch.rand(N, device=dev)
OH, OW, pt, pl, pb, pr = F.conv_geometry(H, W, KH, KW, (1, 1), "same")
M, K = B * OH * OW, KH * KW * C
rt, sh, mk = F.conv_tables_unpadded(B, OH, OW, H, W, KH, KW, pt, pl, C, dev)
rt2, _, mk2 = F.conv_tables_unpadded(B, OH, OW, H, W, KH, KW, pt, pl, C, dev, pool_order=True)
y = torch.empty(M, N, device=dev)
yp = torch.empty(M // 4, N, device=dev)
arg = torch.empty(B, OH // 2, OW // 2, N, dtype=torch.uint8, device=dev)
dz = torch.rand(M, N, device=dev) - 0.5
dw, db = torch.empty(K, N, device=dev), torch.empty(N, device=dev)
ws = torch.empty(max(F.dense_dw_workspace_floats(M, N, K), 1), device=dev)
top, left = KH - 1 - pt, KW - 1 - pl
rtd, shd, mkd = F.conv_tables_unpadded(B, H, W, OH, OW, KH, KW, top, left, N, dev)
wflip = F.conv_flip_weights(w)
dx = torch.empty(B * H * W, C, device=dev)
ops = {
    "fwd (pooled epilogue)": lambda: F.conv_gemm_forward(x, rt2, sh, w.view(K, N), b, F.ACT_RELU, yp, masktab=mk2, pool_arg=arg),
    "fwd": lambda: F.conv_gemm_forward(x, rt, sh, w.view(K, N), b, F.ACT_RELU, y, masktab=mk),
    "wgrad": lambda: F.conv_gemm_wgrad(x, rt, sh, dz, dw, db, ws, masktab=mk),
    "dgrad": lambda: F.conv_gemm_dgrad(dz.view(B, OH, OW, N), rtd, shd, wflip, dx, masktab=mkd),
}
gf = 2.0 * M * K * N / 1e9
out = []
for name, fn in ops.items():
    if os.environ.get("ONLY") and os.environ["ONLY"] != name:
        continue
    for _ in range(30):
        fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) * 1000 / 20
    out.append("%s %.0f us (%.0f TF/s)" % (name, us, gf / us * 1e3))
print(os.environ.get("LABEL", "default") + ": " + "; ".join(out), flush=True)

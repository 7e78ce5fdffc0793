"""Per-block time of the split-limb forward GEMM (DTF_GEMM_DBG bit 8: every block writes its shader cycles and 100 MHz
ticks over its tile's first outputs; +4 skips the split-and-store phase after the first tile, +16 skips the MFMAs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402

dev = torch.device("cuda")
M, K, N = int(os.environ.get("M", 2048)), int(os.environ.get("K", 16384)), int(os.environ.get("N", 16384))
F.set_gemm_split_products(int(os.environ.get("MODE", 8)))
x = torch.randn(M, K, device=dev)
w = torch.randn(K, N, device=dev) * 0.05
b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
for _ in range(3):
    F.dense_forward(x, w, b, 1, out=y)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
F.dense_forward(x, w, b, 1, out=y)
e.record()
torch.cuda.synchronize()
t = y.view(M // 128, 128, N // 128, 128)[:, 0, :, :2].reshape(-1, 2).double().cpu()
cyc, us = t[:, 0], t[:, 1] / 100.0
kit = (K + 31) // 32
print("DBG=%s kernel %.1f us; blocks %d; block us mean %.1f min %.1f max %.1f; cycles mean %.0f; clock %.0f MHz; per k-tile %.0f cycles (%.2f us)"
      % (os.environ.get("DTF_GEMM_DBG"), a.elapsed_time(e) * 1000, cyc.numel(), us.mean(), us.min(), us.max(), cyc.mean(),
         cyc.sum() / us.sum(), cyc.mean() / kit, us.mean() / kit), flush=True)

"""Per-workgroup main-loop time of the dense forward GEMM (DTF_GEMM_DBG bit 8 makes every block write its loop's
shader-clock cycles and 100 MHz ticks over the first two outputs of its tile).  Run with DTF_GEMM_DBG=8 (full loop)
or 15 (MFMA + LDS reads only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402

dev = torch.device("cuda")
M, K, N = int(os.environ.get("M", 65536)), int(os.environ.get("K", 784)), int(os.environ.get("N", 128))
BM = int(os.environ.get("BM", 128))
BN = int(os.environ.get("BN", 128))
x = torch.rand(M, K, device=dev) * 2 - 1
w = torch.rand(K, N, device=dev) * 2 - 1
b = torch.rand(N, device=dev)
y = torch.empty(M, N, device=dev)
for _ in range(5):
    F.dense_forward(x, w, b, 1, out=y)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
F.dense_forward(x, w, b, 1, out=y)
e.record()
torch.cuda.synchronize()
t = y.view(M // BM, BM, N // BN, BN)[:, 0, :, :5].reshape(-1, 5).double().cpu()
cyc, ticks = t[:, 0], t[:, 1]
us = ticks / 100.0
kit = (K + 31) // 32
print("kernel %.1f us; blocks %d" % (a.elapsed_time(e) * 1000, cyc.numel()))
print("loop us  : mean %.1f min %.1f max %.1f" % (us.mean(), us.min(), us.max()))
print("loop cyc : mean %.0f min %.0f max %.0f  (memtime units)" % (cyc.mean(), cyc.min(), cyc.max()))
print("loop only cyc (before the epilogue): mean %.0f min %.0f max %.0f" % (t[:, 2].mean(), t[:, 2].min(), t[:, 2].max()))
entry, exit_ = t[:, 3], t[:, 4]
e0 = entry.min()
print("block entry (us after the first block): mean %.1f max %.1f; exit: min %.1f mean %.1f max %.1f; entry->loop start mean %.1f us"
      % ((entry - e0).mean() / 100, (entry - e0).max() / 100, (exit_ - e0).min() / 100, (exit_ - e0).mean() / 100,
         (exit_ - e0).max() / 100, ((exit_ - entry) - ticks).mean() / 100))
print("memtime units per us: %.1f" % (cyc.sum() / us.sum()))
print("per k-iter: %.2f us -> at 64 MFMA/wave/iter: %.1f ns per MFMA" % (us.mean() / kit, 1000 * us.mean() / kit / 64))
hist = torch.histc(us.float(), bins=8, min=float(us.min()), max=float(us.max()) + 1e-3)
print("hist", hist.tolist())

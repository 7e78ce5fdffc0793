"""Per-block timing of the split-batch wgrad GEMM (DTF_GEMM_DBG=8: every block writes loop cycles, ticks, loop-only
cycles, entry and exit stamps over the first five outputs of its tile in its slab)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402
from distributed_tensorflow_amd.ops import _native as nv  # noqa: E402

dev = torch.device("cuda")
M, K, N = int(os.environ.get("M", 131072)), int(os.environ.get("K", 784)), int(os.environ.get("N", 128))
BM = int(os.environ.get("BM", 64))
x = torch.rand(M, K, device=dev) * 2 - 1
dz = torch.rand(M, N, device=dev) * 2 - 1
dw = torch.empty(K, N, device=dev)
db = torch.empty(N, device=dev)
splits = nv.kernels().dtf_dense_bwd_dw_splits(M, N, K)
ws = torch.zeros(max(F.dense_dw_workspace_floats(M, N, K), 1), device=dev)
for _ in range(3):
    F.dense_backward_dw(x, dz, dw, db, workspace=ws)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
F.dense_backward_dw(x, dz, dw, db, workspace=ws)
e.record()
torch.cuda.synchronize()
stride = ((K * N + N + 3) // 4) * 4
chunk = (((M + splits - 1) // splits + 31) // 32) * 32
eff = (M + chunk - 1) // chunk
rows = []
tiles = (K + BM - 1) // BM
w = ws.cpu().double()
for s in range(eff):
    for ty in range(tiles):
        o = s * stride + ty * BM * N
        rows.append(w[o:o + 5])
t = torch.stack(rows)
cyc, us, loop = t[:, 0], t[:, 1] / 100.0, t[:, 2]
entry, exit_ = t[:, 3], t[:, 4]
e0 = entry.min()
kit = chunk // 32
print("kernel+reduce %.1f us; splits %d x tiles %d = %d blocks; k-tiles per block %d" % (a.elapsed_time(e) * 1000, eff, tiles, len(rows), kit))
print("block us: mean %.1f min %.1f max %.1f; loop-only cycles mean %.0f (%.0f per k-tile); clock %.0f MHz" % (
    us.mean(), us.min(), us.max(), loop.mean(), loop.mean() / kit, cyc.sum() / us.sum()))
print("entry after first block (us): mean %.1f max %.1f; exit: min %.1f mean %.1f max %.1f; prologue (entry -> loop start) mean %.1f us; epilogue mean %.1f us" % (
    (entry - e0).mean() / 100, (entry - e0).max() / 100, (exit_ - e0).min() / 100, (exit_ - e0).mean() / 100, (exit_ - e0).max() / 100,
    ((exit_ - entry) / 100 - us).mean(), ((cyc - loop) / (cyc.sum() / us.sum())).mean()))
ids = torch.arange(len(rows))
blk = (ids % tiles) + (ids // tiles) * tiles           # launch order: x fastest (1), then y (tile), then z (split)
print("by XCD (block id % 8): " + " ".join("%.0f" % us[(blk % 8) == k].mean() for k in range(8)))
print("by row tile: " + " ".join("%.0f" % us[(ids % tiles) == k].mean() for k in range(tiles)))
zs = ids // tiles
print("by split (every 6th): " + " ".join("%.0f" % us[zs == k].mean() for k in range(0, eff, 6)))
print("hist us: ", torch.histc(us.float(), bins=10, min=float(us.min()), max=float(us.max()) + 1e-3).tolist())
print("loop-only us hist: ", torch.histc((loop / 2062).float(), bins=10).tolist(), float((loop / 2062).min()), float((loop / 2062).max()))

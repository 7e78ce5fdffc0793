"""Per-block timing of the split-batch wgrad GEMM (DTF_GEMM_DBG=8: every block writes loop cycles, ticks, loop-only
cycles, entry and exit stamps, HW_ID and XCC_ID over the first seven outputs of its tile in its slab).  WARM=n runs n
launches first so the clock is the steady one."""
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
for _ in range(int(os.environ.get("WARM", 3))):
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
        rows.append(w[o:o + 7])
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
# placement: which CU ran which blocks, when they started, how much loop work each CU got
hw, xcc = t[:, 5].long(), t[:, 6].long()
cu = (xcc << 16) | (hw & 0xFF00)
per = {}
for i in range(len(rows)):
    per.setdefault(int(cu[i]), []).append(i)
sizes = {}
for v in per.values():
    sizes[len(v)] = sizes.get(len(v), 0) + 1
print("distinct CUs %d; blocks per CU histogram %s" % (len(per), dict(sorted(sizes.items()))))
late = (entry - e0) / 100 > 20
print("blocks entering later than 20 us after the first: %d (entry us: %s)" % (int(late.sum()), sorted(((entry - e0) / 100)[late].tolist())[:12]))
work = torch.tensor([sum(float(loop[i]) for i in v) for v in per.values()])
last = torch.tensor([max(float(exit_[i] - e0) / 100 for i in v) for v in per.values()])
print("per-CU sum of loop cycles: mean %.0f min %.0f max %.0f; per-CU last exit us: mean %.1f min %.1f max %.1f" % (
    work.mean(), work.min(), work.max(), last.mean(), last.min(), last.max()))
for n_on_cu in sorted(sizes):
    sel = [k for k, v in per.items() if len(v) == n_on_cu]
    print("  CUs with %d blocks: last exit mean %.1f us; block us mean %.1f" % (
        n_on_cu, sum(max(float(exit_[i] - e0) / 100 for i in per[k]) for k in sel) / len(sel),
        sum(float(us[i]) for k in sel for i in per[k]) / sum(len(per[k]) for k in sel)))

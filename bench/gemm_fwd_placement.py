"""Where and when the blocks of the first-layer forward GEMM run (DTF_GEMM_DBG=8 stamps: entry / exit time, HW_ID,
XCC_ID per block): blocks per CU, the time every CU goes idle, and the start-time pattern of the rounds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DTF_GEMM_DBG", "8")
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402

dev = torch.device("cuda")
M, K, N, BM = int(os.environ.get("M", 262144)), 784, 128, 128
x = torch.rand(M, K, device=dev) * 2 - 1
w = torch.rand(K, N, device=dev) * 2 - 1
b = torch.rand(N, device=dev)
y = torch.empty(M, N, device=dev)
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = int(os.environ.get("WARM", 200))
for i in range(reps):
    if i == reps - 1:
        a.record()
    F.dense_forward(x, w, b, 1, out=y)
e.record()
torch.cuda.synchronize()
t = y.view(M // BM, BM, 1, N)[:, 0, :, :7].reshape(-1, 7).double().cpu()
us, entry, exit_ = t[:, 1] / 100.0, t[:, 3], t[:, 4]
hw, xcc = t[:, 5].long(), t[:, 6].long()
# the stamps are 20-bit counters of 10 ns ticks (10.5 ms): unwrap relative to the earliest entry
e0 = entry.min()
ent = ((entry - e0) % (1 << 20)) / 100.0
ext = ((exit_ - e0) % (1 << 20)) / 100.0
cu = (xcc << 16) | (hw & 0xFF00)
per = {}
for i in range(t.shape[0]):
    per.setdefault(int(cu[i]), []).append(i)
sizes = {}
for v in per.values():
    sizes[len(v)] = sizes.get(len(v), 0) + 1
print("kernel %.1f us; %d blocks on %d CUs; blocks per CU histogram %s" % (a.elapsed_time(e) * 1000, t.shape[0], len(per), dict(sorted(sizes.items()))))
last = torch.tensor([max(float(ext[i]) for i in v) for v in per.values()])
first = torch.tensor([min(float(ent[i]) for i in v) for v in per.values()])
print("per-CU first entry us: min %.1f max %.1f; per-CU last exit us: mean %.1f min %.1f max %.1f" % (first.min(), first.max(), last.mean(), last.min(), last.max()))
print("last-exit histogram (10 bins from min to max):", torch.histc(last.float(), bins=10).tolist())
print("block duration us: mean %.1f min %.1f max %.1f (main loop %.1f)" % ((ext - ent).mean(), (ext - ent).min(), (ext - ent).max(), us.mean()))
for n_on_cu in sorted(sizes):
    sel = [k for k, v in per.items() if len(v) == n_on_cu]
    print("  CUs with %d blocks: last exit mean %.1f us" % (n_on_cu, sum(max(float(ext[i]) for i in per[k]) for k in sel) / len(sel)))
clock = float(t[:, 0].sum() / us.sum())       # cycles per us
print("clock %.0f MHz; main loop %.1f us, epilogue %.1f us, entry -> loop start %.1f us" % (
    clock, (t[:, 2] / clock).mean(), ((t[:, 0] - t[:, 2]) / clock).mean(), ((ext - ent) - us).mean()))
busy = torch.tensor([sum(float(ext[i] - ent[i]) for i in v) for v in per.values()])
print("per-CU sum of block durations / 2 slots: mean %.1f max %.1f us" % (busy.mean() / 2, busy.max() / 2))
order = torch.argsort(ent)
print("entry times of blocks 0, 511, 512, 1023, 1024, 1535, 1536, 2047 in start order:", [round(float(ent[order[i]]), 1) for i in (0, 511, 512, 1023, 1024, 1535, 1536, min(2047, t.shape[0] - 1))])

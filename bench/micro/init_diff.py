"""Where the device initialiser and its CPU twin differ (debugging aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_tensorflow_amd.ops import functional as F
n = 1_000_003
g = F.init_uniform_(torch.empty(n, device="cuda"), 12345, -0.037, 0.037).cpu()
c = F.init_uniform_(torch.empty(n), 12345, -0.037, 0.037)
d = (g != c).nonzero().flatten()
print("differing", d.numel(), "max abs", float((g - c).abs().max()))
if d.numel():
    i = int(d[0])
    print(i, g[i].item(), c[i].item(), g[i].view(torch.int32).item(), c[i].view(torch.int32).item())
    u = (g + 0.037) / 0.074
    print("first idxs", d[:10].tolist())

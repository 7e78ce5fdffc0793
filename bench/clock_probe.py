import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import functional as F
dev = torch.device("cuda")
M, K, N = 32768, 784, 128
x = torch.rand(M, K, device=dev) * 2 - 1
w = torch.rand(K, N, device=dev) * 2 - 1
y = torch.empty(M, N, device=dev)
for _ in range(50):
    F.dense_forward(x, w, None, 0, out=y)
torch.cuda.synchronize()
yy = y.view(M // 64, 64, N)[:, 0, :2].cpu()
cyc, ticks = yy[:, 0], yy[:, 1]
print("loop cycles median %.0f  min %.0f max %.0f; realtime ticks median %.0f (= %.1f us); clock %.2f GHz; cycles per MFMA per wave %.1f" % (
    cyc.median(), cyc.min(), cyc.max(), ticks.median(), ticks.median() / 100, (cyc.median() / ticks.median()) * 0.1, cyc.median() / 800))

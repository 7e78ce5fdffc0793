"""Main-loop cycles of the first-layer forward GEMM with parts of the loop switched off (DTF_GEMM_DBG bits: 1 no global
loads, 2 no barrier, 4 no LDS stores, 16 no fragment reads; 8 = stamps).  Results are wrong with any of 1..16 set;
this only measures where the loop's cycles go.  One process per setting (the flag is read once)."""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from distributed_tensorflow_amd.ops import functional as F
dev = torch.device("cuda")
M, K, N, BM = 262144, 784, 128, 128
x = torch.rand(M, K, device=dev) * 2 - 1
w = torch.rand(K, N, device=dev) * 2 - 1
b = torch.rand(N, device=dev)
y = torch.empty(M, N, device=dev)
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(200):
    if i == 199: a.record()
    F.dense_forward(x, w, b, 1, out=y)
e.record()
torch.cuda.synchronize()
t = y.view(M // BM, BM, 1, N)[:, 0, :, :5].reshape(-1, 5).double().cpu()
cyc, us = t[:, 0], t[:, 1] / 100.0
print("%%-34s kernel %%6.1f us  loop cycles %%7.0f (ideal 200704)  clock %%4.0f MHz" %% (sys.argv[1], a.elapsed_time(e) * 1000, cyc.mean(), cyc.sum() / us.sum()))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for bits, label in ((8, "full loop"), (9, "no global loads"), (8 + 2, "no barrier"), (8 + 4, "no LDS stores"), (8 + 16, "no fragment reads"),
                    (8 + 1 + 4, "no loads, no stores"), (8 + 1 + 2 + 4, "no loads/stores/barrier"), (8 + 1 + 2 + 4 + 16, "MFMAs only")):
    env = dict(os.environ, DTF_GEMM_DBG=str(bits))
    subprocess.run([sys.executable, "-c", CHILD, label], env=env, check=False)

"""Time of the first-layer forward GEMM (K = 784, N = 128) for a given number of rows M (env M) and forced tile
(DTF_GEMM_TILE), 300th back-to-back launch: does the kernel's last, partly filled round of resident blocks cost more
than a second launch with smaller tiles would?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402

dev = torch.device("cuda")
K, N = 784, 128
Mfull = 262144
x = torch.rand(Mfull, K, device=dev) * 2 - 1
w = torch.rand(K, N, device=dev) * 2 - 1
b = torch.rand(N, device=dev)
y = torch.empty(Mfull, N, device=dev)
for M in [int(v) for v in os.environ.get("M", "262144").split(",")]:
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(300):
        if i == 280:
            a.record()
        F.dense_forward(x[:M], w, b, 1, out=y[:M])
    e.record()
    torch.cuda.synchronize()
    print("tile %s M=%d: %.1f us" % (os.environ.get("DTF_GEMM_TILE", "auto"), M, a.elapsed_time(e) * 1000 / 20), flush=True)

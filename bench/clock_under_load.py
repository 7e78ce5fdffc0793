"""Shader clock of the first-layer forward GEMM under different conditions (DTF_GEMM_DBG=8 stamps): one launch after
idle, the 50th of a back-to-back series, and the same with an operand small enough to stay in the Infinity Cache.
Separates 'the clock has not ramped' from 'this kernel's activity lowers the sustained clock'."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DTF_GEMM_DBG", "8")
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402

dev = torch.device("cuda")
K, N, BM = 784, 128, 128


def probe(M, reps, label, rows_resident=None):
    x = torch.rand(M, K, device=dev) * 2 - 1
    w = torch.rand(K, N, device=dev) * 2 - 1
    b = torch.rand(N, device=dev)
    y = torch.empty(M, N, device=dev)
    torch.cuda.synchronize()
    time.sleep(0.5)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(reps):
        if i == reps - 1:
            a.record()
        F.dense_forward(x, w, b, 1, out=y)
    e.record()
    torch.cuda.synchronize()
    t = y.view(M // BM, BM, 1, N)[:, 0, :, :5].reshape(-1, 5).double().cpu()
    cyc, us = t[:, 0], t[:, 1] / 100.0
    print("%-44s kernel %7.1f us  loop cycles mean %7.0f  clock %5.0f MHz  TFLOP/s %.1f" % (
        label, a.elapsed_time(e) * 1000, cyc.mean(), cyc.sum() / us.sum(), 2.0 * M * K * N / (a.elapsed_time(e) * 1e-3) / 1e12))


probe(262144, 1, "M=262144 (822 MB from HBM), after idle")
probe(262144, 50, "M=262144, 50th back-to-back launch")
probe(262144, 400, "M=262144, 400th back-to-back launch")
probe(32768, 400, "M=32768 (103 MB, cache resident), 400th")
probe(8192, 1600, "M=8192 (26 MB), 1600th")

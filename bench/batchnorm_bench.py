"""Batch-normalisation kernels on feature maps of the MNIST CNN's sizes: time and effective bandwidth (forward reads X
twice and writes Y; backward reads X and dY twice and writes dX)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402

dev = torch.device("cuda", 0)
for R, C in ((16384 * 14 * 14, 32), (16384 * 7 * 7, 64), (262144, 128), (16384, 512)):
    x = torch.randn(R, C, device=dev)
    dy = torch.randn(R, C, device=dev)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    sm, si, dg, db = (torch.empty(C, device=dev) for _ in range(4))
    ws = torch.empty(F.batchnorm_workspace_floats(R, C), device=dev)
    res = []
    for fn, nbytes in ((lambda: F.batchnorm_forward(x, gamma, beta, mm, mv, 0.99, 1e-3, True, y, sm, si, ws), 3 * R * C * 4),
                       (lambda: F.batchnorm_backward(x, dy, gamma, sm, si, dg, db, dx, ws), 5 * R * C * 4)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        res.append("%8.1f us %5.2f TB/s" % (dt * 1e6, nbytes / dt / 1e12))
    print("[%9d x %4d]  forward %s   backward %s" % (R, C, res[0], res[1]), flush=True)

#!/usr/bin/env python
"""Achieved HBM bandwidth of the flat Adam apply kernel (28 bytes per parameter) over a range of sizes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import functional as F  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    for n in (1 << 17, 1 << 19, 1664000, 1 << 22, 1 << 24, 1 << 26, 1 << 28):
        w, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(3))
        v.abs_()
        g = torch.randn(n, device=dev) * 1e-3
        for _ in range(3):
            F.adam_apply(w, m, v, g, step)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        a.record()
        for _ in range(iters):
            F.adam_apply(w, m, v, g, step)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / iters * 1000
        print("n=%10d  %8.1f us  %6.2f TB/s" % (n, us, 28.0 * n / us / 1e6))


if __name__ == "__main__":
    main()

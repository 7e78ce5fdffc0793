#!/usr/bin/env python
"""Time of the fused network-tail kernel against the chain of layer-wise kernels it replaces."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_tail_gpu import _inputs, _run_layerwise, _run_tail  # noqa: E402


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1000.0


def main():
    dev = torch.device("cuda", 0)
    step = torch.tensor([3], dtype=torch.int64, device=dev)
    sizes = [int(s) for s in os.environ.get("TAIL_M", "2048,8192,32768,131072").split(",")]
    iters = int(os.environ.get("TAIL_ITERS", "50"))
    for M in sizes:
        args = _inputs(M, 10, dev)
        t_tail = timed(lambda: _run_tail(*args, 0.7, 77, step, 1.0 / 0.7), iters)
        t_layer = timed(lambda: _run_layerwise(*args, 0.7, 77, step, 1.0 / 0.7), iters)
        flops = 2.0 * M * (128 * 128 * 3 + 128 * 10 * 3)
        print("M=%6d  fused tail %7.1f us (%5.1f TFLOP/s)   layer-wise chain %7.1f us" %
              (M, t_tail, flops / t_tail / 1e6, t_layer))


if __name__ == "__main__":
    main()

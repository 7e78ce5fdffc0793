#!/usr/bin/env python
"""Phase timeline of the network-tail kernel: shader-clock stamps of workgroup 0's first 64-row slab."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import _native  # noqa: E402
from tests.test_tail_gpu import _inputs, _run_tail  # noqa: E402

NAMES = ["h1 slab -> LDS (+barrier)", "h2 = relu(h1 W2) + dropout (+barrier)", "logits (16x16 tiles, all waves) (+barrier)",
         "softmax / dz3 (+barrier)", "dW3, dz2 products, db3 (+barrier)", "dz2 mask in place (+barrier)",
         "dW2 + db2", "dz1 products + store (+barrier)"]


def main():
    dev = torch.device("cuda", 0)
    M = int(os.environ.get("TAIL_M", "65536"))
    lib = _native.kernels()
    lib.dtf_mlp_tail_set_profile_buffer.argtypes = [C.c_void_p]
    prof = torch.zeros(32, dtype=torch.int64, device=dev)
    step = torch.tensor([3], dtype=torch.int64, device=dev)
    args = _inputs(M, 10, dev)
    keep = float(os.environ.get("TAIL_KEEP", "0.7"))       # 1.0: no dropout hashing in the h2 phase
    for _ in range(int(os.environ.get("TAIL_WARM", "3"))):
        _run_tail(*args, keep, 77, step, 1.0 / 0.7)
    lib.dtf_mlp_tail_set_profile_buffer(C.c_void_p(prof.data_ptr()))
    _run_tail(*args, keep, 77, step, 1.0 / 0.7)
    torch.cuda.synchronize()
    lib.dtf_mlp_tail_set_profile_buffer(None)
    t = prof.cpu().tolist()
    total = t[8] - t[0]
    print("keep %.1f " % keep, end="")
    print("M=%d: second slab of workgroup 0 = %d shader-clock ticks" % (M, total))
    for i, name in enumerate(NAMES):
        d = t[i + 1] - t[i]
        print("  %-44s %7d ticks  %5.1f %%" % (name, d, 100.0 * d / total))
    print("  SIMD of waves 0..7 of workgroup 0:", t[16:24])
    print("  end of the h2 products per wave (ticks after the phase start):", [t[24 + w] - t[1] for w in range(8)])
    print("  inside the h2 phase (wave 0): products %d, epilogue %d, barrier wait %d ticks" % (t[9] - t[1], t[10] - t[9], t[2] - t[10]))


if __name__ == "__main__":
    main()

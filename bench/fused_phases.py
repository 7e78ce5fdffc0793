#!/usr/bin/env python
"""Phase timeline of the one-launch training step (csrc/kernels/mlp_fused.hip): shader-clock timestamps taken by
the kernel at its phase boundaries, printed as cycles per phase for the last step of a multi-step launch."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.data import get_data  # noqa: E402
from distributed_tensorflow_amd.models import xor_mlp  # noqa: E402
from distributed_tensorflow_amd.ops import _native  # noqa: E402

NAMES = ["batch->LDS", "L1 fwd", "L2 chunk0 (+commit)", "L2 chunk1 + epilogue", "L3 fwd", "loss", "dW3 + dz2 products",
         "dz2 epilogue", "apply W3,b3 | dW2 (held) + b2", "dz1 (2 slices)", "apply W2 | dW1 + apply", "b1"]


def main():
    dev = torch.device("cuda", 0)
    B, steps = int(os.environ.get("B", "50")), 200
    m = xor_mlp(dropout=float(os.environ.get("DROPOUT", "0.3")))
    m.compile(loss="mean_squared_error", optimizer=os.environ.get("OPT", "adam"), metrics=["binary_rounded_accuracy"],
              device=dev, seed=0)
    x, y = get_data(B * steps, val_set_size=0, seed=1, device=dev)[:2]
    prof = torch.zeros(16, dtype=torch.int64, device=dev)
    lib = _native.kernels()
    lib.dtf_mlp3_fused_set_profile_buffer.argtypes = [C.c_void_p]
    lib.dtf_mlp3_fused_set_profile_buffer(prof.data_ptr())
    m.train_steps_fused(x, y, B, steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.train_steps_fused(x, y, B, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.dtf_mlp3_fused_set_profile_buffer(None)
    t = prof.tolist()
    total = t[12] - t[0]
    print("dropout=%s opt=%s" % (os.environ.get("DROPOUT", "0.3"), os.environ.get("OPT", "adam")))
    print("B=%d  %.2f us/step wall (%d steps/launch); last step = %d shader-clock ticks" % (B, 1e6 * dt / steps, steps, total))
    for i, name in enumerate(NAMES):
        d = t[i + 1] - t[i]
        print("  %-24s %8d ticks  %5.1f %%" % (name, d, 100.0 * d / max(total, 1)))


if __name__ == "__main__":
    main()

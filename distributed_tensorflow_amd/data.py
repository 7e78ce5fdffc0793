"""Synthetic datasets.

``get_data`` reproduces the reference generator (/root/reference/example.py:24-48): ``n +
val_set_size`` samples of ``2*bits`` random bits, label bit k = x[k] XOR x[k+bits], the
first ``n`` samples for training and the rest for validation.  The reference draws with
an unseeded ``random.randint`` into Python lists; here the draw is vectorised, seedable
and can be produced directly on the GPU.

``mnist_like`` makes MNIST-shaped data (784 floats in [0,1), 10 classes) whose labels come
class-conditionally: ten fixed random "stroke" prototypes, mixed with a distractor class and uniform pixel
noise, so "test accuracy" on held-out samples is a meaningful number.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .ops import functional as F

BITS = 32
VAL_SET_SIZE = 1000


def get_data(n: int, bits: int = BITS, val_set_size: int = VAL_SET_SIZE, seed: Optional[int] = None,
             device=None, as_lists: bool = False):
    """Return ``x_train, y_train, x_val, y_val`` for the bit-wise XOR task."""
    total = n + val_set_size
    if seed is None:
        seed = int(torch.seed() & 0x7FFFFFFF)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    x, y = F.xor_data(total, bits, seed, dev)
    out = (x[:n], y[:n], x[n:], y[n:])
    if as_lists:
        return tuple(t.to(torch.int64).tolist() for t in out)
    return out


def mnist_like(n: int, seed: int = 0, device=None, num_classes: int = 10, input_dim: int = 784,
               one_hot: bool = False, teacher_seed: int = 1234) -> Tuple[torch.Tensor, torch.Tensor]:
    """``n`` synthetic MNIST-shaped examples: each image is a noisy mixture of its label's prototype, a
    distractor prototype and uniform pixel noise, clipped to [0, 1]."""
    dev = torch.device(device) if device is not None else torch.device("cpu")
    g = torch.Generator(device="cpu").manual_seed(int(teacher_seed))
    protos = (torch.rand(num_classes, input_dim, generator=g) < 0.19).float()   # fixed "stroke" masks, ~19 % ink
    gx = torch.Generator(device="cpu").manual_seed(int(seed) + 17)
    y = torch.randint(0, num_classes, (n,), generator=gx)
    other = torch.randint(0, num_classes, (n,), generator=gx)
    mix = torch.rand(n, 1, generator=gx) * 0.45                 # up to 45 % of a distractor class bleeds in
    x = (1.0 - mix) * protos[y] + mix * protos[other]
    x = (0.55 * x + 0.45 * torch.rand(n, input_dim, generator=gx)).clamp_(0.0, 1.0)
    x, y = x.to(dev), y.to(dev)
    if one_hot:
        y = torch.nn.functional.one_hot(y, num_classes).to(torch.float32)
    return x, y

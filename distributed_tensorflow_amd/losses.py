"""Loss registry (string-keyed like ``model.compile(loss='mean_squared_error')`` at
/root/reference/example2.py:165).

A loss object turns (model output, target) into the gradient that enters the backward
chain and adds (loss sum, accuracy sum, example count) to a device-resident statistics
record, all in one kernel launch.
"""
from __future__ import annotations

import torch

from .ops import functional as F
from .ops._native import ACT_SIGMOID


class Loss:
    name = "loss"
    wants_logits = False          # softmax head should hand over logits
    primary_metric = "binary_rounded"   # which accuracy the fused kernel reports in stats[1]

    def fuse_with_head(self, head) -> bool:
        """Return True if the gradient this loss emits is dL/dz of ``head`` (activation gradient folded in)."""
        return False

    def __call__(self, y_pred, y_true, grad_out, stats, grad_scale=None):
        raise NotImplementedError


class _ElementwiseLoss(Loss):
    """Losses over probabilities, element by element: behind a sigmoid head the kernel emits dL/dz directly."""

    def __init__(self):
        self.through_sigmoid = False

    def fuse_with_head(self, head) -> bool:
        self.through_sigmoid = getattr(head, "act", None) == ACT_SIGMOID and not getattr(head, "softmax", False) \
            and getattr(head, "fused_keep", 1.0) >= 1.0
        return self.through_sigmoid


class MeanSquaredError(_ElementwiseLoss):
    """Keras ``mean_squared_error``: mean over the last axis, then over the batch
    (/root/reference/example.py:163)."""

    name = "mean_squared_error"

    def __call__(self, y_pred, y_true, grad_out, stats, grad_scale=None):
        return F.mse_loss(y_pred, y_true, grad_out, grad_scale, self.through_sigmoid, stats)


class BinaryCrossentropy(_ElementwiseLoss):
    """Keras ``binary_crossentropy`` on sigmoid outputs (probabilities clipped to ``[1e-7, 1 - 1e-7]``, mean over the
    last axis, then over the batch); with a sigmoid head the gradient handed back is dL/dz = p - y."""

    name = "binary_crossentropy"

    def __call__(self, y_pred, y_true, grad_out, stats, grad_scale=None):
        return F.bce_loss(y_pred, y_true, grad_out, grad_scale, self.through_sigmoid, stats)


class CategoricalCrossentropy(Loss):
    """Softmax cross-entropy on one-hot (or probability) targets; consumes logits."""

    name = "categorical_crossentropy"
    wants_logits = True
    primary_metric = "categorical"

    def fuse_with_head(self, head) -> bool:
        return True

    def __call__(self, y_pred, y_true, grad_out, stats, grad_scale=None):
        if y_true.dtype != torch.float32:
            y_true = y_true.to(torch.float32)
        return F.softmax_xent(y_pred, y_true.contiguous(), grad_out, grad_scale, stats)


class SparseCategoricalCrossentropy(Loss):
    """Softmax cross-entropy on int64 class ids; consumes logits."""

    name = "sparse_categorical_crossentropy"
    wants_logits = True
    primary_metric = "categorical"

    def fuse_with_head(self, head) -> bool:
        return True

    def __call__(self, y_pred, y_true, grad_out, stats, grad_scale=None):
        if y_true.dtype != torch.int64:
            y_true = y_true.to(torch.int64)
        return F.softmax_xent(y_pred, y_true.contiguous().view(-1), grad_out, grad_scale, stats)


_REGISTRY = {
    "mean_squared_error": MeanSquaredError,
    "mse": MeanSquaredError,
    "binary_crossentropy": BinaryCrossentropy,
    "categorical_crossentropy": CategoricalCrossentropy,
    "softmax_cross_entropy": CategoricalCrossentropy,
    "sparse_categorical_crossentropy": SparseCategoricalCrossentropy,
    "sparse_softmax_cross_entropy": SparseCategoricalCrossentropy,
}


def get(spec) -> Loss:
    if isinstance(spec, Loss):
        return spec
    if isinstance(spec, type) and issubclass(spec, Loss):
        return spec()
    if spec in _REGISTRY:
        return _REGISTRY[spec]()
    raise ValueError("unknown loss %r (have: %s)" % (spec, ", ".join(sorted(_REGISTRY))))


def binary_crossentropy(y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    """Functional Keras form: per-example binary cross-entropy over the last axis."""
    pc = y_pred.clamp(F.BCE_EPS, 1.0 - F.BCE_EPS)
    return (-(y_true * pc.log() + (1.0 - y_true) * (1.0 - pc).log())).mean(dim=-1)


def mean_squared_error(y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    """Functional Keras form: per-example MSE over the last axis (``losses.mean_squared_error(labels, preds)``)."""
    return ((y_pred - y_true) ** 2).mean(dim=-1)

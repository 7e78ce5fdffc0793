"""``torch.autograd.Function`` wrappers over the fused kernels, and ``torch.nn`` modules on top of them.

TensorFlow differentiates whatever graph the script builds (``tf.gradients`` behind ``minimize``,
/root/reference/example.py:146-170).  The layer classes of this package carry hand-written backward passes for
sequential chains; these wrappers put the same kernels under PyTorch's reverse-mode engine, so they serve any
graph - skip connections, several inputs, summed losses - and any ordinary ``torch.nn.Module``:

* forward / backward call :mod:`.functional` and nothing else: fused GEMM + bias + activation (+ inverted dropout)
  forward, activation gradient, data-gradient and weight-gradient GEMMs with the bias gradient folded in, the fused
  loss + gradient kernels, im2col / col2im convolution, max pooling;
* GPU tensors run the HIP kernels (a missing kernel library is an error), CPU tensors the reference implementations;
* everything is float32 and NHWC, like the rest of the package.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import functional as F
from ._native import ACT_CODES, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t if t.is_contiguous() else t.contiguous()


class _DenseFn(torch.autograd.Function):
    """``y = dropout(act(x @ w + b))`` in one kernel; backward = activation(+dropout) gradient, dx and dw/db GEMMs."""

    @staticmethod
    def forward(ctx, x, w, b, act, keep, seed, step):
        x, w, b = _c(x), _c(w), _c(b)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        if act in (ACT_SIGMOID, ACT_TANH) and keep < 1.0:
            raise ValueError("dropout fused behind a sigmoid / tanh is not supported: apply `dropout` as a separate op")
        y = F.dense_forward(x2, w, b, act=act, keep=keep, seed=seed, step=step)
        ctx.save_for_backward(x2, w, y)
        ctx.cfg = (act, keep, seed, step, b is not None, lead)
        return y.reshape(*lead, w.shape[1])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        act, keep, seed, step, has_bias, lead = ctx.cfg
        dy2 = _c(dy).reshape(-1, w.shape[1])
        if act == ACT_NONE:
            dz = F.dropout(dy2, keep, seed, step) if keep < 1.0 else dy2     # the same mask as in the forward pass
        else:
            # relu (+dropout): (y > 0) / keep - a dropped element is zero in y; sigmoid: y (1 - y); tanh: 1 - y^2
            dz = F.act_grad(dy2, y, act, scale=1.0 / keep)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = F.dense_backward_dx(dz, w).reshape(*lead, w.shape[0])
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(w.shape[1], dtype=torch.float32, device=w.device) if has_bias else None
            F.dense_backward_dw(x2, dz, dw, db)
        return dx, dw, db, None, None, None, None


def dense(x, w, b=None, activation=None, dropout: float = 0.0, seed: int = 0, step=None):
    """Differentiable fused dense layer.  ``dropout`` is the drop rate applied after the activation (inverted
    dropout; the mask is a pure function of ``seed``, the int64 device tensor ``step`` and the element index)."""
    act = ACT_CODES[activation] if not isinstance(activation, int) else activation
    return _DenseFn.apply(x, w, b, act, 1.0 - float(dropout), int(seed), step)


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep, seed, step):
        ctx.cfg = (keep, seed, step)
        return F.dropout(_c(x), keep, seed, step)

    @staticmethod
    def backward(ctx, dy):
        keep, seed, step = ctx.cfg
        return F.dropout(_c(dy), keep, seed, step), None, None, None


def dropout(x, rate: float, seed: int = 0, step=None, training: bool = True):
    if not training or rate <= 0.0:
        return x
    return _DropoutFn.apply(x, 1.0 - float(rate), int(seed), step)


class _ActivationFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        y = F.activation_forward(_c(x), act)
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return F.act_grad(_c(dy), y, ctx.act), None


def activation(x, name):
    act = ACT_CODES[name]
    return x if act == ACT_NONE else _ActivationFn.apply(x, act)


class _SoftmaxXentFn(torch.autograd.Function):
    """Mean softmax cross-entropy of logits ``z`` [M, C] against int64 class ids or a float distribution; the
    gradient comes out of the same kernel as the loss."""

    @staticmethod
    def forward(ctx, z, labels):
        z = _c(z)
        labels = labels.contiguous() if labels.dtype == torch.int64 else _c(labels)
        g = torch.empty_like(z)
        st = F.new_stats(z.device)
        F.softmax_xent(z, labels, g, 1.0 / z.shape[0], st)
        ctx.save_for_backward(g)
        ctx.stats = st
        return st[0] / z.shape[0]

    @staticmethod
    def backward(ctx, dl):
        (g,) = ctx.saved_tensors
        return g * dl, None


def softmax_cross_entropy(logits, labels):
    return _SoftmaxXentFn.apply(logits, labels)


class _MseFn(torch.autograd.Function):
    """Keras ``mean_squared_error`` averaged over the batch: mean over all elements of ``(p - y)^2``."""

    @staticmethod
    def forward(ctx, p, y):
        p, y = _c(p), _c(y)
        p2, y2 = p.reshape(p.shape[0], -1), y.reshape(y.shape[0], -1)
        g = torch.empty_like(p2)
        st = F.new_stats(p.device)
        F.mse_loss(p2, y2, g, None, False, st)
        ctx.save_for_backward(g)
        ctx.shape = p.shape
        return st[0] / p2.shape[0]

    @staticmethod
    def backward(ctx, dl):
        (g,) = ctx.saved_tensors
        return (g * dl).reshape(ctx.shape), None


def mean_squared_error(pred, target):
    return _MseFn.apply(pred, target)


class _BceFn(torch.autograd.Function):
    """Keras ``binary_crossentropy`` averaged over the batch (probabilities clipped to ``[1e-7, 1 - 1e-7]``)."""

    @staticmethod
    def forward(ctx, p, y):
        p, y = _c(p), _c(y)
        p2, y2 = p.reshape(p.shape[0], -1), y.reshape(y.shape[0], -1)
        g = torch.empty_like(p2)
        st = F.new_stats(p.device)
        F.bce_loss(p2, y2, g, None, False, st)
        ctx.save_for_backward(g)
        ctx.shape = p.shape
        return st[0] / p2.shape[0]

    @staticmethod
    def backward(ctx, dl):
        (g,) = ctx.saved_tensors
        return (g * dl).reshape(ctx.shape), None


def binary_crossentropy(pred, target):
    return _BceFn.apply(pred, target)


class _Conv2dFn(torch.autograd.Function):
    """NHWC convolution with a TensorFlow filter ``[KH, KW, C, N]``: patch matrix x filter matrix on the dense GEMM
    kernels (forward, weight gradient, data gradient + col2im), bias and activation fused."""

    @staticmethod
    def forward(ctx, x, w, b, strides, padding, act):
        x, w, b = _c(x), _c(w), _c(b)
        KH, KW, C, N = w.shape
        B, H, W_, _ = x.shape
        OH, OW, _, _ = F.conv_geometry(H, W_, KH, KW, strides, padding)[:4]
        cols = F.im2col(x, KH, KW, strides, padding)
        w2d = w.reshape(KH * KW * C, N)
        y = F.dense_forward(cols, w2d, b, act=act)
        ctx.save_for_backward(cols, w2d, y)
        ctx.cfg = (tuple(x.shape), (KH, KW, C, N), strides, padding, act, b is not None)
        return y.reshape(B, OH, OW, N)

    @staticmethod
    def backward(ctx, dy):
        cols, w2d, y = ctx.saved_tensors
        xshape, (KH, KW, C, N), strides, padding, act, has_bias = ctx.cfg
        dy2 = _c(dy).reshape(-1, N)
        dz = dy2 if act == ACT_NONE else F.act_grad(dy2, y, act)
        dx = dw = db = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dw2d = torch.empty_like(w2d)
            db = torch.empty(N, dtype=torch.float32, device=w2d.device) if has_bias else None
            F.dense_backward_dw(cols, dz, dw2d, db)
            dw = dw2d.reshape(KH, KW, C, N)
        if ctx.needs_input_grad[0]:
            dcols = F.dense_backward_dx(dz, w2d)
            dx = F.col2im(dcols, xshape, KH, KW, strides, padding)
        return dx, dw, db, None, None, None


def conv2d(x, w, b=None, strides=(1, 1), padding="valid", activation=None):
    return _Conv2dFn.apply(x, w, b, tuple(strides), padding, ACT_CODES[activation])


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pool, strides, padding):
        x = _c(x)
        tiling = strides == pool and padding == "valid"
        if tiling:
            y, arg = F.maxpool_forward(x, pool)
        else:
            y, arg = F.maxpool_strided_forward(x, pool, strides, padding)
        ctx.save_for_backward(arg)
        ctx.cfg = (tuple(x.shape), pool, strides, padding, tiling)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        xshape, pool, strides, padding, tiling = ctx.cfg
        if tiling:
            return F.maxpool_backward(_c(dy), arg, xshape, pool), None, None, None
        return F.maxpool_strided_backward(_c(dy), arg, xshape, pool, strides, padding), None, None, None


def max_pool2d(x, pool=(2, 2), strides=None, padding="valid"):
    """Max pooling of an NHWC tensor: any window and strides (default: strides = window), VALID or SAME padding."""
    pool = tuple(pool)
    return _MaxPoolFn.apply(x, pool, tuple(strides) if strides is not None else pool, str(padding).lower())


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pool, strides, padding):
        x = _c(x)
        ctx.cfg = (tuple(x.shape), pool, strides, padding)
        return F.avgpool_forward(x, pool, strides, padding)

    @staticmethod
    def backward(ctx, dy):
        xshape, pool, strides, padding = ctx.cfg
        return F.avgpool_backward(_c(dy), xshape, pool, strides, padding), None, None, None


def avg_pool2d(x, pool=(2, 2), strides=None, padding="valid"):
    """Average pooling of an NHWC tensor (windows over the border average over their in-image positions)."""
    pool = tuple(pool)
    return _AvgPoolFn.apply(x, pool, tuple(strides) if strides is not None else pool, str(padding).lower())


class _BatchNormFn(torch.autograd.Function):
    """Batch normalisation over the last axis (Keras semantics); ``moving_mean`` / ``moving_var`` are updated in place
    while training.  Gradients flow to ``x``, ``gamma`` and ``beta``."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, momentum, eps, training):
        x = _c(x)
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        R = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        inv = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty(F.batchnorm_workspace_floats(R, C), dtype=torch.float32, device=x.device) if x.is_cuda else None
        F.batchnorm_forward(x2, _c(gamma), _c(beta), moving_mean, moving_var, momentum, eps, training, y, mean, inv, ws)
        if not training:
            inv = 1.0 / torch.sqrt(moving_var + eps)
            mean = moving_mean.clone()
        ctx.save_for_backward(x2, gamma if gamma is not None else torch.empty(0), mean, inv)
        ctx.cfg = (tuple(x.shape), gamma is not None, beta is not None, bool(training), ws)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, mean, inv = ctx.saved_tensors
        shape, has_gamma, has_beta, training, ws = ctx.cfg
        dy2 = _c(dy).reshape(x2.shape)
        g = _c(gamma) if has_gamma else None
        if not training:       # the affine map of the moving statistics
            xhat = (x2 - mean) * inv
            dx = dy2 * (inv if g is None else g * inv)
            return (dx.view(shape), (dy2 * xhat).sum(0) if has_gamma else None, dy2.sum(0) if has_beta else None,
                    None, None, None, None, None)
        C = x2.shape[1]
        dgamma = torch.empty(C, dtype=torch.float32, device=x2.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x2.device)
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        F.batchnorm_backward(x2, dy2, g, mean, inv, dgamma, dbeta, dx, ws)
        return (dx.view(shape) if dx is not None else None, dgamma if has_gamma else None, dbeta if has_beta else None,
                None, None, None, None, None)


def batch_norm(x, gamma, beta, moving_mean, moving_var, momentum: float = 0.99, eps: float = 1e-3, training: bool = True):
    """Differentiable batch normalisation over the last axis of ``x`` (``gamma`` / ``beta`` may be None)."""
    return _BatchNormFn.apply(x, gamma, beta, moving_mean, moving_var, float(momentum), float(eps), bool(training))


# ----------------------------------------------------------------------------------------------- torch.nn modules
class Dense(torch.nn.Module):
    """``torch.nn.Module`` whose forward and backward are the fused HIP dense kernels (Glorot-uniform kernel
    ``[in, out]``, zero bias - the Keras defaults of /root/reference/example.py:150-154)."""

    def __init__(self, in_features: int, units: int, activation: Optional[str] = None, use_bias: bool = True,
                 dropout: float = 0.0, seed: int = 0):
        super().__init__()
        limit = math.sqrt(6.0 / (in_features + units))
        self.kernel = torch.nn.Parameter((torch.rand(in_features, units) * 2.0 - 1.0) * limit)
        self.bias = torch.nn.Parameter(torch.zeros(units)) if use_bias else None
        self.activation, self.dropout, self.seed = activation, float(dropout), int(seed)
        self.register_buffer("step", torch.zeros(1, dtype=torch.int64))       # advance it once per training step

    def forward(self, x):
        rate = self.dropout if self.training else 0.0
        return dense(x, self.kernel, self.bias, self.activation, rate, self.seed, self.step if rate > 0.0 else None)


class Conv2D(torch.nn.Module):
    def __init__(self, in_channels: int, filters: int, kernel_size=(3, 3), strides=(1, 1), padding: str = "valid",
                 activation: Optional[str] = None, use_bias: bool = True):
        super().__init__()
        kh, kw = kernel_size
        limit = math.sqrt(6.0 / (kh * kw * (in_channels + filters)))
        self.kernel = torch.nn.Parameter((torch.rand(kh, kw, in_channels, filters) * 2.0 - 1.0) * limit)
        self.bias = torch.nn.Parameter(torch.zeros(filters)) if use_bias else None
        self.strides, self.padding, self.activation = tuple(strides), padding, activation

    def forward(self, x):
        return conv2d(x, self.kernel, self.bias, self.strides, self.padding, self.activation)


class MaxPooling2D(torch.nn.Module):
    def __init__(self, pool_size=(2, 2), strides=None, padding="valid"):
        super().__init__()
        self.pool_size, self.strides, self.padding = tuple(pool_size), strides, padding

    def forward(self, x):
        return max_pool2d(x, self.pool_size, self.strides, self.padding)


class BatchNormalization(torch.nn.Module):
    """Keras-style batch normalisation over the last axis on the HIP kernels (``moving_*`` follow the batch statistics
    with ``momentum`` while ``self.training``)."""

    def __init__(self, channels: int, momentum: float = 0.99, epsilon: float = 1e-3, center: bool = True, scale: bool = True):
        super().__init__()
        self.gamma = torch.nn.Parameter(torch.ones(channels)) if scale else None
        self.beta = torch.nn.Parameter(torch.zeros(channels)) if center else None
        self.register_buffer("moving_mean", torch.zeros(channels))
        self.register_buffer("moving_variance", torch.ones(channels))
        self.momentum, self.epsilon = float(momentum), float(epsilon)

    def forward(self, x):
        return batch_norm(x, self.gamma, self.beta, self.moving_mean, self.moving_variance, self.momentum, self.epsilon,
                          self.training)

"""Functional front-end of the hot ops.

Each function dispatches on the device of its tensors:

* GPU tensors -> the hand-written HIP kernels in ``csrc/kernels`` (launched on the
  current PyTorch stream, so they are captured by ``torch.cuda.graph``);
* CPU tensors -> a plain-PyTorch fp32 reference of the same math.  The reference is
  what the multi-process gloo tests run and what the GPU numerics tests compare to.

The ops mirror what one ``sess.run([accuracy, loss, summ, train_step])`` executes in
/root/reference/example.py:213 (dense forward, dropout, MSE, accuracy, their gradients
and the Adam apply).
"""
from __future__ import annotations

import torch

from . import _native as nv
from ._native import ACT_CODES, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH  # noqa: F401

_M32 = 0xFFFFFFFF


# --------------------------------------------------------------------------- RNG (CPU mirror of common.h)
def _mix32(x: torch.Tensor) -> torch.Tensor:
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def _mix32_int(x: int) -> int:
    x &= _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


def dropout_key(seed: int, step: int) -> int:
    k = _mix32_int((step & _M32) + 0x9E3779B9)
    k = _mix32_int(k ^ (((step >> 32) & _M32) * 0x85EBCA6B & _M32) ^ (seed & _M32))
    return k


def dropout_uniform(key: int, n: int, device="cpu") -> torch.Tensor:
    """u[i] in [0,1) for flat element index i: bit-identical to ``dtf_uniform`` on the GPU."""
    idx = torch.arange(n, dtype=torch.int64, device=device)
    hi = (idx >> 32) & _M32
    r = _mix32((((idx & _M32) * 0x9E3779B1 + key) & _M32) ^ (((hi << 16) | (hi >> 16)) & _M32))
    return (r >> 8).to(torch.float32) * (1.0 / 16777216.0)


def _step_value(step) -> int:
    if step is None:
        return 0
    if isinstance(step, torch.Tensor):
        return int(step.item())
    return int(step)


# --------------------------------------------------------------------------- dense layer
def _apply_act(z: torch.Tensor, act: int) -> torch.Tensor:
    if act == ACT_RELU:
        return torch.relu(z)
    if act == ACT_SIGMOID:
        return torch.sigmoid(z)
    if act == ACT_TANH:
        return torch.tanh(z)
    return z


def dense_bf16_supported(M: int, N: int, K: int) -> bool:
    """True when all three products of a [M,K] x [K,N] dense layer can run on the bf16-operand MFMA kernels (wide
    layers: at least 128 output tiles of 128x128 each way)."""
    return nv.kernels_available() and bool(nv.kernels().dtf_dense_bf16_supported(M, N, K))


def set_gemm_split_products(n: int) -> None:
    """Process-wide switch of the fp32 dense GEMMs: 0 (default) multiplies on the fp32 MFMA; 6, 8 or 9 splits every
    fp32 operand exactly into three bf16 limbs and accumulates that many limb products in fp32 on the bf16 matrix
    cores (``gemm_x3_kernel`` in csrc/kernels/gemm_f32.hip; 8 drops only the 2^-32 term).  Set it before the first
    step: wgrad workspace sizes depend on it.  No effect on the CPU reference path."""
    if nv.kernels_available():
        nv.kernels().dtf_gemm_set_split_products(int(n))


def gemm_split_products() -> int:
    return int(nv.kernels().dtf_gemm_get_split_products()) if nv.kernels_available() else 0


def _bf16_round(t: torch.Tensor) -> torch.Tensor:
    """fp32 values rounded to bf16 (nearest even), as the bf16-operand kernels round their operands."""
    return t.to(torch.bfloat16).to(torch.float32)


def dense_fwd_workspace_floats(M: int, N: int, K: int) -> int:
    """Floats of scratch with which ``dense_forward`` may split a long K (few output tiles); 0 = never splits."""
    if not nv.kernels_available():
        return 0
    s = nv.kernels().dtf_dense_fwd_splits(M, N, K)
    return s * M * N if s > 1 else 0


def dense_forward(x, w, b=None, act=ACT_NONE, keep=1.0, seed=0, step=None, out=None, workspace=None, bf16=False):
    """``dropout(act(x @ w + b))`` with inverted dropout (keep probability ``keep``).

    ``bf16=True``: both operands are rounded to bf16 for the product (fp32 accumulation, fp32 everything else)."""
    M, K = x.shape
    K2, N = w.shape
    assert K == K2, "x[%d,%d] @ w[%d,%d]" % (M, K, K2, N)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    if x.is_cuda:
        nv.require_f32_contig(x, w, b, out)
        if step is not None:
            assert step.dtype == torch.int64 and step.is_cuda
        if act == ACT_TANH:
            # the GEMM epilogues fuse relu and sigmoid; tanh is one more pass over the output, in place
            if keep < 1.0:
                raise ValueError("dropout is fused into relu layers only")
            dense_forward(x, w, b, ACT_NONE, 1.0, seed, step, out, workspace, bf16)
            return activation_forward(out, ACT_TANH, out=out)
        if bf16:
            rc = nv.kernels().dtf_dense_fwd_bf16(nv.ptr(x), nv.ptr(w), nv.ptr(b), nv.ptr(out), M, N, K, int(act),
                                                 float(keep), int(seed) & _M32, nv.ptr(step), nv.stream_ptr())
        elif workspace is not None:
            rc = nv.kernels().dtf_dense_fwd_ws(nv.ptr(x), nv.ptr(w), nv.ptr(b), nv.ptr(out), M, N, K, int(act),
                                               float(keep), int(seed) & _M32, nv.ptr(step), nv.ptr(workspace),
                                               workspace.numel(), nv.stream_ptr())
        else:
            rc = nv.kernels().dtf_dense_fwd(nv.ptr(x), nv.ptr(w), nv.ptr(b), nv.ptr(out), M, N, K, int(act),
                                            float(keep), int(seed) & _M32, nv.ptr(step), nv.stream_ptr())
        nv.check(rc, "dtf_dense_fwd")
        return out
    z = _bf16_round(x) @ _bf16_round(w) if bf16 else x @ w
    if b is not None:
        z = z + b
    y = _apply_act(z, act)
    if keep < 1.0:
        u = dropout_uniform(dropout_key(seed, _step_value(step)), M * N).view(M, N)
        y = torch.where(u < keep, y * (1.0 / keep), torch.zeros_like(y))
    out.copy_(y)
    return out


def dense_backward_dx(dz, w, yprev=None, act_prev=ACT_NONE, scale_prev=1.0, out=None, bf16=False):
    """``dz @ w.T`` followed by the producing layer's activation/dropout gradient.

    ``yprev`` is the producing layer's OUTPUT (this layer's input).  relu(+dropout):
    ``(yprev > 0) * scale_prev``; sigmoid: ``yprev * (1 - yprev)``.
    """
    M, N = dz.shape
    K, N2 = w.shape
    assert N == N2
    if out is None:
        out = torch.empty((M, K), dtype=torch.float32, device=dz.device)
    if dz.is_cuda:
        nv.require_f32_contig(dz, w, yprev, out)
        fn = nv.kernels().dtf_dense_bwd_dx_bf16 if bf16 else nv.kernels().dtf_dense_bwd_dx
        rc = fn(nv.ptr(dz), nv.ptr(w), nv.ptr(out), M, N, K, nv.ptr(yprev), int(act_prev), float(scale_prev),
                nv.stream_ptr())
        nv.check(rc, "dtf_dense_bwd_dx")
        return out
    dx = _bf16_round(dz) @ _bf16_round(w).t() if bf16 else dz @ w.t()
    if yprev is not None:
        if act_prev == ACT_RELU:
            dx = torch.where(yprev > 0, dx * scale_prev, torch.zeros_like(dx))
        elif act_prev == ACT_SIGMOID:
            dx = dx * yprev * (1.0 - yprev)
    out.copy_(dx)
    return out


def dense_dw_workspace_floats(M: int, N: int, K: int) -> int:
    """Floats of scratch ``dense_backward_dw`` (and ``conv_gemm_wgrad`` of the same product) needs for a
    [M,K] x [M,N] weight gradient: one slab per batch split, for the larger of the two kernels' split counts."""
    if not nv.kernels_available():
        return 0
    s = max(nv.kernels().dtf_dense_bwd_dw_splits(M, N, K), nv.kernels().dtf_conv_wgrad_splits(M, N, K))
    return s * ((K * N + N + 3) // 4 * 4) if s > 1 else 0


def dense_backward_dw(x, dz, dw, db=None, workspace=None, bf16=False):
    """``dw = x.T @ dz`` and ``db = dz.sum(0)`` written into the given (flat-buffer) views."""
    M, K = x.shape
    M2, N = dz.shape
    assert M == M2 and tuple(dw.shape) == (K, N)
    if x.is_cuda:
        nv.require_f32_contig(x, dz, dw, db)
        if bf16:
            rc = nv.kernels().dtf_dense_bwd_dw_bf16(nv.ptr(x), nv.ptr(dz), nv.ptr(dw), nv.ptr(db), M, N, K, nv.stream_ptr())
            nv.check(rc, "dtf_dense_bwd_dw_bf16")
            return dw, db
        need = dense_dw_workspace_floats(M, N, K)
        if need and (workspace is None or workspace.numel() < need):
            workspace = torch.empty(need, dtype=torch.float32, device=x.device)
        rc = nv.kernels().dtf_dense_bwd_dw(nv.ptr(x), nv.ptr(dz), nv.ptr(dw), nv.ptr(db), M, N, K,
                                           nv.ptr(workspace), nv.stream_ptr())
        nv.check(rc, "dtf_dense_bwd_dw")
        return dw, db
    dw.copy_(_bf16_round(x).t() @ _bf16_round(dz) if bf16 else x.t() @ dz)
    if db is not None:
        db.copy_(dz.sum(0))
    return dw, db


def act_grad(dy, y, act, scale=1.0, out=None):
    """``dz = dy * act'(y)`` (relu/sigmoid expressed through the activation OUTPUT)."""
    if out is None:
        out = torch.empty_like(dy)
    if dy.is_cuda:
        nv.require_f32_contig(dy, y, out)
        rc = nv.kernels().dtf_act_grad(nv.ptr(dy), nv.ptr(y), nv.ptr(out), dy.numel(), int(act), float(scale),
                                       nv.stream_ptr())
        nv.check(rc, "dtf_act_grad")
        return out
    if act == ACT_RELU:
        r = torch.where(y > 0, dy * scale, torch.zeros_like(dy))
    elif act == ACT_SIGMOID:
        r = dy * y * (1.0 - y)
    elif act == ACT_TANH:
        r = dy * (1.0 - y * y)
    else:
        r = dy * scale
    out.copy_(r)
    return out


def dropout(x, keep, seed=0, step=None, out=None):
    """Inverted dropout with a mask that is a pure function of (seed, step, index).

    Calling it again on a gradient with the same (seed, step) applies the same mask, which
    is the backward pass.
    """
    if out is None:
        out = torch.empty_like(x)
    if x.is_cuda:
        nv.require_f32_contig(x, out)
        rc = nv.kernels().dtf_dropout(nv.ptr(x), nv.ptr(out), x.numel(), float(keep), int(seed) & _M32,
                                      nv.ptr(step), nv.stream_ptr())
        nv.check(rc, "dtf_dropout")
        return out
    u = dropout_uniform(dropout_key(seed, _step_value(step)), x.numel()).view(x.shape)
    out.copy_(torch.where(u < keep, x * (1.0 / keep), torch.zeros_like(x)))
    return out


def activation_forward(x, act, out=None):
    """Stand-alone activation (the fused form lives in the dense forward epilogue)."""
    if out is None:
        out = torch.empty_like(x)
    if x.is_cuda:
        nv.require_f32_contig(x, out)
        rc = nv.kernels().dtf_activation(nv.ptr(x), nv.ptr(out), x.numel(), int(act), nv.stream_ptr())
        nv.check(rc, "dtf_activation")
        return out
    out.copy_(_apply_act(x, act))
    return out


def batchnorm_workspace_floats(R: int, C: int) -> int:
    """Scratch floats ``batchnorm_forward`` / ``batchnorm_backward`` need on the GPU for ``[R, C]`` activations."""
    return int(nv.kernels().dtf_batchnorm_workspace_floats(int(R), int(C)))


def batchnorm_forward(x, gamma, beta, moving_mean, moving_var, momentum, eps, training, out, save_mean, save_inv_std,
                      workspace=None):
    """Batch normalisation over the last axis of ``x`` [R, C] (Keras ``BatchNormalization(axis=-1)``).

    ``training``: normalise with the batch's mean and (biased) variance, keep them in ``save_mean`` / ``save_inv_std``
    for the backward pass and move ``moving_mean`` / ``moving_var`` towards them (``m <- momentum m + (1 - momentum)
    batch``); else normalise with the moving statistics.  ``gamma`` / ``beta`` may be None."""
    R, C = x.shape
    if x.is_cuda:
        nv.require_f32_contig(x, gamma, beta, moving_mean, moving_var, out, save_mean, save_inv_std, workspace)
        assert workspace is not None and workspace.numel() >= batchnorm_workspace_floats(R, C)
        rc = nv.kernels().dtf_batchnorm_fwd(nv.ptr(x), nv.ptr(out), R, C, nv.ptr(gamma), nv.ptr(beta), nv.ptr(moving_mean),
                                            nv.ptr(moving_var), float(momentum), float(eps), int(bool(training)),
                                            nv.ptr(save_mean), nv.ptr(save_inv_std), nv.ptr(workspace), nv.stream_ptr())
        nv.check(rc, "dtf_batchnorm_fwd")
        return out
    if training:
        mean = x.mean(dim=0)
        var = x.var(dim=0, unbiased=False)
        save_mean.copy_(mean)
        save_inv_std.copy_(1.0 / torch.sqrt(var + eps))
        moving_mean.mul_(momentum).add_(mean, alpha=1.0 - momentum)
        moving_var.mul_(momentum).add_(var, alpha=1.0 - momentum)
        inv = save_inv_std
    else:
        mean, inv = moving_mean, 1.0 / torch.sqrt(moving_var + eps)
    y = (x - mean) * inv
    if gamma is not None:
        y = y * gamma
    if beta is not None:
        y = y + beta
    out.copy_(y)
    return out


def batchnorm_backward(x, dy, gamma, save_mean, save_inv_std, dgamma=None, dbeta=None, dx=None, workspace=None):
    """Gradient of the training-mode ``batchnorm_forward``: ``dbeta = sum dy``, ``dgamma = sum dy xhat``,
    ``dx = gamma inv_std (dy - dbeta / R - xhat dgamma / R)``; each output may be None."""
    R, C = x.shape
    if x.is_cuda:
        nv.require_f32_contig(x, dy, gamma, save_mean, save_inv_std, dgamma, dbeta, dx, workspace)
        assert workspace is not None and workspace.numel() >= batchnorm_workspace_floats(R, C)
        rc = nv.kernels().dtf_batchnorm_bwd(nv.ptr(x), nv.ptr(dy), nv.ptr(dx), R, C, nv.ptr(gamma), nv.ptr(save_mean),
                                            nv.ptr(save_inv_std), nv.ptr(dgamma), nv.ptr(dbeta), nv.ptr(workspace),
                                            nv.stream_ptr())
        nv.check(rc, "dtf_batchnorm_bwd")
        return dx
    xhat = (x - save_mean) * save_inv_std
    s0, s1 = dy.sum(dim=0), (dy * xhat).sum(dim=0)
    if dbeta is not None:
        dbeta.copy_(s0)
    if dgamma is not None:
        dgamma.copy_(s1)
    if dx is not None:
        a = save_inv_std if gamma is None else gamma * save_inv_std
        dx.copy_(a * (dy - s0 / R - xhat * (s1 / R)))
    return dx


def softmax(z, out=None):
    """Row softmax of logits [M, C]."""
    if out is None:
        out = torch.empty_like(z)
    M, Cn = z.shape
    if z.is_cuda:
        nv.require_f32_contig(z, out)
        rc = nv.kernels().dtf_softmax_rows(nv.ptr(z), nv.ptr(out), M, Cn, nv.stream_ptr())
        nv.check(rc, "dtf_softmax_rows")
        return out
    out.copy_(torch.softmax(z, dim=1))
    return out


# --------------------------------------------------------------------------- losses / metrics
def new_stats(device) -> torch.Tensor:
    """Running-statistics record: [sum of per-example loss, sum of per-example accuracy, examples, unused]."""
    return torch.zeros(4, dtype=torch.float32, device=device)


def mse_loss(p, y, grad_out=None, grad_scale=None, through_sigmoid=False, stats=None):
    """Keras ``mean_squared_error`` + rounded-equality accuracy, gradient fused.

    ``grad_out`` (optional, same shape as ``p``) receives d(mean loss)/dp, multiplied by
    ``p (1 - p)`` when ``through_sigmoid`` (the gradient w.r.t. the pre-sigmoid logits).
    ``stats`` accumulates (loss sum, accuracy sum, example count).
    """
    M, N = p.shape
    if grad_scale is None:
        grad_scale = 1.0 / (M * N)
    if p.is_cuda:
        nv.require_f32_contig(p, y, grad_out, stats)
        rc = nv.kernels().dtf_mse_loss(nv.ptr(p), nv.ptr(y), nv.ptr(grad_out), M, N, float(grad_scale),
                                       int(bool(through_sigmoid)), nv.ptr(stats), nv.stream_ptr())
        nv.check(rc, "dtf_mse_loss")
        return grad_out
    d = p - y
    if grad_out is not None:
        g = 2.0 * d * grad_scale
        if through_sigmoid:
            g = g * p * (1.0 - p)
        grad_out.copy_(g)
    if stats is not None:
        stats[0] += (d * d).mean(dim=1).sum()
        stats[1] += (torch.round(p) == torch.round(y)).to(torch.float32).mean(dim=1).sum()
        stats[2] += float(M)
    return grad_out


BCE_EPS = 1e-7


def bce_loss(p, y, grad_out=None, grad_scale=None, through_sigmoid=False, stats=None):
    """Keras ``binary_crossentropy`` on probabilities ``p`` (clipped to ``[1e-7, 1 - 1e-7]``, mean over the last axis)
    + rounded-equality accuracy, gradient fused: ``grad_out`` receives d(mean loss)/dp - or, ``through_sigmoid``, the
    gradient w.r.t. the logits, ``p - y`` - and zero where the clip is active."""
    M, N = p.shape
    if grad_scale is None:
        grad_scale = 1.0 / (M * N)
    if p.is_cuda:
        nv.require_f32_contig(p, y, grad_out, stats)
        rc = nv.kernels().dtf_bce_loss(nv.ptr(p), nv.ptr(y), nv.ptr(grad_out), M, N, float(grad_scale),
                                       int(bool(through_sigmoid)), nv.ptr(stats), nv.stream_ptr())
        nv.check(rc, "dtf_bce_loss")
        return grad_out
    pc = p.clamp(BCE_EPS, 1.0 - BCE_EPS)
    if grad_out is not None:
        g = (pc - y) if through_sigmoid else (pc - y) / (pc * (1.0 - pc))
        grad_out.copy_(torch.where(pc == p, g * grad_scale, torch.zeros_like(g)))
    if stats is not None:
        stats[0] += (-(y * pc.log() + (1.0 - y) * (1.0 - pc).log())).mean(dim=1).sum()
        stats[1] += (torch.round(p) == torch.round(y)).to(torch.float32).mean(dim=1).sum()
        stats[2] += float(M)
    return grad_out


def softmax_xent(z, labels, grad_out=None, grad_scale=None, stats=None):
    """Softmax cross-entropy on logits ``z`` [M,C] + argmax accuracy, gradient fused.

    ``labels`` is int64 [M] (sparse) or float32 [M,C] (one-hot / distribution).
    """
    M, Cn = z.shape
    if grad_scale is None:
        grad_scale = 1.0 / M
    sparse = labels.dtype == torch.int64
    if z.is_cuda:
        nv.require_f32_contig(z, grad_out, stats)
        assert labels.is_contiguous()
        if not sparse:
            nv.require_f32_contig(labels)
        rc = nv.kernels().dtf_softmax_xent(nv.ptr(z), nv.ptr(labels) if sparse else 0,
                                           0 if sparse else nv.ptr(labels), nv.ptr(grad_out), M, Cn,
                                           float(grad_scale), nv.ptr(stats), nv.stream_ptr())
        nv.check(rc, "dtf_softmax_xent")
        return grad_out
    logp = torch.log_softmax(z, dim=1)
    sm = logp.exp()
    if sparse:
        onehot = torch.zeros_like(z).scatter_(1, labels.view(-1, 1), 1.0)
        label_arg = labels
    else:
        onehot = labels
        label_arg = labels.argmax(dim=1)
    if grad_out is not None:
        grad_out.copy_((sm * onehot.sum(dim=1, keepdim=True) - onehot) * grad_scale)
    if stats is not None:
        stats[0] += -(onehot * logp).sum()
        stats[1] += (z.argmax(dim=1) == label_arg).to(torch.float32).sum()
        stats[2] += float(M)
    return grad_out


def categorical_accuracy_sum(p, y, acc_sum):
    """acc_sum[0] += number of rows with argmax(p) == argmax(y) (Keras categorical_accuracy)."""
    M, N = p.shape
    if p.is_cuda:
        nv.require_f32_contig(p, y, acc_sum)
        rc = nv.kernels().dtf_categorical_accuracy(nv.ptr(p), nv.ptr(y), M, N, nv.ptr(acc_sum), nv.stream_ptr())
        nv.check(rc, "dtf_categorical_accuracy")
        return acc_sum
    acc_sum[0] += (p.argmax(dim=1) == y.argmax(dim=1)).to(torch.float32).sum()
    return acc_sum


# --------------------------------------------------------------------------- optimizers
_tickets = {}


def _ticket(device) -> torch.Tensor:
    """Zeroed device counter the apply kernels use to elect their last workgroup (one per device)."""
    key = str(device)
    t = _tickets.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=device)
        _tickets[key] = t
    return t


def adam_apply(var, m, v, grad, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, advance=None,
               beta_powers=None):
    """TensorFlow-formula Adam on flat fp32 ranges; ``step`` is the int64 global_step BEFORE this update.
    ``advance`` (the global_step tensor, or None): also do ``step_advance`` at the end of the same launch."""
    n = var.numel()
    assert m.numel() == n and v.numel() == n and grad.numel() == n
    if var.is_cuda:
        nv.require_f32_contig(var, m, v, grad)
        rc = nv.kernels().dtf_adam_apply(nv.ptr(var), nv.ptr(m), nv.ptr(v), nv.ptr(grad), n, float(lr),
                                         float(beta1), float(beta2), float(eps), float(grad_scale),
                                         nv.ptr(step), nv.ptr(advance), nv.ptr(beta_powers if advance is not None else None),
                                         nv.ptr(_ticket(var.device)) if advance is not None else 0, nv.stream_ptr())
        nv.check(rc, "dtf_adam_apply")
        return var
    t = float(_step_value(step) + 1)
    b1 = torch.tensor(beta1, dtype=torch.float32)
    b2 = torch.tensor(beta2, dtype=torch.float32)
    lr_t = lr * torch.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    g = grad * grad_scale
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    var.sub_(lr_t * m / (v.sqrt() + eps))
    if advance is not None:
        step_advance(advance, beta_powers, beta1, beta2)
    return var


def sgd_apply(var, grad, lr, momentum_buf=None, momentum=0.0, grad_scale=1.0, advance=None):
    """``var -= lr * g`` (optionally with a classical momentum accumulator) on flat fp32 ranges.
    ``advance`` (the global_step tensor, or None): also increment it at the end of the same launch."""
    n = var.numel()
    if var.is_cuda:
        nv.require_f32_contig(var, grad, momentum_buf)
        rc = nv.kernels().dtf_sgd_apply(nv.ptr(var), nv.ptr(momentum_buf), nv.ptr(grad), n, float(lr),
                                        float(momentum), float(grad_scale), nv.ptr(advance),
                                        nv.ptr(_ticket(var.device)) if advance is not None else 0, nv.stream_ptr())
        nv.check(rc, "dtf_sgd_apply")
        return var
    g = grad * grad_scale
    if momentum_buf is not None:
        momentum_buf.mul_(momentum).add_(g)
        g = momentum_buf
    var.sub_(lr * g)
    if advance is not None:
        step_advance(advance)
    return var


def rmsprop_apply(var, ms, grad, lr, rho=0.9, momentum_buf=None, momentum=0.0, eps=1e-10, eps_inside=True, grad_scale=1.0,
                  advance=None):
    """RMSProp on flat fp32 ranges: ``ms <- rho ms + (1-rho) g^2``, ``u = lr g / D(ms)``, with a momentum
    accumulator ``mom <- momentum mom + u; var -= mom``, else ``var -= u``.  ``D(a) = sqrt(a + eps)`` (TensorFlow's
    ApplyRMSProp) or ``sqrt(a) + eps`` (Keras) by ``eps_inside``."""
    n = var.numel()
    assert ms.numel() == n and grad.numel() == n and (momentum_buf is None or momentum_buf.numel() == n)
    if var.is_cuda:
        nv.require_f32_contig(var, ms, grad, momentum_buf)
        rc = nv.kernels().dtf_rmsprop_apply(nv.ptr(var), nv.ptr(ms), nv.ptr(momentum_buf), nv.ptr(grad), n, float(lr),
                                            float(rho), float(momentum), float(eps), int(bool(eps_inside)), float(grad_scale),
                                            nv.ptr(advance), nv.ptr(_ticket(var.device)) if advance is not None else 0,
                                            nv.stream_ptr())
        nv.check(rc, "dtf_rmsprop_apply")
        return var
    g = grad * grad_scale
    ms.mul_(rho).addcmul_(g, g, value=1.0 - rho)
    u = lr * g / ((ms + eps).sqrt() if eps_inside else ms.sqrt() + eps)
    if momentum_buf is not None:
        momentum_buf.mul_(momentum).add_(u)
        u = momentum_buf
    var.sub_(u)
    if advance is not None:
        step_advance(advance)
    return var


def adagrad_apply(var, acc, grad, lr, eps=0.0, eps_inside=True, grad_scale=1.0, advance=None):
    """Adagrad on flat fp32 ranges: ``acc += g^2; var -= lr g / D(acc)`` (``D`` as in ``rmsprop_apply``; TensorFlow's
    ApplyAdagrad has no epsilon and starts ``acc`` at 0.1)."""
    n = var.numel()
    assert acc.numel() == n and grad.numel() == n
    if var.is_cuda:
        nv.require_f32_contig(var, acc, grad)
        rc = nv.kernels().dtf_adagrad_apply(nv.ptr(var), nv.ptr(acc), nv.ptr(grad), n, float(lr), float(eps),
                                            int(bool(eps_inside)), float(grad_scale), nv.ptr(advance),
                                            nv.ptr(_ticket(var.device)) if advance is not None else 0, nv.stream_ptr())
        nv.check(rc, "dtf_adagrad_apply")
        return var
    g = grad * grad_scale
    acc.addcmul_(g, g)
    var.sub_(lr * g / ((acc + eps).sqrt() if eps_inside else acc.sqrt() + eps))
    if advance is not None:
        step_advance(advance)
    return var


def adadelta_apply(var, acc, acc_update, grad, lr=1.0, rho=0.95, eps=1e-8, grad_scale=1.0, advance=None):
    """Adadelta on flat fp32 ranges: ``acc <- rho acc + (1-rho) g^2``, ``u = g sqrt(acc_update + eps) / sqrt(acc + eps)``,
    ``var -= lr u``, ``acc_update <- rho acc_update + (1-rho) u^2`` (TensorFlow's ApplyAdadelta = Keras' Adadelta)."""
    n = var.numel()
    assert acc.numel() == n and acc_update.numel() == n and grad.numel() == n
    if var.is_cuda:
        nv.require_f32_contig(var, acc, acc_update, grad)
        rc = nv.kernels().dtf_adadelta_apply(nv.ptr(var), nv.ptr(acc), nv.ptr(acc_update), nv.ptr(grad), n, float(lr), float(rho),
                                             float(eps), float(grad_scale), nv.ptr(advance),
                                             nv.ptr(_ticket(var.device)) if advance is not None else 0, nv.stream_ptr())
        nv.check(rc, "dtf_adadelta_apply")
        return var
    g = grad * grad_scale
    acc.mul_(rho).addcmul_(g, g, value=1.0 - rho)
    u = g * (acc_update + eps).sqrt() / (acc + eps).sqrt()
    acc_update.mul_(rho).addcmul_(u, u, value=1.0 - rho)
    var.sub_(lr * u)
    if advance is not None:
        step_advance(advance)
    return var


def step_advance(step, beta_powers=None, beta1=0.9, beta2=0.999):
    """``global_step += 1``; TF's non-slot ``beta1_power``/``beta2_power`` variables ``*= beta``."""
    if step.is_cuda:
        rc = nv.kernels().dtf_step_advance(nv.ptr(step), nv.ptr(beta_powers), float(beta1), float(beta2),
                                           nv.stream_ptr())
        nv.check(rc, "dtf_step_advance")
        return step
    step += 1
    if beta_powers is not None:
        beta_powers[0] *= beta1
        beta_powers[1] *= beta2
    return step


def init_uniform_(x: torch.Tensor, seed: int, lo: float, hi: float) -> torch.Tensor:
    """Fill ``x`` (flat fp32) with ``lo + (hi - lo) * u(seed, i)``, ``u`` a counter-based uniform in [0, 1): the same
    values on the GPU (one HBM-speed launch) and on the CPU."""
    n = x.numel()
    if x.is_cuda:
        nv.require_f32_contig(x)
        nv.check(nv.kernels().dtf_init_uniform(nv.ptr(x), n, int(seed) & _M32, float(lo), float(hi), nv.stream_ptr()),
                 "dtf_init_uniform")
        return x
    out = x.view(-1)
    chunk = 1 << 22
    for b in range(0, n, chunk):
        i = torch.arange(b, min(n, b + chunk), dtype=torch.int64)
        a = _mix32((((i & _M32) * 0x9E3779B1) + (int(seed) & _M32)) & _M32)
        r = _mix32(a ^ ((((i >> 32) * 0x85EBCA6B) + 0x27D4EB2F) & _M32))
        u = (r >> 8).to(torch.float32) * (1.0 / 16777216.0)
        lo32, hi32 = torch.tensor(lo, dtype=torch.float32), torch.tensor(hi, dtype=torch.float32)
        out[b:b + i.numel()] = lo32 + (hi32 - lo32) * u
    return x


def xor_data(n, bits, seed, device):
    """Device-side generator of the XOR task of /root/reference/example.py:24-48."""
    x = torch.empty((n, 2 * bits), dtype=torch.float32, device=device)
    y = torch.empty((n, bits), dtype=torch.float32, device=device)
    if x.is_cuda:
        rc = nv.kernels().dtf_xor_data(nv.ptr(x), nv.ptr(y), n, bits, int(seed) & _M32, nv.stream_ptr())
        nv.check(rc, "dtf_xor_data")
        return x, y
    s = torch.arange(n, dtype=torch.int64).view(n, 1)
    k = torch.arange(bits, dtype=torch.int64).view(1, bits)
    r = _mix32((_mix32(((s * 0x9E3779B1) + (seed & _M32)) & _M32) + ((k * 0x85EBCA6B) & _M32)) & _M32)
    a = (r >> 9) & 1
    b = (r >> 21) & 1
    x[:, :bits] = a.to(torch.float32)
    x[:, bits:] = b.to(torch.float32)
    y.copy_((a ^ b).to(torch.float32))
    return x, y


def mlp3_fused_supported(dims, batch: int) -> bool:
    """True when the whole-step kernel (csrc/kernels/mlp_fused.hip) covers a 3-layer MLP of these widths."""
    if len(dims) != 4 or not nv.kernels_available():
        return False
    lib = nv.kernels()
    return hasattr(lib, "dtf_mlp3_fused_supported") and bool(lib.dtf_mlp3_fused_supported(*map(int, dims), int(batch)))


_fused_ctl = {}


def _fused_control_words(device) -> torch.Tensor:
    """The four control words of the asynchronous-workers launch ({elected XCD, admitted, left, tickets}); every
    launch leaves them as it found them."""
    key = (device.type, device.index)
    if key not in _fused_ctl:
        _fused_ctl[key] = torch.tensor([-1, 0, 0, 0], dtype=torch.int32, device=device)
    return _fused_ctl[key]


def mlp3_fused_train(x, y, params, slot_m, slot_v, offsets, dims, batch, nsteps, training, opt, want_cat, keep1, keep2,
                     seed1, seed2, lr, beta1, beta2, eps, step, beta_powers, stats, workers: int = 1):
    """``nsteps`` complete training steps (forward, MSE, backward, Adam/SGD update) of the reference MLP in ONE
    launch on mini-batches ``x[k*batch:(k+1)*batch]``; GPU only.  ``offsets``: element offsets of
    W1, b1, W2, b2, W3, b3 inside ``params`` (and the slot buffers).

    ``workers > 1``: that many asynchronous workers (one compute unit each) take the ``nsteps`` steps between them,
    updating the shared variables without locks - the reference's parameter-server semantics
    (/root/reference/example.py:168-170) inside one GPU; ``global_step`` still advances by exactly ``nsteps``."""
    import ctypes as C
    nv.require_f32_contig(x, y, params, slot_m, slot_v)
    if not x.is_cuda:
        raise ValueError("the fused step kernel runs on the GPU only")
    if x.shape[0] < batch * nsteps or y.shape[0] < batch * nsteps or x.shape[1] != dims[0] or y.shape[1] != dims[3]:
        raise ValueError("fused step: data %s/%s does not hold %d batches of %d x %s" %
                         (tuple(x.shape), tuple(y.shape), nsteps, batch, (dims[0], dims[3])))
    need = max(offsets[0] + dims[0] * dims[1], offsets[2] + dims[1] * dims[2], offsets[4] + dims[2] * dims[3],
               offsets[1] + dims[1], offsets[3] + dims[2], offsets[5] + dims[3])
    for t in (params, slot_m, slot_v):
        if t is not None and t.numel() < need:
            raise ValueError("fused step: flat buffer of %d elements is shorter than the layout (%d)" % (t.numel(), need))
    if any(o % 4 for o in offsets):
        raise ValueError("fused step: variable offsets must be multiples of 4 elements")
    offs = (C.c_longlong * 6)(*[int(o) for o in offsets])
    if int(workers) > 1:
        fn = nv.declare(nv.kernels(), "dtf_mlp3_fused_train_async",
                        [nv.c_p] * 6 + [nv.c_i] * 9 + [nv.c_f, nv.c_f, nv.c_u32, nv.c_u32] + [nv.c_f] * 4 + [nv.c_p] * 4
                        + [nv.c_i, nv.c_p])
        rc = fn(nv.ptr(x), nv.ptr(y), nv.ptr(params), nv.ptr(slot_m), nv.ptr(slot_v), C.cast(offs, C.c_void_p),
                int(dims[0]), int(dims[1]), int(dims[2]), int(dims[3]), int(batch), int(nsteps), int(bool(training)),
                int(opt), int(bool(want_cat)), float(keep1), float(keep2), int(seed1) & _M32, int(seed2) & _M32, float(lr),
                float(beta1), float(beta2), float(eps), nv.ptr(step), nv.ptr(beta_powers), nv.ptr(stats),
                nv.ptr(_fused_control_words(x.device)), int(workers), nv.stream_ptr())
        nv.check(rc, "dtf_mlp3_fused_train_async")
        return
    rc = nv.kernels().dtf_mlp3_fused_train(
        nv.ptr(x), nv.ptr(y), nv.ptr(params), nv.ptr(slot_m), nv.ptr(slot_v), C.cast(offs, C.c_void_p),
        int(dims[0]), int(dims[1]), int(dims[2]), int(dims[3]), int(batch), int(nsteps), int(bool(training)), int(opt),
        int(bool(want_cat)), float(keep1), float(keep2), int(seed1) & _M32, int(seed2) & _M32, float(lr), float(beta1),
        float(beta2), float(eps), nv.ptr(step), nv.ptr(beta_powers), nv.ptr(stats), nv.stream_ptr())
    nv.check(rc, "dtf_mlp3_fused_train")


def mlp3_fused_grad(x, y, params, grads, offsets, dims, training, want_cat, keep1, keep2, seed1, seed2, step, stats):
    """Forward, MSE loss (+ metrics into ``stats``) and backward of the reference MLP on ONE mini-batch in one launch:
    the gradients land in ``grads`` at the variables' ``offsets``; variables, step counter and optimizer state are not
    touched.  GPU only; batch <= 64."""
    import ctypes as C
    nv.require_f32_contig(x, y, params, grads)
    if not x.is_cuda:
        raise ValueError("the fused gradient kernel runs on the GPU only")
    batch = int(x.shape[0])
    if y.shape[0] != batch or x.shape[1] != dims[0] or y.shape[1] != dims[3]:
        raise ValueError("fused gradients: batch %s/%s does not match %s" % (tuple(x.shape), tuple(y.shape), (dims[0], dims[3])))
    need = max(offsets[0] + dims[0] * dims[1], offsets[2] + dims[1] * dims[2], offsets[4] + dims[2] * dims[3],
               offsets[1] + dims[1], offsets[3] + dims[2], offsets[5] + dims[3])
    if params.numel() < need or grads.numel() < need:
        raise ValueError("fused gradients: flat buffers are shorter than the layout (%d)" % need)
    if any(o % 4 for o in offsets):
        raise ValueError("fused gradients: variable offsets must be multiples of 4 elements")
    offs = (C.c_longlong * 6)(*[int(o) for o in offsets])
    fn = nv.declare(nv.kernels(), "dtf_mlp3_fused_grad",
                    [nv.c_p] * 5 + [nv.c_i] * 7 + [nv.c_f, nv.c_f, nv.c_u32, nv.c_u32] + [nv.c_p] * 3)
    rc = fn(nv.ptr(x), nv.ptr(y), nv.ptr(params), nv.ptr(grads), C.cast(offs, C.c_void_p), int(dims[0]), int(dims[1]),
            int(dims[2]), int(dims[3]), batch, int(bool(training)), int(bool(want_cat)), float(keep1), float(keep2),
            int(seed1) & _M32, int(seed2) & _M32, nv.ptr(step), nv.ptr(stats), nv.stream_ptr())
    nv.check(rc, "dtf_mlp3_fused_grad")


def mlp3_fused_grad_direct(x, y, var_ptrs, grad_ptrs, dims, training, want_cat, keep1, keep2, seed1, seed2, step_address,
                           stats, blocks, worker, owners, slot, seq, clock):
    """A parameter-server worker's whole step in one launch: the reference MLP's variables are read from
    ``var_ptrs`` (six device addresses in peer-mapped memory of the owning ps tasks), the gradients written to
    ``grad_ptrs`` (this worker's inbox slots there), and - with ``blocks`` (a ``ControlBlocks`` in replica mode) - the
    kernel ends by ringing the doorbells of owners ``0 .. owners-1`` for worker ``worker`` itself; ``blocks=None``:
    the caller rings them behind this launch.  ``step_address``: device address of the global step.  GPU only;
    batch <= 64."""
    import ctypes as C
    nv.require_f32_contig(x, y)
    if not x.is_cuda:
        raise ValueError("the fused gradient kernel runs on the GPU only")
    batch = int(x.shape[0])
    if y.shape[0] != batch or x.shape[1] != dims[0] or y.shape[1] != dims[3]:
        raise ValueError("fused gradients: batch %s/%s does not match %s" % (tuple(x.shape), tuple(y.shape), (dims[0], dims[3])))
    if len(var_ptrs) != 6 or len(grad_ptrs) != 6 or not all(var_ptrs) or not all(grad_ptrs) or not step_address:
        raise ValueError("fused gradients: six variables, no null device address")
    n = int(owners) if blocks is not None else 0
    if n > 8 or (blocks is not None and blocks.replicas is None):
        raise ValueError("fused gradients: in-kernel doorbells need replica-mode control blocks and at most eight owners")
    fn = nv.declare(nv.kernels(), "dtf_mlp3_fused_grad_direct",
                    [nv.c_p] * 4 + [nv.c_i] * 7 + [nv.c_f, nv.c_f, nv.c_u32, nv.c_u32] + [nv.c_p] * 3
                    + [nv.c_i, nv.c_i, nv.c_p, nv.c_p, nv.c_i, nv.c_i, C.c_ulonglong, C.c_ulonglong, nv.c_p])
    bw = (C.c_longlong * n)(*[blocks.block_word(j) for j in range(n)]) if n else None
    lw = (C.c_longlong * n)(*[blocks.lane_word(j, worker) for j in range(n)]) if n else None
    rc = fn(nv.ptr(x), nv.ptr(y), (C.c_void_p * 6)(*var_ptrs), (C.c_void_p * 6)(*grad_ptrs), int(dims[0]), int(dims[1]),
            int(dims[2]), int(dims[3]), batch, int(bool(training)), int(bool(want_cat)), float(keep1), float(keep2),
            int(seed1) & _M32, int(seed2) & _M32, C.c_void_p(step_address), nv.ptr(stats),
            *(blocks.ring_array() if n else (None, 0, 0)), bw, lw, n, int(slot), int(seq), int(clock), nv.stream_ptr())
    nv.check(rc, "dtf_mlp3_fused_grad_direct")


# ----------------------------------------------------------------------------- convolution support (NHWC)
def conv_geometry(H, W, KH, KW, strides=(1, 1), padding="valid"):
    """Output size and top/left zero padding, TensorFlow rules ('same': out = ceil(in / stride), extra padding goes
    to the bottom/right)."""
    sh, sw = strides
    if padding == "same":
        OH, OW = -(-H // sh), -(-W // sw)
        ph = max((OH - 1) * sh + KH - H, 0)
        pw = max((OW - 1) * sw + KW - W, 0)
        return OH, OW, ph // 2, pw // 2, ph - ph // 2, pw - pw // 2
    if padding != "valid":
        raise ValueError("padding must be 'valid' or 'same'")
    return (H - KH) // sh + 1, (W - KW) // sw + 1, 0, 0, 0, 0


def im2col(x, KH, KW, strides=(1, 1), padding="valid", out=None):
    """Patch matrix ``[B*OH*OW, KH*KW*C]`` of an NHWC tensor, column order (kh, kw, c) = a TF filter flattened."""
    B, H, W, C = x.shape
    OH, OW, pt, pl, pb, pr = conv_geometry(H, W, KH, KW, strides, padding)
    M, K = B * OH * OW, KH * KW * C
    if out is None:
        out = torch.empty((M, K), dtype=torch.float32, device=x.device)
    if x.is_cuda:
        nv.require_f32_contig(x, out)
        rc = nv.kernels().dtf_im2col(nv.ptr(x), nv.ptr(out), B, H, W, C, KH, KW, OH, OW, pt, pl, strides[0], strides[1],
                                     nv.stream_ptr())
        nv.check(rc, "dtf_im2col")
        return out
    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    u = torch.nn.functional.unfold(xp, (KH, KW), stride=strides)                    # [B, C*KH*KW, L], (c, kh, kw) order
    u = u.view(B, C, KH, KW, OH * OW).permute(0, 4, 2, 3, 1).reshape(M, K)
    out.copy_(u)
    return out


def col2im(dcols, x_shape, KH, KW, strides=(1, 1), padding="valid", yprev=None, out=None):
    """Adjoint of ``im2col``: scatter-add patch gradients back to the NHWC image; ``yprev`` (the producing layer's ReLU
    output, same shape as the image) folds that layer's activation gradient in."""
    B, H, W, C = x_shape
    OH, OW, pt, pl, pb, pr = conv_geometry(H, W, KH, KW, strides, padding)
    if out is None:
        out = torch.empty(tuple(x_shape), dtype=torch.float32, device=dcols.device)
    if dcols.is_cuda:
        nv.require_f32_contig(dcols, out, yprev)
        rc = nv.kernels().dtf_col2im(nv.ptr(dcols), nv.ptr(out), nv.ptr(yprev), B, H, W, C, KH, KW, OH, OW, pt, pl,
                                     strides[0], strides[1], nv.stream_ptr())
        nv.check(rc, "dtf_col2im")
        return out
    u = dcols.view(B, OH * OW, KH, KW, C).permute(0, 4, 2, 3, 1).reshape(B, C * KH * KW, OH * OW)
    f = torch.nn.functional.fold(u, (H + pt + pb, W + pl + pr), (KH, KW), stride=strides)
    dx = f[:, :, pt:pt + H, pl:pl + W].permute(0, 2, 3, 1)
    if yprev is not None:
        dx = dx * (yprev > 0).to(dx.dtype)
    out.copy_(dx)
    return out


def maxpool_forward(x, pool=(2, 2), out=None, arg=None):
    """Non-overlapping max pooling (stride = window, VALID) of an NHWC tensor; returns ``(y, argmax uint8)``."""
    B, H, W, C = x.shape
    PH, PW = pool
    OH, OW = H // PH, W // PW
    if out is None:
        out = torch.empty((B, OH, OW, C), dtype=torch.float32, device=x.device)
    if arg is None:
        arg = torch.empty((B, OH, OW, C), dtype=torch.uint8, device=x.device)
    if x.is_cuda:
        nv.require_f32_contig(x, out)
        rc = nv.kernels().dtf_maxpool_fwd(nv.ptr(x), nv.ptr(out), nv.ptr(arg), B, H, W, C, PH, PW, nv.stream_ptr())
        nv.check(rc, "dtf_maxpool_fwd")
        return out, arg
    win = x[:, :OH * PH, :OW * PW, :].reshape(B, OH, PH, OW, PW, C).permute(0, 1, 3, 5, 2, 4).reshape(B, OH, OW, C, PH * PW)
    v, i = win.max(dim=-1)
    out.copy_(v)
    arg.copy_(i.to(torch.uint8))
    return out, arg


def pool_geometry(H, W, PH, PW, strides, padding="valid"):
    """``(OH, OW, pad_top, pad_left)`` of a pooling window, TensorFlow rules ('same': out = ceil(in / stride), the extra
    padding goes to the bottom / right; padded positions never win a maximum)."""
    SH, SW = strides
    if str(padding).lower() == "same":
        OH, OW = -(-H // SH), -(-W // SW)
        pt = max((OH - 1) * SH + PH - H, 0) // 2
        pl = max((OW - 1) * SW + PW - W, 0) // 2
    else:
        OH, OW = (H - PH) // SH + 1, (W - PW) // SW + 1
        pt = pl = 0
    return OH, OW, pt, pl


def _avg_counts(H, W, PH, PW, strides, padding, device):
    """Number of in-image positions of every pooling window, ``[OH, OW]``."""
    OH, OW, pt, pl = pool_geometry(H, W, PH, PW, strides, padding)
    oh = torch.arange(OH, device=device) * strides[0] - pt
    ow = torch.arange(OW, device=device) * strides[1] - pl
    ch = (oh + PH).clamp(max=H) - oh.clamp(min=0)
    cw = (ow + PW).clamp(max=W) - ow.clamp(min=0)
    return (ch[:, None] * cw[None, :]).to(torch.float32)


def avgpool_forward(x, pool=(2, 2), strides=None, padding="valid", out=None):
    """Average pooling of an NHWC tensor; a window over the border averages over its in-image positions (TensorFlow)."""
    B, H, W, C = x.shape
    PH, PW = pool
    strides = tuple(strides) if strides is not None else (PH, PW)
    OH, OW, pt, pl = pool_geometry(H, W, PH, PW, strides, padding)
    if out is None:
        out = torch.empty((B, OH, OW, C), dtype=torch.float32, device=x.device)
    if x.is_cuda:
        nv.require_f32_contig(x, out)
        rc = nv.kernels().dtf_avgpool_fwd(nv.ptr(x), nv.ptr(out), B, H, W, C, PH, PW, strides[0], strides[1], OH, OW, pt, pl,
                                          nv.stream_ptr())
        nv.check(rc, "dtf_avgpool_fwd")
        return out
    pb = max((OH - 1) * strides[0] + PH - H - pt, 0)
    pr = max((OW - 1) * strides[1] + PW - W - pl, 0)
    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    sums = torch.nn.functional.avg_pool2d(xp, (PH, PW), strides, divisor_override=1)
    out.copy_((sums / _avg_counts(H, W, PH, PW, strides, padding, x.device)).permute(0, 2, 3, 1))
    return out


def avgpool_backward(dy, x_shape, pool=(2, 2), strides=None, padding="valid", out=None):
    """Gradient of ``avgpool_forward``: every input position collects ``dy / count`` of the windows that cover it."""
    B, H, W, C = x_shape
    PH, PW = pool
    strides = tuple(strides) if strides is not None else (PH, PW)
    OH, OW, pt, pl = pool_geometry(H, W, PH, PW, strides, padding)
    if out is None:
        out = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    if dy.is_cuda:
        nv.require_f32_contig(dy, out)
        rc = nv.kernels().dtf_avgpool_bwd(nv.ptr(dy), nv.ptr(out), B, H, W, C, PH, PW, strides[0], strides[1], OH, OW, pt, pl,
                                          nv.stream_ptr())
        nv.check(rc, "dtf_avgpool_bwd")
        return out
    # reference path: scatter dy / count back through the windows
    g = (dy / _avg_counts(H, W, PH, PW, strides, padding, dy.device)[None, :, :, None]).permute(0, 3, 1, 2)
    pb = max((OH - 1) * strides[0] + PH - H - pt, 0)
    pr = max((OW - 1) * strides[1] + PW - W - pl, 0)
    Hp, Wp = H + pt + pb, W + pl + pr
    cols = g.reshape(B, C, 1, OH * OW).expand(B, C, PH * PW, OH * OW).reshape(B, C * PH * PW, OH * OW)
    full = torch.nn.functional.fold(cols, (Hp, Wp), (PH, PW), stride=strides)
    out.copy_(full[:, :, pt:pt + H, pl:pl + W].permute(0, 2, 3, 1))
    return out


def maxpool_strided_forward(x, pool=(2, 2), strides=(2, 2), padding="valid", out=None, arg=None):
    """Max pooling of an NHWC tensor with any window, strides and VALID / SAME padding; returns ``(y, arg)`` with
    ``arg`` the window position ``i * PW + j`` of each maximum."""
    B, H, W, C = x.shape
    PH, PW = pool
    SH, SW = strides
    OH, OW, pt, pl = pool_geometry(H, W, PH, PW, strides, padding)
    if out is None:
        out = torch.empty((B, OH, OW, C), dtype=torch.float32, device=x.device)
    if arg is None:
        arg = torch.empty((B, OH, OW, C), dtype=torch.uint8, device=x.device)
    if x.is_cuda:
        nv.require_f32_contig(x, out)
        rc = nv.kernels().dtf_maxpool_strided_fwd(nv.ptr(x), nv.ptr(out), nv.ptr(arg), B, H, W, C, PH, PW, SH, SW, OH, OW, pt,
                                                  pl, nv.stream_ptr())
        nv.check(rc, "dtf_maxpool_strided_fwd")
        return out, arg
    pb, pr = max((OH - 1) * SH + PH - H - pt, 0), max((OW - 1) * SW + PW - W - pl, 0)
    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float("-inf"))
    u = torch.nn.functional.unfold(xp, (PH, PW), stride=(SH, SW)).view(B, C, PH * PW, OH * OW)
    v, i = u.max(dim=2)                                       # first maximum in (i, j) order, as the kernel's scan
    out.copy_(v.view(B, C, OH, OW).permute(0, 2, 3, 1))
    arg.copy_(i.view(B, C, OH, OW).permute(0, 2, 3, 1).to(torch.uint8))
    return out, arg


def maxpool_strided_backward(dy, arg, x_shape, pool=(2, 2), strides=(2, 2), padding="valid", x_relu=None, out=None):
    """Gradient of ``maxpool_strided_forward``: every input element receives the gradients of the windows that chose
    it (summed where windows overlap); ``x_relu`` (the pooling's input) folds the gradient of a ReLU in front in."""
    B, H, W, C = x_shape
    PH, PW = pool
    SH, SW = strides
    OH, OW, pt, pl = pool_geometry(H, W, PH, PW, strides, padding)
    if out is None:
        out = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    if dy.is_cuda:
        nv.require_f32_contig(dy, out, x_relu)
        rc = nv.kernels().dtf_maxpool_strided_bwd(nv.ptr(dy), nv.ptr(arg), nv.ptr(x_relu), nv.ptr(out), B, H, W, C, PH, PW, SH,
                                                  SW, OH, OW, pt, pl, nv.stream_ptr())
        nv.check(rc, "dtf_maxpool_strided_bwd")
        return out
    a = arg.to(torch.int64)
    oh = torch.arange(OH).view(1, OH, 1, 1)
    ow = torch.arange(OW).view(1, 1, OW, 1)
    ih = oh * SH - pt + a // PW                                # input position every window chose
    iw = ow * SW - pl + a % PW
    flat = ((torch.arange(B).view(B, 1, 1, 1) * H + ih) * W + iw) * C + torch.arange(C).view(1, 1, 1, C)
    out.zero_()
    out.view(-1).index_add_(0, flat.reshape(-1), dy.reshape(-1))
    if x_relu is not None:
        out.mul_((x_relu > 0).to(out.dtype))
    return out


def conv_pool_supported(x_shape, KH, KW, filters, strides=(1, 1), padding="valid", pool=(2, 2)) -> bool:
    """True when a convolution followed by 2x2 max pooling can run as the fused first-layer kernels
    (csrc/kernels/conv_pool.hip): stride 1, KH*KW*Cin < 32, filters a multiple of 32, even conv output size."""
    if not nv.kernels_available() or tuple(strides) != (1, 1) or tuple(pool) != (2, 2):
        return False
    _, H, W, C = x_shape
    OH, OW = conv_geometry(H, W, KH, KW, strides, padding)[:2]
    if OH <= 0 or OW <= 0 or OH % 2 or OW % 2:
        return False
    return bool(nv.kernels().dtf_conv_pool_supported(C, KH, KW, filters))


_conv_pool_env_read = False


def conv_pool_staged_fits(B, H, W, C, KH, KW, pad_t, pad_l, PH, PW, filters) -> bool:
    """True when the fused first-layer kernels can stage whole images of this geometry in LDS (zero padding included):
    they then read the unpadded input.  ``DTF_CONV_POOL_STAGED=0`` switches the staged form off."""
    global _conv_pool_env_read
    if not nv.kernels_available():
        return False
    lib = nv.kernels()
    if not _conv_pool_env_read:
        import os
        _conv_pool_env_read = True
        if os.environ.get("DTF_CONV_POOL_STAGED", "1") == "0":
            lib.dtf_conv_pool_set_staged(0)
    return bool(lib.dtf_conv_pool_staged_fits(int(B), int(H), int(W), int(C), int(KH), int(KW), int(pad_t), int(pad_l), int(PH),
                                              int(PW), int(filters)))


def conv_pool_workspace_floats(C, KH, KW, filters) -> int:
    k = nv.kernels()
    k.dtf_conv_pool_workspace_floats.restype = nv.c_ll
    return int(k.dtf_conv_pool_workspace_floats(C, KH, KW, filters))


def conv_pool_forward(x, w2d, bias, act, KH, KW, padding, ypool, arg):
    """``ypool = maxpool2x2(act(conv(x, w) + bias))`` and the window member of each max, in one kernel (GPU only)."""
    B, H, W, C = x.shape
    OH, OW, pt, pl, _, _ = conv_geometry(H, W, KH, KW, (1, 1), padding)
    Fo = w2d.shape[1]
    nv.require_f32_contig(x, w2d, bias, ypool)
    assert tuple(ypool.shape) == (B, OH // 2, OW // 2, Fo) and arg.shape == ypool.shape and arg.dtype == torch.uint8
    rc = nv.kernels().dtf_conv_pool_fwd(nv.ptr(x), nv.ptr(w2d), nv.ptr(bias), nv.ptr(ypool), nv.ptr(arg), B, H, W, C, KH, KW,
                                        pt, pl, OH // 2, OW // 2, Fo, act, nv.stream_ptr())
    nv.check(rc, "dtf_conv_pool_fwd")
    return ypool, arg


def conv_pool_wgrad(x, dpool, ypool, arg, dw2d, db, act, KH, KW, padding, workspace):
    """Filter and bias gradients of the fused convolution + pooling from the POOLED gradient (GPU only)."""
    B, H, W, C = x.shape
    OH, OW, pt, pl, _, _ = conv_geometry(H, W, KH, KW, (1, 1), padding)
    Fo = dw2d.shape[1]
    nv.require_f32_contig(x, dpool, ypool, dw2d, db, workspace)
    assert dpool.shape == ypool.shape == arg.shape and dw2d.shape[0] == KH * KW * C
    assert workspace.numel() >= conv_pool_workspace_floats(C, KH, KW, Fo)
    rc = nv.kernels().dtf_conv_pool_wgrad(nv.ptr(x), nv.ptr(dpool), nv.ptr(ypool), nv.ptr(arg), nv.ptr(dw2d), nv.ptr(db), B, H, W,
                                          C, KH, KW, pt, pl, OH // 2, OW // 2, Fo, act, nv.ptr(workspace), nv.stream_ptr())
    nv.check(rc, "dtf_conv_pool_wgrad")


def maxpool_backward(dy, arg, x_shape, pool=(2, 2), x_relu=None, out=None, mask_pooled=False):
    """Route ``dy`` to each window's arg-max position; ``x_relu`` folds the gradient of the ReLU in front of the pooling
    in: the pooling's full-resolution input, or - ``mask_pooled`` - its pooled OUTPUT (same value at the arg-max)."""
    B, H, W, C = x_shape
    PH, PW = pool
    OH, OW = H // PH, W // PW
    if out is None:
        out = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    if dy.is_cuda:
        nv.require_f32_contig(dy, out, x_relu)
        rc = nv.kernels().dtf_maxpool_bwd(nv.ptr(dy), nv.ptr(arg), nv.ptr(x_relu), nv.ptr(out), B, H, W, C, PH, PW,
                                          int(mask_pooled), nv.stream_ptr())
        nv.check(rc, "dtf_maxpool_bwd")
        return out
    onehot = torch.nn.functional.one_hot(arg.to(torch.int64), PH * PW).to(dy.dtype) * dy.unsqueeze(-1)   # [B,OH,OW,C,P]
    g = onehot.view(B, OH, OW, C, PH, PW).permute(0, 1, 4, 2, 5, 3).reshape(B, OH * PH, OW * PW, C)
    out.zero_()
    out[:, :OH * PH, :OW * PW, :] = g
    if x_relu is not None:
        if mask_pooled:
            keep = (x_relu > 0).to(out.dtype).repeat_interleave(PH, dim=1).repeat_interleave(PW, dim=2)
            out[:, :OH * PH, :OW * PW, :].mul_(keep)
        else:
            out.mul_((x_relu > 0).to(out.dtype))
    return out


# ----------------------------------------------------------------------------- fused network tail
def mlp_tail_supported(h1: int, h2: int, classes: int) -> bool:
    """True when csrc/kernels/mlp_tail.hip covers Dense(h1->h2, relu[, dropout]) -> Dense(h2->classes) -> sparse
    softmax cross-entropy."""
    if not nv.kernels_available():
        return False
    lib = nv.kernels()
    return hasattr(lib, "dtf_mlp_tail_supported") and bool(lib.dtf_mlp_tail_supported(int(h1), int(h2), int(classes)))


def mlp_tail_workspace_floats(M: int, classes: int) -> int:
    import ctypes as C
    fn = nv.declare(nv.kernels(), "dtf_mlp_tail_workspace_floats", [C.c_int, C.c_int], C.c_longlong)
    return int(fn(int(M), int(classes)))


def mlp_tail(h1, w2, b2, w3, b3, labels, dz1, dw2, db2, dw3, db3, workspace, keep2, seed2, step, scale1, grad_scale,
             stats):
    """Forward from ``h1`` through the softmax cross-entropy and backward to ``dz1`` / ``dw2`` / ``db2`` / ``dw3`` /
    ``db3`` in one fused kernel (plus two slab reductions); GPU only."""
    M, H1 = h1.shape
    H2, C = w3.shape
    nv.require_f32_contig(h1, w2, b2, w3, b3, dz1, dw2, db2, dw3, db3, workspace)
    if not h1.is_cuda:
        raise ValueError("the fused tail kernel runs on the GPU only")
    if labels.dtype != torch.int64 or labels.shape != (M,) or not labels.is_contiguous():
        raise ValueError("fused tail: labels must be a contiguous int64 vector of length %d" % M)
    if tuple(w2.shape) != (H1, H2) or tuple(dz1.shape) != (M, H1) or tuple(dw2.shape) != (H1, H2) or \
            tuple(dw3.shape) != (H2, C) or b2.numel() != H2 or b3.numel() != C or db2.numel() != H2 or db3.numel() != C:
        raise ValueError("fused tail: inconsistent operand shapes")
    rc = nv.kernels().dtf_mlp_tail(
        nv.ptr(h1), nv.ptr(w2), nv.ptr(b2), nv.ptr(w3), nv.ptr(b3), nv.ptr(labels), nv.ptr(dz1), nv.ptr(dw2), nv.ptr(db2),
        nv.ptr(dw3), nv.ptr(db3), nv.ptr(workspace), int(workspace.numel()), int(M), int(H1), int(H2), int(C),
        float(keep2), int(seed2) & _M32, nv.ptr(step), float(scale1), float(grad_scale), nv.ptr(stats), nv.stream_ptr())
    nv.check(rc, "dtf_mlp_tail")


# ----------------------------------------------------------------------------- implicit-GEMM convolution (GPU, stride 1)
def conv_implicit_supported(c_in: int, c_out: int, strides, need_dgrad: bool = True) -> bool:
    """The GEMM kernels can gather patch rows straight from the image (no patch matrix in HBM) when a 4-float vector
    lies inside one window position: ``c_in % 4 == 0``, any strides (forward and filter gradient).  The data gradient
    is gathered the same way - a convolution over the output gradient, ``c_out % 4 == 0`` - at stride 1; a strided
    layer computes it as a patch-gradient product scattered back by ``col2im``.  Multiples of 32 channels take the
    loaders' fast path (a whole 32-deep k tile inside one window position)."""
    unit = tuple(strides) == (1, 1)
    return c_in % 4 == 0 and (not need_dgrad or not unit or c_out % 4 == 0) and nv.kernels_available() \
        and hasattr(nv.kernels(), "dtf_conv_gemm_fwd")


def conv_pad(x, pt: int, pl: int, Hp: int, Wp: int, out=None):
    """Zero-padded copy ``[B, Hp, Wp, C]`` of an NHWC tensor with the image at offset (pt, pl)."""
    B, H, W, C = x.shape
    if out is None:
        out = torch.empty((B, Hp, Wp, C), dtype=torch.float32, device=x.device)
    nv.require_f32_contig(x, out)
    rc = nv.kernels().dtf_conv_pad(nv.ptr(x), nv.ptr(out), B, H, W, C, pt, pl, Hp, Wp, nv.stream_ptr())
    nv.check(rc, "dtf_conv_pad")
    return out


def conv_tables(B: int, OH: int, OW: int, Hp: int, Wp: int, KH: int, KW: int, C: int, device, strides=(1, 1)):
    """``(rowtab, shift)``: element offset of every output pixel's patch origin in the padded image, and of every
    window position relative to that origin."""
    rowtab = torch.empty(B * OH * OW, dtype=torch.int32, device=device)
    rc = nv.kernels().dtf_conv_rowtab(nv.ptr(rowtab), B, OH, OW, Hp, Wp, C, int(strides[0]), int(strides[1]), nv.stream_ptr())
    nv.check(rc, "dtf_conv_rowtab")
    kh = torch.arange(KH, dtype=torch.int32).view(KH, 1)
    kw = torch.arange(KW, dtype=torch.int32).view(1, KW)
    shift = ((kh * Wp + kw) * C).reshape(-1).contiguous().to(device)
    return rowtab, shift


def conv_tables_unpadded(B: int, OH: int, OW: int, H: int, W: int, KH: int, KW: int, pt: int, pl: int, C: int, device,
                         pool_order: bool = False, strides=(1, 1)):
    """``(rowtab, shift, masktab)`` for gathering patches from an image that is NOT zero-padded: window origins (negative
    at the top / left border), window-position offsets, and per output pixel the bit set of window positions that lie
    inside the image (the GEMM loader reads the others as zero).  Needs ``KH * KW <= 32``.  ``pool_order`` numbers the
    output pixels window by window of a 2x2 pooling (four consecutive rows = one window; even ``OH``, ``OW``)."""
    rowtab = torch.empty(B * OH * OW, dtype=torch.int32, device=device)
    masktab = torch.empty(B * OH * OW, dtype=torch.int32, device=device)
    rc = nv.kernels().dtf_conv_rowtab_masked(nv.ptr(rowtab), nv.ptr(masktab), B, OH, OW, H, W, KH, KW, pt, pl, C,
                                             int(pool_order), int(strides[0]), int(strides[1]), nv.stream_ptr())
    nv.check(rc, "dtf_conv_rowtab_masked")
    kh = torch.arange(KH, dtype=torch.int32).view(KH, 1)
    kw = torch.arange(KW, dtype=torch.int32).view(1, KW)
    shift = ((kh * W + kw) * C).reshape(-1).contiguous().to(device)
    return rowtab, shift, masktab


def conv_flip_weights(w, out=None):
    """``[KH, KW, C, N]`` filter -> ``[(KH*KW*N), C]`` matrix of the data-gradient convolution (spatially flipped,
    channels transposed)."""
    KH, KW, C, N = w.shape
    if out is None:
        out = torch.empty((KH * KW * N, C), dtype=torch.float32, device=w.device)
    nv.require_f32_contig(w, out)
    rc = nv.kernels().dtf_conv_flip_weights(nv.ptr(w), nv.ptr(out), KH, KW, C, N, nv.stream_ptr())
    nv.check(rc, "dtf_conv_flip_weights")
    return out


def conv_gemm_forward(xpad, rowtab, shift, w2d, bias, act, out, masktab=None, pool_arg=None):
    """``out[M, N] = act(patches(xpad) @ w2d + bias)`` with the patches gathered inside the GEMM (``masktab``: the image
    is not padded, see ``conv_tables_unpadded``).  With ``pool_arg`` (uint8 ``[M/4, N]``) and pool-ordered tables,
    ``out`` is the 2x2-max-pooled result ``[M/4, N]`` and ``pool_arg`` its arg-max positions."""
    K, N = w2d.shape
    M = rowtab.numel()
    C = xpad.shape[-1]
    nv.require_f32_contig(xpad, w2d, bias, out)
    if act == ACT_TANH:
        if pool_arg is not None:
            raise ValueError("the pooling epilogue fuses relu / linear convolutions only")
        conv_gemm_forward(xpad, rowtab, shift, w2d, bias, ACT_NONE, out, masktab)
        return activation_forward(out, ACT_TANH, out=out)
    rc = nv.kernels().dtf_conv_gemm_fwd(nv.ptr(xpad), xpad.numel(), nv.ptr(rowtab), nv.ptr(shift), C, nv.ptr(w2d), nv.ptr(bias),
                                        nv.ptr(out), M, N, K, int(act), nv.ptr(masktab), nv.ptr(pool_arg),
                                        nv.stream_ptr())
    nv.check(rc, "dtf_conv_gemm_fwd")
    return out


def conv_gemm_wgrad(xpad, rowtab, shift, dz, dw2d, db, workspace, masktab=None):
    """``dw2d[K, N] = patches(xpad)^T @ dz``, ``db = colsum(dz)``."""
    K, N = dw2d.shape
    M = rowtab.numel()
    nv.require_f32_contig(xpad, dz, dw2d, db, workspace)
    need = dense_dw_workspace_floats(M, N, K)
    if workspace is None or workspace.numel() < max(need, 1):
        raise ValueError("conv wgrad workspace needs %d floats" % need)
    rc = nv.kernels().dtf_conv_gemm_wgrad(nv.ptr(xpad), xpad.numel(), nv.ptr(rowtab), nv.ptr(shift), xpad.shape[-1], nv.ptr(dz),
                                          nv.ptr(dw2d), nv.ptr(db), M, N, K, nv.ptr(workspace), nv.ptr(masktab), nv.stream_ptr())
    nv.check(rc, "dtf_conv_gemm_wgrad")


def conv_gemm_dgrad(dzpad, rowtab, shift, wflip, dx, yprev=None, masktab=None):
    """``dx[M, C] = patches(dzpad) @ wflip`` (optionally times the ReLU gradient of ``yprev``)."""
    K, C = wflip.shape
    M = rowtab.numel()
    nv.require_f32_contig(dzpad, wflip, dx, yprev)
    rc = nv.kernels().dtf_conv_gemm_dgrad(nv.ptr(dzpad), dzpad.numel(), nv.ptr(rowtab), nv.ptr(shift), dzpad.shape[-1],
                                          nv.ptr(wflip), nv.ptr(dx), M, C, K, nv.ptr(yprev),
                                          ACT_RELU if yprev is not None else ACT_NONE, nv.ptr(masktab), nv.stream_ptr())
    nv.check(rc, "dtf_conv_gemm_dgrad")
    return dx

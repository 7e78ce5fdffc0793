"""Keras-style layers backed by the HIP kernels.

API surface of /root/reference/example2.py:151-156 (``Dense(units, input_dim=, activation=)``,
``Dropout(rate)``) and of the functional use at /root/reference/example.py:150-154.

Layers implement an explicit ``forward`` / ``backward`` pair (a static tape: the model
calls them in order and in reverse), which keeps a training step a fixed sequence of
kernel launches that a hipGraph can capture.  ``Sequential`` tells neighbouring layers
about each other so that elementwise work rides in GEMM epilogues:

* ``Dense`` + following ``Dropout`` -> dropout applied in the forward GEMM epilogue;
* the dgrad GEMM of layer l+1 applies layer l's ReLU/dropout (or sigmoid) gradient in
  its epilogue, so layer l receives dL/dz directly (``grad_is_dz``).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from .ops import functional as F
from .ops._native import ACT_CODES, ACT_NONE, ACT_RELU, ACT_SIGMOID
from .variables import VariableStore

_seed_counter = [0x1234567]


def _next_seed() -> int:
    _seed_counter[0] = (_seed_counter[0] * 1103515245 + 12345) & 0x7FFFFFFF
    return _seed_counter[0]


class Layer:
    prefix = "layer"

    def __init__(self, name: Optional[str] = None, input_shape=None, input_dim=None):
        self.name = name
        self.input_shape: Optional[Tuple[int, ...]] = None
        if input_dim is not None:
            self.input_shape = (int(input_dim),)
        elif input_shape is not None:
            self.input_shape = tuple(int(s) for s in input_shape)
        self.output_shape: Optional[Tuple[int, ...]] = None
        self.built = False
        self.store: Optional[VariableStore] = None
        # set by Sequential planning
        self.grad_is_dz = False      # incoming gradient already includes this layer's activation gradient
        self.need_input_grad = True

    def build(self, input_shape, store: VariableStore):
        self.store = store
        if self.name is None:
            self.name = store.unique_layer_name(self.prefix)
        self.input_shape = tuple(input_shape)
        self.output_shape = self.compute_output_shape(self.input_shape)
        self.built = True
        return self.output_shape

    def compute_output_shape(self, input_shape):
        return tuple(input_shape)

    @property
    def variables(self):
        return []

    def forward(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        raise NotImplementedError

    def backward(self, dy: torch.Tensor) -> Optional[torch.Tensor]:
        raise NotImplementedError

    def get_config(self):
        return {"class": type(self).__name__, "name": self.name}


class _Buffers:
    """Per-layer activation buffers keyed by (tag, shape): steady-state steps allocate nothing."""

    def __init__(self):
        self._b = {}

    def get(self, tag: str, shape, device) -> torch.Tensor:
        key = (tag, tuple(shape), str(device))
        t = self._b.get(key)
        if t is None:
            t = torch.empty(tuple(shape), dtype=torch.float32, device=device)
            self._b[key] = t
        return t

    def clear(self):
        self._b.clear()


class Dense(Layer):
    """``y = activation(x @ kernel + bias)``; kernel is [in, out] as in TF/Keras checkpoints."""

    prefix = "dense"

    def __init__(self, units: int, activation=None, use_bias: bool = True, input_dim: Optional[int] = None,
                 input_shape=None, kernel_initializer="glorot_uniform", bias_initializer="zeros", name=None):
        super().__init__(name=name, input_shape=input_shape, input_dim=input_dim)
        self.units = int(units)
        if activation == "softmax":
            self.softmax = True
            activation = None
        else:
            self.softmax = False
        if activation not in ACT_CODES:
            raise ValueError("unsupported activation %r (have: relu, sigmoid, tanh, softmax, linear)" % (activation,))
        self.activation = activation
        self.act = ACT_CODES[activation]
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.kernel = None
        self.bias = None
        # explicit variable names (a graph written with tf.Variable / tf.matmul keeps the names of its variables in
        # checkpoints); None = Keras naming, "<layer>/kernel" and "<layer>/bias"
        self.kernel_name: Optional[str] = None
        self.bias_name: Optional[str] = None
        self.fused_keep = 1.0        # keep-probability of a Dropout folded into the forward epilogue
        self.seed = _next_seed()
        # producing layer's gradient folded into this layer's dgrad epilogue
        self.prev_act = ACT_NONE
        self.prev_scale = 1.0
        self.fuse_prev_grad = False
        self.emit_logits = False     # softmax head feeding a fused cross-entropy loss
        self._buf = _Buffers()
        self._ws = None
        self._x = self._y = None
        self._keep_used = 1.0

    def get_config(self):
        return {"class": "Dense", "name": self.name, "units": self.units,
                "activation": "softmax" if self.softmax else self.activation, "use_bias": self.use_bias,
                "kernel_initializer": self.kernel_initializer, "bias_initializer": self.bias_initializer}

    def compute_output_shape(self, input_shape):
        if len(input_shape) != 1:
            raise ValueError("Dense expects flat inputs, got per-example shape %s (add Flatten())" % (input_shape,))
        return (self.units,)

    def build(self, input_shape, store):
        out = super().build(input_shape, store)
        k = int(input_shape[0])
        self.kernel = store.create(self.kernel_name or self.name + "/kernel", (k, self.units), self.kernel_initializer)
        if self.use_bias:
            self.bias = store.create(self.bias_name or self.name + "/bias", (self.units,), self.bias_initializer)
        return out

    @property
    def variables(self):
        return [v for v in (self.kernel, self.bias) if v is not None]

    def forward(self, x, training):
        M = x.shape[0]
        keep = self.fused_keep if training else 1.0
        y = self._buf.get("y", (M, self.units), x.device)
        ws = None
        # opt-in mixed precision (compile(matmul_dtype="bf16")): wide layers multiply bf16-rounded operands
        bf16 = getattr(self, "matmul_dtype", "fp32") == "bf16" and x.is_cuda and \
            F.dense_bf16_supported(M, self.units, x.shape[1])
        self._bf16_used = bf16
        if x.is_cuda and not bf16:
            need = F.dense_fwd_workspace_floats(M, self.units, x.shape[1])      # long K, few output tiles: K is split
            if need:
                ws = self._buf.get("fwd_ws", (need,), x.device)
        F.dense_forward(x, self.kernel.tensor, self.bias.tensor if self.bias is not None else None, self.act,
                        keep, self.seed, self.store.global_step, out=y, workspace=ws, bf16=bf16)
        self._x, self._y, self._keep_used = x, y, keep
        if self.softmax and not self.emit_logits:
            # probabilities are only materialised when no fused cross-entropy consumes the logits
            p = self._buf.get("p", (M, self.units), x.device)
            return F.softmax(y, out=p)
        return y

    def backward(self, dy):
        x, y = self._x, self._y
        M = x.shape[0]
        if self.softmax and not self.emit_logits:
            raise NotImplementedError("softmax activation trains through (sparse_)categorical_crossentropy only")
        scale = 1.0 / self._keep_used
        if self.grad_is_dz or (self.act == ACT_NONE and self._keep_used >= 1.0):
            dz = dy
        else:
            if self.act == ACT_NONE and self._keep_used < 1.0:
                raise RuntimeError("dropout fused into a linear Dense needs the stand-alone Dropout layer")
            dz = F.act_grad(dy, y, self.act, scale, out=self._buf.get("dz", y.shape, y.device))
        K = x.shape[1]
        if self._ws is None:
            need = F.dense_dw_workspace_floats(M, self.units, K) if x.is_cuda else 0
            self._ws = (M, torch.empty(max(need, 1), dtype=torch.float32, device=x.device))
        elif self._ws[0] != M:
            need = F.dense_dw_workspace_floats(M, self.units, K) if x.is_cuda else 0
            if need > self._ws[1].numel():
                self._ws = (M, torch.empty(need, dtype=torch.float32, device=x.device))
            else:
                self._ws = (M, self._ws[1])
        bf16 = getattr(self, "_bf16_used", False)
        F.dense_backward_dw(x, dz, self.kernel.grad, self.bias.grad if self.bias is not None else None,
                            workspace=self._ws[1], bf16=bf16)
        if not self.need_input_grad:
            return None
        dx = self._buf.get("dx", x.shape, x.device)
        F.dense_backward_dx(dz, self.kernel.tensor, x if self.fuse_prev_grad else None, self.prev_act,
                            self.prev_scale, out=dx, bf16=bf16)
        return dx


class Dropout(Layer):
    """Inverted dropout, active only in training (Keras learning-phase semantics,
    /root/reference/example.py:213 feeds ``K.learning_phase(): 1``, :225 feeds 0)."""

    prefix = "dropout"

    def __init__(self, rate: float, name=None, seed: Optional[int] = None):
        super().__init__(name=name)
        if not 0.0 <= rate < 1.0:
            raise ValueError("dropout rate must be in [0, 1)")
        self.rate = float(rate)
        self.seed = _next_seed() if seed is None else int(seed)
        self.fused_into_previous = False
        self._buf = _Buffers()
        self._active = False

    def get_config(self):
        return {"class": "Dropout", "name": self.name, "rate": self.rate, "seed": self.seed}

    def forward(self, x, training):
        self._active = training and self.rate > 0.0 and not self.fused_into_previous
        if not self._active:
            return x
        y = self._buf.get("y", x.shape, x.device)
        return F.dropout(x, 1.0 - self.rate, self.seed, self.store.global_step, out=y)

    def backward(self, dy):
        if not self._active:
            return dy
        dx = self._buf.get("dx", dy.shape, dy.device)
        return F.dropout(dy, 1.0 - self.rate, self.seed, self.store.global_step, out=dx)


class Activation(Layer):
    prefix = "activation"

    def __init__(self, activation, name=None):
        super().__init__(name=name)
        if activation not in ("relu", "sigmoid", "tanh"):
            raise ValueError("Activation layer supports relu / sigmoid / tanh")
        self.activation = activation
        self.act = ACT_CODES[activation]
        self._buf = _Buffers()

    def get_config(self):
        return {"class": "Activation", "name": self.name, "activation": self.activation}

    def forward(self, x, training):
        y = self._buf.get("y", x.shape, x.device)
        F.activation_forward(x, self.act, out=y)
        self._y = y
        return y

    def backward(self, dy):
        return F.act_grad(dy, self._y, self.act, 1.0, out=self._buf.get("dx", dy.shape, dy.device))


class BatchNormalization(Layer):
    """Keras ``BatchNormalization(axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True)``: every channel of
    the last axis is normalised with the statistics of the batch while training (learning phase 1) and with the moving
    averages otherwise.  Variables as Keras names them: ``gamma`` / ``beta`` (trained), ``moving_mean`` /
    ``moving_variance`` (not trained; they start at 0 / 1 and follow the batch statistics with ``momentum``).  Works
    on dense activations ``[B, C]`` and feature maps ``[B, H, W, C]`` alike (rows = everything but the last axis).
    Under the cluster strategies the moving averages are each replica's own, like the batch statistics."""

    prefix = "batch_normalization"

    def __init__(self, axis: int = -1, momentum: float = 0.99, epsilon: float = 1e-3, center: bool = True,
                 scale: bool = True, name=None, input_shape=None):
        super().__init__(name=name, input_shape=input_shape)
        if axis not in (-1, 3, 1):
            raise ValueError("BatchNormalization normalises the last axis (axis=-1)")
        self.axis = axis
        self.momentum, self.epsilon = float(momentum), float(epsilon)
        self.center, self.scale = bool(center), bool(scale)
        self.gamma = self.beta = self.moving_mean = self.moving_variance = None
        self._buf = _Buffers()
        self._x = None
        self._trained = False

    def get_config(self):
        return {"class": "BatchNormalization", "name": self.name, "axis": -1, "momentum": self.momentum,
                "epsilon": self.epsilon, "center": self.center, "scale": self.scale}

    def build(self, input_shape, store):
        out = super().build(input_shape, store)
        if self.axis != -1 and self.axis != len(input_shape):
            raise ValueError("BatchNormalization normalises the last axis; got axis=%d for per-example shape %s"
                             % (self.axis, tuple(input_shape)))
        c = int(input_shape[-1])
        if self.scale:
            self.gamma = store.create(self.name + "/gamma", (c,), "ones")
        if self.center:
            self.beta = store.create(self.name + "/beta", (c,), "zeros")
        self.moving_mean = store.create(self.name + "/moving_mean", (c,), "zeros", trainable=False)
        self.moving_variance = store.create(self.name + "/moving_variance", (c,), "ones", trainable=False)
        return out

    @property
    def variables(self):
        return [v for v in (self.gamma, self.beta) if v is not None]

    @property
    def non_trainable_variables(self):
        return [self.moving_mean, self.moving_variance]

    def _workspace(self, R, C, device):
        if device.type != "cuda":
            return None
        return self._buf.get("ws", (F.batchnorm_workspace_floats(R, C),), device)

    def forward(self, x, training):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        R = x2.shape[0]
        y = self._buf.get("y", x2.shape, x.device)
        mean = self._buf.get("mean", (C,), x.device)
        inv = self._buf.get("inv_std", (C,), x.device)
        F.batchnorm_forward(x2, self.gamma.tensor if self.gamma is not None else None,
                            self.beta.tensor if self.beta is not None else None, self.moving_mean.tensor,
                            self.moving_variance.tensor, self.momentum, self.epsilon, training, y, mean, inv,
                            self._workspace(R, C, x.device))
        self._x, self._trained = x2, bool(training)
        return y.view(x.shape)

    def backward(self, dy):
        x2 = self._x
        R, C = x2.shape
        dy2 = dy.reshape(R, C)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        gamma = self.gamma.tensor if self.gamma is not None else None
        if not self._trained:
            # learning phase 0 with a training op: the layer is the affine map of the moving statistics
            inv = 1.0 / torch.sqrt(self.moving_variance.tensor + self.epsilon)
            xhat = (x2 - self.moving_mean.tensor) * inv
            if self.gamma is not None:
                self.gamma.grad.copy_((dy2 * xhat).sum(dim=0))
            if self.beta is not None:
                self.beta.grad.copy_(dy2.sum(dim=0))
            if not self.need_input_grad:
                return None
            return (dy2 * (inv if gamma is None else gamma * inv)).view(dy.shape)
        dx = self._buf.get("dx", x2.shape, x2.device) if self.need_input_grad else None
        F.batchnorm_backward(x2, dy2, gamma, self._buf.get("mean", (C,), x2.device), self._buf.get("inv_std", (C,), x2.device),
                             self.gamma.grad if self.gamma is not None else None,
                             self.beta.grad if self.beta is not None else None, dx, self._workspace(R, C, x2.device))
        return dx.view(dy.shape) if dx is not None else None


class Flatten(Layer):
    prefix = "flatten"

    def compute_output_shape(self, input_shape):
        return (int(math.prod(input_shape)),)

    def forward(self, x, training):
        self._in_shape = x.shape
        return x.reshape(x.shape[0], -1)

    def backward(self, dy):
        return dy.reshape(self._in_shape)


def _pair(v) -> Tuple[int, int]:
    if isinstance(v, (tuple, list)):
        if len(v) != 2:
            raise ValueError("expected an int or a pair, got %r" % (v,))
        return int(v[0]), int(v[1])
    return int(v), int(v)


class Conv2D(Layer):
    """NHWC 2-D convolution, ``y = activation(conv(x, kernel) + bias)``; kernel is ``[kh, kw, cin, cout]`` as in
    TF/Keras checkpoints.

    Executed as patch matrix x filter matrix on the MFMA GEMM kernels: the kernel variable viewed as
    ``[kh*kw*cin, cout]`` is the GEMM's B operand as stored, the GEMM output ``[B*OH*OW, cout]`` is the NHWC result as
    stored, and bias / activation / bias-gradient ride in the GEMM epilogues exactly as for ``Dense``.  On the GPU
    with stride 1 and channel counts that are multiples of 4 the patch matrix is implicit: the GEMM's operand loader
    gathers patch rows from a zero-padded copy of the image (forward and weight gradient), and the data gradient is
    the same kind of convolution over the zero-padded output gradient with the flipped filter.  Other shapes
    materialise the patch matrix, keep it from forward to backward (wgrad reads it) and reuse it as the buffer of
    the patch gradients.
    """

    prefix = "conv2d"

    def __init__(self, filters: int, kernel_size, strides=(1, 1), padding: str = "valid", activation=None,
                 use_bias: bool = True, input_shape=None, kernel_initializer="glorot_uniform", bias_initializer="zeros",
                 name=None):
        super().__init__(name=name, input_shape=input_shape)
        self.filters = int(filters)
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        self.padding = str(padding).lower()
        if self.padding not in ("valid", "same"):
            raise ValueError("padding must be 'valid' or 'same'")
        if activation not in ACT_CODES:
            raise ValueError("unsupported activation %r (have: relu, sigmoid, tanh, linear)" % (activation,))
        self.activation = activation
        self.act = ACT_CODES[activation]
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.kernel = None
        self.bias = None
        self.fuse_prev_relu = False      # producing layer's ReLU gradient folded into this layer's col2im
        self.fused_pool = None           # the 2x2 MaxPooling2D that follows, when the pair may run as one kernel
        self._fused_ran = False          # ran as the first-layer conv+pool kernel
        self._gemm_pool_ran = False      # ran as the implicit GEMM with the pooling in its epilogue
        self._buf = _Buffers()
        self._ws = None
        self._x = self._y = self._cols = None
        self._geo = {}
        self._used_implicit = False

    def compute_output_shape(self, input_shape):
        if len(input_shape) != 3:
            raise ValueError("Conv2D expects per-example shape (H, W, C), got %s" % (input_shape,))
        H, W, _ = input_shape
        OH, OW = F.conv_geometry(H, W, *self.kernel_size, self.strides, self.padding)[:2]
        if OH <= 0 or OW <= 0:
            raise ValueError("kernel %s does not fit input %s" % (self.kernel_size, input_shape))
        return (OH, OW, self.filters)

    def build(self, input_shape, store):
        out = super().build(input_shape, store)
        kh, kw = self.kernel_size
        self.kernel = store.create(self.name + "/kernel", (kh, kw, int(input_shape[2]), self.filters),
                                   self.kernel_initializer)
        if self.use_bias:
            self.bias = store.create(self.name + "/bias", (self.filters,), self.bias_initializer)
        return out

    @property
    def variables(self):
        return [v for v in (self.kernel, self.bias) if v is not None]

    def _patches(self, x):
        kh, kw = self.kernel_size
        B, H, W, C = x.shape
        if (kh, kw) == (1, 1) and self.strides == (1, 1):
            return x.reshape(B * H * W, C)                       # a 1x1 convolution is a plain GEMM over pixels
        OH, OW, _ = self.output_shape
        cols = self._buf.get("cols", (B * OH * OW, kh * kw * C), x.device)
        return F.im2col(x, kh, kw, self.strides, self.padding, out=cols)

    # ---- implicit-GEMM path (GPU, 4-channel multiples, any strides): the GEMM kernels gather patch rows straight from
    #      the image, so no [B*OH*OW, kh*kw*cin] patch matrix ever exists in HBM for the forward pass and the filter
    #      gradient; the data gradient is gathered the same way at stride 1 and goes through col2im otherwise
    def _implicit(self, x) -> bool:
        if not x.is_cuda or self.kernel_size == (1, 1):
            return False
        need_dgrad = self.need_input_grad
        return F.conv_implicit_supported(x.shape[3], self.filters, self.strides, need_dgrad)

    def _geometry(self, x):
        """Per-batch-size tables and buffers of the implicit path (built once, reused every step)."""
        B, H, W, C = x.shape
        key = (B, str(x.device))
        g = self._geo.get(key)
        if g is None:
            kh, kw = self.kernel_size
            OH, OW, pt, pl, pb, pr = F.conv_geometry(H, W, kh, kw, self.strides, self.padding)
            Hp, Wp = H + pt + pb, W + pl + pr
            # windows of up to 32 positions: patches are gathered straight from the activations as they are stored and a
            # per-pixel bit set tells the loader which taps fall outside the image - no zero-padded copy is written
            # (for the MNIST CNN that removes three memory-bound passes per step); larger windows go through a copy
            g = {"pt": pt, "pl": pl, "Hp": Hp, "Wp": Wp,
                 "masked": self._masked_tables_fit(B, H, W, C, OH, OW, self.filters, kh, kw, self.need_input_grad)}
            if g["masked"]:
                g["rowtab"], g["shift"], g["mask"] = F.conv_tables_unpadded(B, OH, OW, H, W, kh, kw, pt, pl, C, x.device,
                                                                            strides=self.strides)
                if self.strides == (1, 1) and self._pool_in_epilogue_ok(OH, OW):
                    # same patches, numbered pooling window by pooling window: the forward GEMM pools in its epilogue
                    g["rowtab_pool"], _, g["mask_pool"] = F.conv_tables_unpadded(B, OH, OW, H, W, kh, kw, pt, pl, C,
                                                                                 x.device, pool_order=True)
            else:
                g["xpad"] = torch.empty((B, Hp, Wp, C), dtype=torch.float32, device=x.device)
                g["rowtab"], g["shift"] = F.conv_tables(B, OH, OW, Hp, Wp, kh, kw, C, x.device, strides=self.strides)
                g["mask"] = None
            if self.need_input_grad and self.strides == (1, 1):
                N = self.filters
                g["top"], g["left"], g["Hq"], g["Wq"] = kh - 1 - pt, kw - 1 - pl, H + kh - 1, W + kw - 1
                if g["masked"]:
                    g["rowtab_d"], g["shift_d"], g["mask_d"] = F.conv_tables_unpadded(B, H, W, OH, OW, kh, kw, g["top"],
                                                                                      g["left"], N, x.device)
                else:
                    g["dzpad"] = torch.empty((B, g["Hq"], g["Wq"], N), dtype=torch.float32, device=x.device)
                    g["rowtab_d"], g["shift_d"] = F.conv_tables(B, H, W, g["Hq"], g["Wq"], kh, kw, N, x.device)
                    g["mask_d"] = None
                g["wflip"] = torch.empty((kh * kw * N, C), dtype=torch.float32, device=x.device)
            self._geo[key] = g
        return g

    def forward(self, x, training):
        if x.dim() != 4:
            raise ValueError("Conv2D expects NHWC input, got shape %s" % (tuple(x.shape),))
        x = x.contiguous()
        B = x.shape[0]
        OH, OW, Fo = self.output_shape
        bias = self.bias.tensor if self.bias is not None else None
        self._fused_ran = self._fusable_with_pool(x)
        if self._fused_ran:
            # first layer + 2x2 pooling in one kernel: only the pooled tensor and the arg-max leave the chip; the
            # pooling layer passes that tensor (and, backwards, the pooled gradient) through untouched
            pool = self.fused_pool
            shape = (B, OH // 2, OW // 2, Fo)
            ypool = pool._buf.get("y", shape, x.device)
            arg = pool._arg_buffer(shape, x.device)
            kh, kw = self.kernel_size
            xin, self._fused_padding = self._fused_input(x)
            F.conv_pool_forward(xin, self.kernel.tensor.view(-1, Fo), bias, self.act, kh, kw, self._fused_padding, ypool, arg)
            self._x, self._cols, self._y = xin, None, ypool
            pool._x, pool._arg = None, arg
            return ypool
        self._used_implicit = self._implicit(x)
        self._gemm_pool_ran = False
        if self._used_implicit:
            g = self._geometry(x)
            if "rowtab_pool" in g:
                # convolution + bias + ReLU + 2x2 max pooling in one GEMM: only the pooled tensor and the arg-max
                # leave the chip (the full-resolution activation, four times the size, is never written)
                pool = self.fused_pool
                shape = (B, OH // 2, OW // 2, Fo)
                ypool = pool._buf.get("y", shape, x.device)
                arg = pool._arg_buffer(shape, x.device)
                F.conv_gemm_forward(x, g["rowtab_pool"], g["shift"], self.kernel.tensor.view(-1, Fo), bias, self.act,
                                    ypool.view(-1, Fo), masktab=g["mask_pool"], pool_arg=arg)
                self._x, self._cols, self._y = x, None, None
                pool._x, pool._arg, pool._ypool = None, arg, ypool
                self._gemm_pool_ran = True
                return ypool
        y = self._buf.get("y", (B * OH * OW, Fo), x.device)
        if self._used_implicit:
            src = x if g["masked"] else F.conv_pad(x, g["pt"], g["pl"], g["Hp"], g["Wp"], out=g["xpad"])
            F.conv_gemm_forward(src, g["rowtab"], g["shift"], self.kernel.tensor.view(-1, Fo), bias, self.act, y,
                                masktab=g["mask"])
            cols = None
        else:
            cols = self._patches(x)
            F.dense_forward(cols, self.kernel.tensor.view(-1, Fo), bias, self.act, out=y)
        self._x, self._cols, self._y = x, cols, y
        return y.view(B, OH, OW, Fo)

    @staticmethod
    def _masked_tables_fit(B, H, W, C, OH, OW, N, kh, kw, need_input_grad) -> bool:
        """May the implicit GEMM gather from the activations as stored (per-pixel bit set of in-image taps)?  Windows
        of up to 32 positions only, and the gathered tensor - the input for forward / wgrad, the output gradient for
        dgrad - must stay below 2^30 elements (the unpadded tables hold 30-bit element offsets; the padded-copy path
        takes twice that)."""
        if kh * kw > 32:
            return False
        if B * H * W * C >= (1 << 30):
            return False
        return not (need_input_grad and B * OH * OW * N >= (1 << 30))

    def _pool_in_epilogue_ok(self, OH: int, OW: int) -> bool:
        """May the 2x2 pooling that follows run in the epilogue of this layer's implicit GEMM?"""
        return (self.fused_pool is not None and OH % 2 == 0 and OW % 2 == 0 and self.filters % 4 == 0
                and self.act in (ACT_NONE, ACT_RELU) and (self.grad_is_dz or self.act == ACT_NONE))

    def _fusable_with_pool(self, x) -> bool:
        return (self.fused_pool is not None and x.is_cuda and not self.need_input_grad and self.act in (ACT_NONE, ACT_RELU)
                and F.conv_pool_supported(x.shape, *self.kernel_size, self.filters, self.strides, self.padding,
                                          self.fused_pool.pool_size))

    def _fused_input(self, x):
        """(input, padding mode) for the fused kernels.  Their staged form copies whole images into LDS and pads there,
        so it reads the input as stored; the direct form runs a VALID convolution on a copy padded once per step (a few
        MB for a first layer), which frees every tap from bounds tests."""
        B, H, W, C = x.shape
        kh, kw = self.kernel_size
        OH, OW, pt, pl, pb, pr = F.conv_geometry(H, W, kh, kw, self.strides, self.padding)
        if pt == pl == pb == pr == 0:
            return x, "valid"
        if F.conv_pool_staged_fits(B, H, W, C, kh, kw, pt, pl, OH // 2, OW // 2, self.filters):
            return x, self.padding
        xpad = self._buf.get("xpad_fused", (B, H + pt + pb, W + pl + pr, C), x.device)
        return F.conv_pad(x, pt, pl, H + pt + pb, W + pl + pr, out=xpad), "valid"

    def _backward_fused_pool(self, dpool):
        x, ypool = self._x, self._y
        kh, kw = self.kernel_size
        Fo = self.filters
        need = F.conv_pool_workspace_floats(x.shape[3], kh, kw, Fo)
        if self._ws is None or self._ws[1].numel() < need:
            self._ws = (0, torch.empty(need, dtype=torch.float32, device=x.device))
        F.conv_pool_wgrad(x, dpool.contiguous(), ypool, self.fused_pool._arg, self.kernel.grad.view(-1, Fo),
                          self.bias.grad if self.bias is not None else None, self.act, kh, kw, self._fused_padding,
                          self._ws[1])
        return None

    def backward(self, dy):
        if self._fused_ran:
            return self._backward_fused_pool(dy)
        x, cols, y = self._x, self._cols, self._y
        OHo, OWo, Fo = self.output_shape
        M = x.shape[0] * OHo * OWo
        kh, kw = self.kernel_size
        K = kh * kw * x.shape[3]
        dy2 = dy.reshape(M, Fo)
        if self.grad_is_dz or self.act == ACT_NONE:
            dz = dy2 if dy2.is_contiguous() else dy2.contiguous()
        else:
            dz = F.act_grad(dy2.contiguous(), y, self.act, 1.0, out=self._buf.get("dz", y.shape, y.device))
        if self._ws is None or self._ws[0] != M:
            need = F.dense_dw_workspace_floats(M, Fo, K) if x.is_cuda else 0
            if self._ws is None or need > self._ws[1].numel():
                self._ws = (M, torch.empty(max(need, 1), dtype=torch.float32, device=x.device))
            else:
                self._ws = (M, self._ws[1])
        dw2d = self.kernel.grad.view(-1, Fo)
        db = self.bias.grad if self.bias is not None else None
        if self._used_implicit:
            g = self._geometry(x)
            F.conv_gemm_wgrad(x if g["masked"] else g["xpad"], g["rowtab"], g["shift"], dz, dw2d, db, self._ws[1],
                              masktab=g["mask"])
            if not self.need_input_grad:
                return None
            if self.strides != (1, 1):
                # strided: the patch gradients dz . W^T, scattered back (overlapping windows summed) by col2im
                dcols = F.dense_backward_dx(dz, self.kernel.tensor.view(-1, Fo), out=self._buf.get("dcols", (M, K), x.device))
                dx = self._buf.get("dx", x.shape, x.device)
                return F.col2im(dcols, x.shape, kh, kw, self.strides, self.padding, x if self.fuse_prev_relu else None, out=dx)
            B, H, W, C = x.shape
            OH, OW, _ = self.output_shape
            dz4 = dz.view(B, OH, OW, Fo)
            dsrc = dz4 if g["masked"] else F.conv_pad(dz4, g["top"], g["left"], g["Hq"], g["Wq"], out=g["dzpad"])
            F.conv_flip_weights(self.kernel.tensor, out=g["wflip"])
            dx = self._buf.get("dx", (B * H * W, C), x.device)
            F.conv_gemm_dgrad(dsrc, g["rowtab_d"], g["shift_d"], g["wflip"], dx,
                              yprev=x.view(-1, C) if self.fuse_prev_relu else None, masktab=g["mask_d"])
            return dx.view(x.shape)
        F.dense_backward_dw(cols, dz, dw2d, db, workspace=self._ws[1])
        if not self.need_input_grad:
            return None
        if (kh, kw) == (1, 1) and self.strides == (1, 1):
            dx = self._buf.get("dx", (M, K), x.device)
            F.dense_backward_dx(dz, self.kernel.tensor.view(-1, Fo), x.reshape(M, K) if self.fuse_prev_relu else None,
                                ACT_RELU if self.fuse_prev_relu else ACT_NONE, 1.0, out=dx)
            return dx.view(x.shape)
        # patch gradients overwrite the patch matrix (wgrad above was its last reader)
        dcols = F.dense_backward_dx(dz, self.kernel.tensor.view(-1, Fo), out=cols)
        dx = self._buf.get("dx", x.shape, x.device)
        return F.col2im(dcols, x.shape, kh, kw, self.strides, self.padding, x if self.fuse_prev_relu else None, out=dx)

    def get_config(self):
        return {"class": "Conv2D", "name": self.name, "filters": self.filters, "kernel_size": self.kernel_size,
                "strides": self.strides, "padding": self.padding, "activation": self.activation, "use_bias": self.use_bias,
                "kernel_initializer": self.kernel_initializer, "bias_initializer": self.bias_initializer}


class MaxPooling2D(Layer):
    """NHWC max pooling.  The Keras default - ``strides`` = ``pool_size``, ``padding='valid'`` - takes the fast kernels
    (and can run inside the convolution in front); any other strides or ``padding='same'`` take the general ones."""

    prefix = "max_pooling2d"

    def __init__(self, pool_size=(2, 2), strides=None, padding: str = "valid", name=None):
        super().__init__(name=name)
        self.pool_size = _pair(pool_size)
        self.strides = _pair(strides) if strides is not None else self.pool_size
        self.padding = str(padding).lower()
        if self.padding not in ("valid", "same"):
            raise ValueError("MaxPooling2D: padding must be 'valid' or 'same'")
        if self.pool_size[0] * self.pool_size[1] > 255:
            raise ValueError("MaxPooling2D: windows of more than 255 positions are not supported")
        # windows that tile the image: the case the fused / vectorised kernels cover
        self.tiling = self.strides == self.pool_size and self.padding == "valid"
        self.fold_relu_grad = False      # the pooled tensor is a ReLU output whose gradient this layer applies
        self.fused_into = None           # the Conv2D in front, when the pair may run as one kernel
        self._buf = _Buffers()
        self._x = self._arg = None

    def compute_output_shape(self, input_shape):
        if len(input_shape) != 3:
            raise ValueError("MaxPooling2D expects per-example shape (H, W, C), got %s" % (input_shape,))
        H, W, C = input_shape
        if self.tiling:
            OH, OW = H // self.pool_size[0], W // self.pool_size[1]
        else:
            OH, OW = F.pool_geometry(H, W, *self.pool_size, self.strides, self.padding)[:2]
        if OH <= 0 or OW <= 0:
            raise ValueError("pool %s does not fit input %s" % (self.pool_size, input_shape))
        return (OH, OW, C)

    def _arg_buffer(self, shape, device):
        key = ("arg", tuple(shape), str(device))
        arg = self._buf._b.get(key)
        if arg is None:
            arg = torch.empty(tuple(shape), dtype=torch.uint8, device=device)
            self._buf._b[key] = arg
        return arg

    def forward(self, x, training):
        if self.fused_into is not None and (self.fused_into._fused_ran or self.fused_into._gemm_pool_ran):
            return x                     # already pooled by the convolution's kernel
        x = x.contiguous()
        B = x.shape[0]
        shape = (B,) + tuple(self.output_shape)
        y = self._buf.get("y", shape, x.device)
        arg = self._arg_buffer(shape, x.device)
        if self.tiling:
            F.maxpool_forward(x, self.pool_size, out=y, arg=arg)
        else:
            F.maxpool_strided_forward(x, self.pool_size, self.strides, self.padding, out=y, arg=arg)
        self._x, self._arg = x, arg
        return y

    def backward(self, dy):
        if self.fused_into is not None and self.fused_into._fused_ran:
            return dy                    # the convolution's wgrad kernel consumes the pooled gradient
        if self.fused_into is not None and self.fused_into._gemm_pool_ran:
            # pooled in the convolution's GEMM epilogue: the full-resolution activation was never stored; the ReLU
            # mask comes from the pooled tensor (equal to it at the arg-max, the only position with a gradient)
            conv = self.fused_into
            shape = (dy.shape[0],) + tuple(conv.output_shape)
            dx = self._buf.get("dx", shape, dy.device)
            return F.maxpool_backward(dy.contiguous(), self._arg, shape, self.pool_size,
                                      self._ypool if self.fold_relu_grad else None, out=dx, mask_pooled=True)
        x = self._x
        dx = self._buf.get("dx", x.shape, x.device)
        if not self.tiling:
            return F.maxpool_strided_backward(dy.contiguous(), self._arg, x.shape, self.pool_size, self.strides, self.padding,
                                              x if self.fold_relu_grad else None, out=dx)
        return F.maxpool_backward(dy.contiguous(), self._arg, x.shape, self.pool_size,
                                  x if self.fold_relu_grad else None, out=dx)

    def get_config(self):
        return {"class": "MaxPooling2D", "name": self.name, "pool_size": self.pool_size, "strides": self.strides,
                "padding": self.padding}


class AveragePooling2D(Layer):
    """NHWC average pooling (Keras ``AveragePooling2D``): any window and strides, ``valid`` or ``same`` padding; windows
    over the border average over their in-image positions."""

    prefix = "average_pooling2d"

    def __init__(self, pool_size=(2, 2), strides=None, padding: str = "valid", name=None):
        super().__init__(name=name)
        self.pool_size = _pair(pool_size)
        self.strides = _pair(strides) if strides is not None else self.pool_size
        self.padding = str(padding).lower()
        if self.padding not in ("valid", "same"):
            raise ValueError("AveragePooling2D: padding must be 'valid' or 'same'")
        self._buf = _Buffers()
        self._in_shape = None

    def get_config(self):
        return {"class": type(self).__name__, "name": self.name, "pool_size": list(self.pool_size),
                "strides": list(self.strides), "padding": self.padding}

    def compute_output_shape(self, input_shape):
        if len(input_shape) != 3:
            raise ValueError("%s expects per-example shape (H, W, C), got %s" % (type(self).__name__, input_shape))
        H, W, C = input_shape
        OH, OW = F.pool_geometry(H, W, *self.pool_size, self.strides, self.padding)[:2]
        if OH <= 0 or OW <= 0:
            raise ValueError("pool %s does not fit input %s" % (self.pool_size, input_shape))
        return (OH, OW, C)

    def forward(self, x, training):
        x = x.contiguous()
        self._in_shape = tuple(x.shape)
        y = self._buf.get("y", (x.shape[0],) + tuple(self.output_shape), x.device)
        return F.avgpool_forward(x, self.pool_size, self.strides, self.padding, out=y)

    def backward(self, dy):
        dx = self._buf.get("dx", self._in_shape, dy.device)
        return F.avgpool_backward(dy.contiguous(), self._in_shape, self.pool_size, self.strides, self.padding, out=dx)


class GlobalAveragePooling2D(Layer):
    """Mean over the spatial axes: ``[B, H, W, C] -> [B, C]`` (one averaging window as large as the feature map)."""

    prefix = "global_average_pooling2d"

    def __init__(self, name=None):
        super().__init__(name=name)
        self._buf = _Buffers()
        self._in_shape = None

    def compute_output_shape(self, input_shape):
        if len(input_shape) != 3:
            raise ValueError("GlobalAveragePooling2D expects per-example shape (H, W, C), got %s" % (input_shape,))
        return (int(input_shape[2]),)

    def forward(self, x, training):
        x = x.contiguous()
        B, H, W, C = x.shape
        self._in_shape = (B, H, W, C)
        y = self._buf.get("y", (B, 1, 1, C), x.device)
        return F.avgpool_forward(x, (H, W), (H, W), "valid", out=y).view(B, C)

    def backward(self, dy):
        B, H, W, C = self._in_shape
        dx = self._buf.get("dx", self._in_shape, dy.device)
        return F.avgpool_backward(dy.contiguous().view(B, 1, 1, C), self._in_shape, (H, W), (H, W), "valid", out=dx)


def layer_from_config(cfg: dict) -> Layer:
    """Rebuild a layer from its ``get_config()`` dictionary (``class`` names the layer type)."""
    classes = {c.__name__: c for c in (Dense, Dropout, Activation, Flatten, Conv2D, MaxPooling2D, BatchNormalization,
                                       AveragePooling2D, GlobalAveragePooling2D)}
    kw = {k: v for k, v in cfg.items() if k != "class"}
    try:
        cls = classes[cfg["class"]]
    except KeyError:
        raise ValueError("unknown layer class %r" % (cfg.get("class"),))
    for k in ("kernel_size", "strides", "pool_size", "input_shape"):
        if k in kw and kw[k] is not None:
            kw[k] = tuple(kw[k])
    return cls(**kw)

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
            raise ValueError("unsupported activation %r (have: relu, sigmoid, softmax, linear)" % (activation,))
        self.activation = activation
        self.act = ACT_CODES[activation]
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.kernel = None
        self.bias = None
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

    def compute_output_shape(self, input_shape):
        if len(input_shape) != 1:
            raise ValueError("Dense expects flat inputs, got per-example shape %s (add Flatten())" % (input_shape,))
        return (self.units,)

    def build(self, input_shape, store):
        out = super().build(input_shape, store)
        k = int(input_shape[0])
        self.kernel = store.create(self.name + "/kernel", (k, self.units), self.kernel_initializer)
        if self.use_bias:
            self.bias = store.create(self.name + "/bias", (self.units,), self.bias_initializer)
        return out

    @property
    def variables(self):
        return [v for v in (self.kernel, self.bias) if v is not None]

    def forward(self, x, training):
        M = x.shape[0]
        keep = self.fused_keep if training else 1.0
        y = self._buf.get("y", (M, self.units), x.device)
        F.dense_forward(x, self.kernel.tensor, self.bias.tensor if self.bias is not None else None, self.act,
                        keep, self.seed, self.store.global_step, out=y)
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
        F.dense_backward_dw(x, dz, self.kernel.grad, self.bias.grad if self.bias is not None else None,
                            workspace=self._ws[1])
        if not self.need_input_grad:
            return None
        dx = self._buf.get("dx", x.shape, x.device)
        F.dense_backward_dx(dz, self.kernel.tensor, x if self.fuse_prev_grad else None, self.prev_act,
                            self.prev_scale, out=dx)
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
        if activation not in ("relu", "sigmoid"):
            raise ValueError("Activation layer supports relu / sigmoid")
        self.act = ACT_CODES[activation]
        self._buf = _Buffers()

    def forward(self, x, training):
        y = self._buf.get("y", x.shape, x.device)
        F.activation_forward(x, self.act, out=y)
        self._y = y
        return y

    def backward(self, dy):
        return F.act_grad(dy, self._y, self.act, 1.0, out=self._buf.get("dx", dy.shape, dy.device))


class Flatten(Layer):
    prefix = "flatten"

    def compute_output_shape(self, input_shape):
        return (int(math.prod(input_shape)),)

    def forward(self, x, training):
        self._in_shape = x.shape
        return x.reshape(x.shape[0], -1)

    def backward(self, dy):
        return dy.reshape(self._in_shape)

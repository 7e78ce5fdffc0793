"""``Sequential`` / ``compile`` / ``fit`` / ``evaluate`` / ``predict``.

The Keras workflow of /root/reference/example2.py:151-165,200, made to actually work on
the cluster runtime (the reference's own ``fit`` call sits inside a finalized TF graph).

Execution model: a training step is a fixed list of kernel launches (forward GEMMs with
fused bias/activation/dropout, one fused loss+gradient kernel, backward GEMMs, one fused
apply-gradients launch over the flat variable buffer).  Nothing in a step synchronises
with the host: loss/accuracy accumulate in a device-resident statistics record that is
read once per epoch (or whenever a callback asks), and ``steps_per_graph`` lets ``fit``
replay many steps from one captured hipGraph when batches are launch-bound.
"""
from __future__ import annotations

import math
import os
import time
from typing import Dict, List, Optional, Sequence

import torch

from .. import callbacks as cb
from .. import losses as losses_mod
from .. import optimizers as optim_mod
from ..layers import Conv2D, Dense, Dropout, Layer, MaxPooling2D
from ..ops import functional as F
from ..ops._native import ACT_NONE, ACT_RELU, ACT_SIGMOID
from ..variables import VariableStore

_FUSED_GRAD = os.environ.get("DTF_FUSED_GRAD", "1") != "0"


def nv_kernels():
    from ..ops import _native as nv
    return nv.kernels()


def default_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def _as_tensor(a, device, dtype=None) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        t = a
        # already where and what it should be (the per-step case of a session loop): skip the dispatcher round trip
        if dtype is None and t.dtype in (torch.float32, torch.int64) and t.device.type == device.type \
                and (device.index is None or t.device.index == device.index) and t.is_contiguous() \
                and (t.dtype == torch.float32 or t.dim() == 1):
            return t
    else:
        t = torch.as_tensor(a)
    if dtype is None:
        dtype = torch.int64 if t.dtype in (torch.int64, torch.int32, torch.int16, torch.uint8) and t.dim() == 1 \
            else torch.float32
    return t.to(device=device, dtype=dtype).contiguous()


class LocalStrategy:
    """Single-replica update: apply the whole flat gradient, advance global_step."""

    world_size = 1
    rank = 0

    def prepare(self, model):
        pass

    def apply_gradients(self, model):
        model.optimizer.apply_and_finish(model.store)

    def reduce_stats(self, stats: torch.Tensor) -> torch.Tensor:
        return stats


class Sequential:
    def __init__(self, layers: Optional[Sequence[Layer]] = None, name: str = "sequential"):
        self.name = name
        self.layers: List[Layer] = []
        self.store = VariableStore()
        self.built = False
        self.compiled = False
        self.loss = None
        self.optimizer = None
        self.metrics: List[str] = []
        self.strategy = LocalStrategy()
        self.device: Optional[torch.device] = None
        self.stats: Optional[torch.Tensor] = None
        self.stop_training = False
        self._grad_buf: Dict = {}
        self._host_step = 0           # host mirror of global_step (no device sync needed)
        self._graphs: Dict = {}
        self._fused_plans: Dict = {}
        self._tail_plan = False       # False = not looked at yet
        self.use_tail_kernel = True   # set False to force the layer-by-layer kernels through the network's tail
        self._after_layer_backward = None   # strategy hook: called as each layer's gradients become final
        self.async_workers = 1        # asynchronous workers inside the GPU for the one-launch step kernel
        for layer in layers or []:
            self.add(layer)

    # ------------------------------------------------------------------ construction
    def add(self, layer: Layer):
        if self.built:
            raise RuntimeError("cannot add layers after the model is built")
        if not self.layers and layer.input_shape is None:
            # shape may still arrive through build(input_shape=...)
            pass
        self.layers.append(layer)

    def build(self, input_shape=None, device=None, seed: Optional[int] = None, pad_multiple: int = 64):
        if self.built:
            return self
        if input_shape is None:
            input_shape = self.layers[0].input_shape
        if input_shape is None:
            raise ValueError("first layer needs input_dim= / input_shape= (or pass build(input_shape=...))")
        shape = tuple(input_shape)
        self.input_shape = shape
        for layer in self.layers:
            shape = layer.build(shape, self.store)
        self.output_shape = shape
        if seed is not None:
            # dropout streams follow the model seed, not the process-wide count of layers created so far: a run is
            # reproducible from compile(seed=...) whatever else the process has built before
            for idx, layer in enumerate(self.layers):
                if hasattr(layer, "seed"):
                    layer.seed = (int(seed) * 1000003 + idx * 7919 + 12345) & 0x7FFFFFFF
        self._plan_fusion()
        self.device = torch.device(device) if device is not None else default_device()
        self.store.finalize(self.device, seed=seed, pad_multiple=pad_multiple)
        self.stats = F.new_stats(self.device)
        self.built = True
        return self

    def _plan_fusion(self):
        """Fold Dropout into the preceding ReLU Dense and activation gradients into the next dgrad GEMM."""
        L = self.layers
        for i, layer in enumerate(L):
            if isinstance(layer, Dropout) and i > 0 and isinstance(L[i - 1], Dense) \
                    and L[i - 1].act == ACT_RELU and not L[i - 1].softmax and layer.rate > 0.0:
                L[i - 1].fused_keep = 1.0 - layer.rate
                L[i - 1].seed = layer.seed
                layer.fused_into_previous = True
            elif isinstance(layer, Dropout) and layer.rate == 0.0:
                layer.fused_into_previous = True      # a no-op: leave it out of the effective chain
        # chain of "effective" layers (skipping folded dropouts)
        eff = [l for l in L if not (isinstance(l, Dropout) and l.fused_into_previous)]
        for a, b in zip(eff[:-1], eff[1:]):
            if isinstance(a, Dense) and isinstance(b, Dense) and not a.softmax and a.act in (ACT_RELU, ACT_SIGMOID) \
                    and not (a.act == ACT_SIGMOID and a.fused_keep < 1.0):
                b.fuse_prev_grad = True
                b.prev_act = a.act
                b.prev_scale = 1.0 / a.fused_keep
                a.grad_is_dz = True
            elif isinstance(a, Conv2D) and a.act == ACT_RELU and isinstance(b, MaxPooling2D):
                b.fold_relu_grad = True          # pooling backward also applies the conv's ReLU gradient
                a.grad_is_dz = True
            if isinstance(a, Conv2D) and isinstance(b, MaxPooling2D) and b.pool_size == (2, 2) and b.tiling \
                    and a.act in (ACT_NONE, ACT_RELU):
                # the pair may run with the pooling inside the convolution's kernel (decided per call: GPU, shape):
                # the first-layer kernel of conv_pool.hip, or the pooling epilogue of the implicit-GEMM convolution
                a.fused_pool = b
                b.fused_into = a
            elif isinstance(a, Conv2D) and a.act == ACT_RELU and isinstance(b, Conv2D):
                b.fuse_prev_relu = True          # col2im (or the 1x1 dgrad epilogue) applies it
                a.grad_is_dz = True
        if eff:
            eff[0].need_input_grad = False
        self._effective = eff

    # ------------------------------------------------------------------ compile
    def compile(self, loss="mean_squared_error", optimizer="adam", metrics: Optional[Sequence[str]] = None,
                strategy=None, device=None, seed: Optional[int] = None, matmul_dtype: str = "fp32"):
        """``matmul_dtype="bf16"`` (opt-in mixed precision): dense layers wide enough for the bf16-operand MFMA
        kernels multiply bf16-rounded operands with fp32 accumulation; variables, gradients, optimizer state and all
        narrow layers stay fp32, as in the reference (/root/reference/example.py:149,155: tf.float32 throughout).
        ``matmul_dtype="fp32_split"`` keeps fp32 accuracy but multiplies on the bf16 matrix cores: every dense-layer
        GEMM large enough splits its fp32 operands exactly into three bf16 limbs and accumulates six limb products
        in fp32 (process-wide switch, see ``ops.functional.set_gemm_split_products``; DTF_GEMM_SPLIT=8 or 9 adds the
        2^-24 / 2^-32 terms)."""
        if matmul_dtype not in ("fp32", "bf16", "fp32_split"):
            raise ValueError("matmul_dtype must be 'fp32', 'fp32_split' or 'bf16'")
        self.matmul_dtype = matmul_dtype
        if matmul_dtype == "fp32_split":
            import os as _os
            F.set_gemm_split_products(int(_os.environ.get("DTF_GEMM_SPLIT", "6")))
        else:
            F.set_gemm_split_products(0)
        for l in self.layers:
            if isinstance(l, Dense):
                l.matmul_dtype = matmul_dtype
        self.loss = losses_mod.get(loss)
        self.optimizer = optim_mod.get(optimizer)
        self.metrics = list(metrics or [])
        if strategy is not None:
            self.strategy = strategy
        if not self.built:
            pad = self.strategy.var_alignment() if hasattr(self.strategy, "var_alignment") else 64
            self.build(device=device, seed=seed, pad_multiple=pad)
        head = self._effective[-1] if self._effective else None
        if isinstance(head, Dense):
            if self.loss.wants_logits:
                if head.act != ACT_NONE:
                    raise ValueError("%s expects a linear or softmax head, got activation=%r"
                                     % (self.loss.name, head.activation))
                head.emit_logits = True
            if self.loss.fuse_with_head(head):
                head.grad_is_dz = True
        if not getattr(self.strategy, "owns_optimizer_state", False):
            self.optimizer.create_slots(self.store)      # (parameter-server workers hold none: their owners do)
        self.strategy.prepare(self)
        # Keras rule: 'accuracy' with a non-crossentropy loss and output dim > 1 is categorical accuracy.
        self._extra_categorical = any(m in ("accuracy", "acc", "categorical_accuracy") for m in self.metrics) \
            and self.loss.primary_metric != "categorical"
        self._report_binary = any(m in ("binary_rounded_accuracy", "xor_accuracy", "bit_accuracy")
                                  for m in self.metrics)
        self.compiled = True
        return self

    # ------------------------------------------------------------------ execution pieces
    def forward(self, x: torch.Tensor, training: bool = False) -> torch.Tensor:
        fresh = getattr(self.strategy, "ensure_fresh", None)
        if fresh is not None:
            fresh(self)      # a worker whose last steps read the parameter servers' memory directly: refresh its copy
        h = x
        for layer in self.layers:
            h = layer.forward(h, training)
        return h

    def backward(self, dy: torch.Tensor):
        g = dy
        hook = self._after_layer_backward
        for layer in reversed(self.layers):
            g = layer.backward(g)
            if hook is not None:
                hook(layer, self)
            if g is None:
                break

    def _grad_like(self, out: torch.Tensor) -> torch.Tensor:
        key = (tuple(out.shape), str(out.device))
        t = self._grad_buf.get(key)
        if t is None:
            t = torch.empty_like(out)
            self._grad_buf[key] = t
        return t

    def _loss_and_metrics(self, out, y, want_grad: bool, stats: torch.Tensor):
        g = self._grad_like(out) if want_grad else None
        self.loss(out, y, g, stats)
        if self._extra_categorical:
            F.categorical_accuracy_sum(out, y if y.dim() == 2 else
                                       torch.nn.functional.one_hot(y, out.shape[1]).to(torch.float32),
                                       stats[3:4])
        return g

    # ------------------------------------------------------------------ fused network tail
    def tail_plan(self):
        """(index of the first tail layer, Dense b, Dense c) when the model ends in
        ``Dense(128, relu)[+Dropout] -> Dense(128, relu)[+Dropout] -> Dense(C <= 16, softmax)`` trained with sparse
        softmax cross-entropy on the GPU: everything after the first of those three layers (forward, loss, backward
        down to that layer's dL/dz) then runs in one kernel, ``csrc/kernels/mlp_tail.hip``.  ``None`` otherwise."""
        if self._tail_plan is not False:
            return self._tail_plan
        self._tail_plan = None
        eff = self._effective
        if not self.use_tail_kernel or self.device is None or self.device.type != "cuda" or len(eff) < 3:
            return None
        a, b, c = eff[-3], eff[-2], eff[-1]
        if not (isinstance(a, Dense) and isinstance(b, Dense) and isinstance(c, Dense)):
            return None
        if not isinstance(self.loss, losses_mod.SparseCategoricalCrossentropy) or not c.emit_logits or c.act != ACT_NONE:
            return None
        if a.act != ACT_RELU or a.softmax or b.act != ACT_RELU or b.softmax or not b.fuse_prev_grad or not a.grad_is_dz:
            return None
        if b.bias is None or c.bias is None or c.fused_keep < 1.0 or not c.fuse_prev_grad:
            return None
        ia = self.layers.index(a)
        rest = self.layers[ia + 1:]
        if any(not (l is b or l is c or (isinstance(l, Dropout) and l.fused_into_previous)) for l in rest):
            return None
        if not F.mlp_tail_supported(a.units, b.units, c.units):
            return None
        self._tail_plan = (ia + 1, b, c)
        return self._tail_plan

    def _compute_gradients_tail(self, x, y, stats, plan):
        first, b, c = plan
        h = x
        for layer in self.layers[:first]:
            h = layer.forward(h, True)
        M = h.shape[0]
        key = ("tail", M, str(h.device))
        bufs = self._grad_buf.get(key)
        if bufs is None:
            bufs = (torch.empty((M, b.kernel.shape[0]), dtype=torch.float32, device=h.device),
                    torch.empty(F.mlp_tail_workspace_floats(M, c.units), dtype=torch.float32, device=h.device))
            self._grad_buf[key] = bufs
        dz1, ws = bufs
        F.mlp_tail(h, b.kernel.tensor, b.bias.tensor, c.kernel.tensor, c.bias.tensor, y, dz1, b.kernel.grad, b.bias.grad,
                   c.kernel.grad, c.bias.grad, ws, b.fused_keep, b.seed, self.store.global_step, b.prev_scale, 1.0 / M, stats)
        hook = self._after_layer_backward
        if hook is not None:
            hook(c, self)
            hook(b, self)
        g = dz1
        for layer in reversed(self.layers[:first]):
            g = layer.backward(g)
            if hook is not None:
                hook(layer, self)
            if g is None:
                break

    def fused_grad_plan(self, batch_size: int):
        """Arguments for the one-launch gradient kernel (``dtf_mlp3_fused_grad``) when this model is the reference
        network shape at a batch of at most 64 on the GPU and trains under a cluster strategy that takes the gradients
        after the backward pass (asynchronous workers, un-overlapped synchronous replicas); ``None`` otherwise.
        ``DTF_FUSED_GRAD=0`` keeps the layer-by-layer kernels."""
        if not self.compiled or self.device is None or self.device.type != "cuda" or not _FUSED_GRAD:
            return None
        if type(self.strategy) is LocalStrategy or self._after_layer_backward is not None:
            return None         # a single replica has the whole-step kernel (fused_step_plan)
        key = ("grad", int(batch_size))
        if key not in self._fused_plans:
            plan = None
            eff = self._effective
            if len(eff) == 3 and all(isinstance(l, Dense) and l.bias is not None and not l.softmax for l in eff) \
                    and [l.act for l in eff] == [ACT_RELU, ACT_RELU, ACT_SIGMOID] and eff[2].fused_keep >= 1.0 \
                    and not any(isinstance(l, Dropout) and not l.fused_into_previous and l.rate > 0.0 for l in self.layers) \
                    and isinstance(self.loss, losses_mod.MeanSquaredError):
                dims = (eff[0].kernel.shape[0], eff[0].units, eff[1].units, eff[2].units)
                if F.mlp3_fused_supported(dims, batch_size) and hasattr(nv_kernels(), "dtf_mlp3_fused_grad"):
                    plan = dict(dims=dims, offsets=[v.offset for l in eff for v in (l.kernel, l.bias)], l1=eff[0], l2=eff[1])
            self._fused_plans[key] = plan
        return self._fused_plans[key]

    def compute_gradients(self, x, y, stats: Optional[torch.Tensor] = None):
        """Forward + loss + backward: leaves dL/dvar in ``store.flat_grad`` (no variable update)."""
        fresh = getattr(self.strategy, "ensure_fresh", None)
        if fresh is not None:
            fresh(self)
        if x.is_cuda and x.dim() == 2 and y.dim() == 2 and y.dtype == torch.float32 and x.shape[0] <= 64:
            gp = self.fused_grad_plan(int(x.shape[0]))
            if gp is not None:
                st = self.store
                F.mlp3_fused_grad(x.contiguous(), y.contiguous(), st.flat, st.flat_grad, gp["offsets"], gp["dims"], True,
                                  self._extra_categorical, gp["l1"].fused_keep, gp["l2"].fused_keep, gp["l1"].seed,
                                  gp["l2"].seed, st.global_step, self.stats if stats is None else stats)
                return
        plan = self.tail_plan() if (x.is_cuda and y.dtype == torch.int64 and y.dim() == 1) else None
        if plan is not None:
            return self._compute_gradients_tail(x, y.contiguous(), self.stats if stats is None else stats, plan)
        out = self.forward(x, training=True)
        g = self._loss_and_metrics(out, y, True, self.stats if stats is None else stats)
        self.backward(g)

    def train_step(self, x: torch.Tensor, y: torch.Tensor):
        """One optimisation step on a device-resident batch; returns nothing (no host sync)."""
        direct = getattr(self.strategy, "train_step_direct", None)
        if direct is not None and direct(self, x, y):      # a parameter-server worker's step as one kernel
            self._host_step += 1
            return
        self.compute_gradients(x, y)
        self.strategy.apply_gradients(self)
        self._host_step += 1

    #: forward + loss + backward of this model can be captured in a hipGraph (static launch sequence)
    supports_gradient_graph = True

    def train_step_graphed(self, x: torch.Tensor, y: torch.Tensor):
        """``train_step`` for launch-bound batches under a strategy that exchanges gradients after the backward pass
        (asynchronous workers, un-overlapped synchronous replicas): the batch is copied into persistent buffers of its
        shape and forward + loss + backward are replayed from one hipGraph - three host calls instead of one launch per
        kernel; the strategy's exchange follows as usual.  Falls back to ``train_step`` where a graph cannot be used."""
        if not (self.supports_gradient_graph and x.is_cuda and self._after_layer_backward is None):
            return self.train_step(x, y)
        key = ("stage", tuple(x.shape), tuple(y.shape), y.dtype)
        bufs = self._graphs.get(key)
        if bufs is None:
            bufs = self._graphs[key] = (torch.empty_like(x, memory_format=torch.contiguous_format),
                                        torch.empty_like(y, memory_format=torch.contiguous_format))
        bufs[0].copy_(x)
        bufs[1].copy_(y)
        self.graphed_gradients(bufs[0], bufs[1]).replay()
        self.strategy.apply_gradients(self)
        self._host_step += 1

    # ------------------------------------------------------------------ statistics
    def read_stats(self, reset: bool = True, stats: Optional[torch.Tensor] = None) -> Dict[str, float]:
        """Mean loss / accuracy since the last reset (ONE device->host read)."""
        s = self.stats if stats is None else stats
        red = self.strategy.reduce_stats(s) if stats is None else s
        vals = red.tolist()
        n = max(vals[2], 1.0)
        logs = {"loss": vals[0] / n}
        if self.metrics:
            if self._extra_categorical:
                logs["acc"] = vals[3] / n
                if self._report_binary:
                    logs["binary_rounded_accuracy"] = vals[1] / n
            else:
                logs["acc"] = vals[1] / n
        if reset:
            s.zero_()
        return logs

    @property
    def global_step(self) -> int:
        return int(self.store.global_step.item())

    # ------------------------------------------------------------------ Keras entry points
    def train_on_batch(self, x, y) -> Dict[str, float]:
        self._check_compiled()
        xb, yb = _as_tensor(x, self.device), _as_tensor(y, self.device)
        self.stats.zero_()
        self.train_step(xb, yb)
        return self.read_stats()

    def test_on_batch(self, x, y) -> Dict[str, float]:
        self._check_compiled()
        xb, yb = _as_tensor(x, self.device), _as_tensor(y, self.device)
        st = F.new_stats(self.device)
        out = self.forward(xb, training=False)
        self._loss_and_metrics(out, yb, False, st)
        return self.read_stats(stats=st)

    def evaluate(self, x, y, batch_size: int = 32, verbose: int = 0) -> List[float]:
        self._check_compiled()
        xd, yd = _as_tensor(x, self.device), _as_tensor(y, self.device)
        st = F.new_stats(self.device)
        n = xd.shape[0]
        for i in range(0, n, batch_size):
            out = self.forward(xd[i:i + batch_size], training=False)
            self._loss_and_metrics(out, yd[i:i + batch_size], False, st)
        logs = self.read_stats(stats=self.strategy.reduce_stats(st))
        return [logs["loss"]] + ([logs["acc"]] if "acc" in logs else [])

    def predict(self, x, batch_size: int = 32) -> torch.Tensor:
        if not self.built:
            raise RuntimeError("model is not built")
        xd = _as_tensor(x, self.device)
        outs = []
        head = self._effective[-1] if self._effective else None
        for i in range(0, xd.shape[0], batch_size):
            out = self.forward(xd[i:i + batch_size], training=False)
            if isinstance(head, Dense) and head.softmax and head.emit_logits:
                out = F.softmax(out)
            outs.append(out.clone())
        return torch.cat(outs, dim=0)

    def _check_compiled(self):
        if not self.compiled:
            raise RuntimeError("call compile() first")

    # ------------------------------------------------------------------ graph-captured steps
    def _graphed_steps(self, x_buf: torch.Tensor, y_buf: torch.Tensor, batch_size: int, nsteps: int):
        """Capture ``nsteps`` consecutive training steps reading ``x_buf[k*B:(k+1)*B]`` into one hipGraph."""
        key = (x_buf.data_ptr(), y_buf.data_ptr(), batch_size, nsteps)
        g = self._graphs.get(key)
        if g is not None:
            return g
        # warm-up outside capture so every lazily allocated buffer exists
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        saved = self._snapshot()
        with torch.cuda.stream(side):
            self.train_step(x_buf[:batch_size], y_buf[:batch_size])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._restore(saved)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) keep working
            for k in range(nsteps):
                self.train_step(x_buf[k * batch_size:(k + 1) * batch_size],
                                y_buf[k * batch_size:(k + 1) * batch_size])
        self._host_step -= nsteps + 1  # capture and warm-up did not execute training steps
        self._graphs[key] = g
        return g

    def graphed_gradients(self, x_buf: torch.Tensor, y_buf: torch.Tensor):
        """hipGraph of forward + loss + backward on a fixed batch buffer (no variable update).  A replica whose
        gradient exchange runs outside the graph replays this, then calls ``strategy.apply_gradients`` - two host
        calls per step instead of one launch per kernel.  Not usable while a strategy overlaps collectives with
        backward through the per-layer hook."""
        if self._after_layer_backward is not None:
            raise RuntimeError("gradient graphs cannot contain the strategy's per-layer collectives")
        key = ("grad", x_buf.data_ptr(), y_buf.data_ptr(), tuple(x_buf.shape))
        g = self._graphs.get(key)
        if g is not None:
            return g
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        saved_stats = self.stats.clone()
        saved_other = {v.name: v.tensor.clone() for v in self.store.variables.values()
                       if not v.trainable and v.tensor is not None}       # moving averages: the forward pass writes them
        with torch.cuda.stream(side):
            self.compute_gradients(x_buf, y_buf)       # warm-up: every lazily allocated buffer exists afterwards
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.stats.copy_(saved_stats)
        for name, t in saved_other.items():
            self.store.variables[name].tensor.copy_(t)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) keep working
            self.compute_gradients(x_buf, y_buf)
        self._graphs[key] = g
        return g

    # ------------------------------------------------------------------ whole-step kernel
    def fused_step_plan(self, batch_size: int):
        """Arguments for the one-launch training step (``csrc/kernels/mlp_fused.hip``) when this model is the
        reference network shape (Dense relu [+Dropout], Dense relu [+Dropout], Dense sigmoid; MSE; Adam or plain
        SGD; single replica; batch <= 64) and runs on the GPU; ``None`` otherwise."""
        if not self.compiled or self.device is None or self.device.type != "cuda":
            return None
        if type(self.strategy) is not LocalStrategy or self._after_layer_backward is not None:
            return None
        key = (int(batch_size), id(self.optimizer))
        if key not in self._fused_plans:
            self._fused_plans[key] = self._make_fused_step_plan(batch_size)
        return self._fused_plans[key]

    def _make_fused_step_plan(self, batch_size: int):
        eff = self._effective
        if len(eff) != 3 or not all(isinstance(l, Dense) and l.bias is not None and not l.softmax for l in eff):
            return None
        if [l.act for l in eff] != [ACT_RELU, ACT_RELU, ACT_SIGMOID] or eff[2].fused_keep < 1.0:
            return None
        if any(isinstance(l, Dropout) and not l.fused_into_previous and l.rate > 0.0 for l in self.layers):
            return None
        if not isinstance(self.loss, losses_mod.MeanSquaredError):
            return None
        opt = self.optimizer
        if isinstance(opt, optim_mod.Adam):
            kind = 0
            m, v = self.store.slot("m"), self.store.slot("v")
        elif isinstance(opt, optim_mod.SGD) and opt.momentum == 0.0:
            kind = 1
            m = v = None
        else:
            return None
        dims = (eff[0].kernel.shape[0], eff[0].units, eff[1].units, eff[2].units)
        if not F.mlp3_fused_supported(dims, batch_size):
            return None
        offsets = [x.offset for l in eff for x in (l.kernel, l.bias)]
        return dict(dims=dims, offsets=offsets, opt=kind, m=m, v=v, l1=eff[0], l2=eff[1])

    def train_steps_fused(self, x_buf: torch.Tensor, y_buf: torch.Tensor, batch_size: int, nsteps: int, plan=None,
                          workers: Optional[int] = None):
        """``nsteps`` training steps on ``x_buf[k*B:(k+1)*B]`` in a single kernel launch.  ``workers`` (default:
        ``self.async_workers``, 1): that many asynchronous workers inside the GPU share the variables lock-free and
        take the steps between them (the reference's training mode, /root/reference/example.py:168-170)."""
        plan = plan or self.fused_step_plan(batch_size)
        if plan is None:
            raise RuntimeError("this model / batch size has no fused step kernel")
        st, opt, l1, l2 = self.store, self.optimizer, plan["l1"], plan["l2"]
        hp = (opt.lr, opt.beta1, opt.beta2, opt.epsilon) if plan["opt"] == 0 else (opt.lr, 0.0, 0.0, 0.0)
        # hyper-parameters, keep probabilities and dropout seeds are read at call time, like the layer-wise path does
        F.mlp3_fused_train(x_buf, y_buf, st.flat, plan["m"], plan["v"], plan["offsets"], plan["dims"], batch_size,
                           nsteps, True, plan["opt"], self._extra_categorical, l1.fused_keep, l2.fused_keep,
                           l1.seed, l2.seed, *hp, st.global_step, st.beta_powers, self.stats,
                           workers=max(1, min(int(workers if workers is not None else getattr(self, "async_workers", 1)),
                                              int(nsteps))))
        self._host_step += nsteps

    def _snapshot(self):
        st = self.store
        # everything a training step writes: variables, slots, counters, statistics - and the variables that are not
        # trained but updated by the forward pass (moving averages of batch normalisation)
        other = {v.name: v.tensor.clone() for v in st.variables.values() if not v.trainable and v.tensor is not None}
        return (st.flat.clone(), {k: v.clone() for k, v in st.slots.items()}, st.global_step.clone(),
                st.beta_powers.clone(), self.stats.clone(), other)

    def _restore(self, saved):
        st = self.store
        st.flat.copy_(saved[0])
        for k, v in saved[1].items():
            st.slots[k].copy_(v)
        st.global_step.copy_(saved[2])
        st.beta_powers.copy_(saved[3])
        self.stats.copy_(saved[4])
        for name, t in saved[5].items():
            st.variables[name].tensor.copy_(t)

    # ------------------------------------------------------------------ fit
    def fit(self, x, y, epochs: int = 1, batch_size: int = 32, validation_data=None, callbacks=None,
            shuffle: bool = True, verbose: int = 1, steps_per_graph: int = 0, initial_epoch: int = 0,
            fused: Optional[bool] = None, stream_from_host: bool = False, shuffle_seed: int = 0,
            async_workers: Optional[int] = None, validation_split: float = 0.0):
        """Keras ``fit``.  Data is moved to the device once; each epoch reshuffles on the device.

        ``validation_split`` (ignored when ``validation_data`` is given): that fraction of the examples, taken from
        the END of ``x`` / ``y`` before any shuffling, is held out and evaluated after every epoch (Keras' rule).

        ``stream_from_host=True`` keeps ``x`` / ``y`` in host memory instead (data sets larger than HBM): C++ worker
        threads gather each epoch's shuffled batches into pinned buffers and a copy stream moves them to the device
        ahead of the step that consumes them (``input_pipeline.HostPipeline``).

        ``steps_per_graph > 0`` (GPU only) replays that many consecutive steps from one
        captured hipGraph; 0 launches every kernel eagerly.  ``fused`` (default: automatic) selects the
        one-launch whole-step kernel when the model has one (``fused_step_plan``); ``False`` forces the
        layer-by-layer path.  ``async_workers=k`` (with that kernel): ``k`` asynchronous workers inside the GPU train
        on the shared variables lock-free and take an epoch's steps between them, as the reference's worker processes
        do against its parameter servers (/root/reference/example.py:168-170); ``global_step`` counts every step once.
        """
        if async_workers is not None:
            self.async_workers = max(1, int(async_workers))
        self._check_compiled()
        if validation_data is None and validation_split:
            if not 0.0 < validation_split < 1.0:
                raise ValueError("validation_split must be in (0, 1)")
            cut = int(len(x) * (1.0 - validation_split))
            if cut <= 0 or cut >= len(x):
                raise ValueError("validation_split=%r leaves no training or no validation examples" % validation_split)
            validation_data = (x[cut:], y[cut:])
            x, y = x[:cut], y[:cut]
        if stream_from_host:
            return self._fit_streamed(x, y, epochs, batch_size, validation_data, callbacks, shuffle, verbose,
                                      initial_epoch, shuffle_seed)
        xd, yd = _as_tensor(x, self.device), _as_tensor(y, self.device)
        n = xd.shape[0]
        val = None
        if validation_data is not None:
            val = (_as_tensor(validation_data[0], self.device), _as_tensor(validation_data[1], self.device))
        steps = int(math.ceil(n / float(batch_size)))
        params = {"epochs": epochs, "batch_size": batch_size, "steps": steps, "samples": n,
                  "metrics": ["loss"] + (["acc"] if self.metrics else [])}
        history = cb.History()
        cbs = list(callbacks or []) + ([cb.ProgbarLogger()] if verbose else []) + [history]   # History last, as in Keras: it sees what the others add to logs
        cl = cb.CallbackList(cbs, self, params)
        self.stop_training = False
        cl.call("on_train_begin", None)
        plan = self.fused_step_plan(batch_size) if fused in (None, True) else None
        if fused and plan is None:
            raise RuntimeError("fused=True but this model / batch size has no fused step kernel")
        use_graph = steps_per_graph > 0 and self.device.type == "cuda" and plan is None
        x_ep = torch.empty_like(xd) if (shuffle or use_graph) else xd
        y_ep = torch.empty_like(yd) if (shuffle or use_graph) else yd
        full_steps = n // batch_size
        # the epoch permutations come from their own generator: a run is reproducible from (shuffle_seed, initial_epoch)
        # whatever else has drawn from the process-wide generators
        shuffle_gen = torch.Generator(device=self.device).manual_seed(int(shuffle_seed) * 1000003 + int(initial_epoch)) \
            if shuffle else None
        for epoch in range(initial_epoch, epochs):
            if self.stop_training:
                break
            cl.call("on_epoch_begin", epoch, None)
            if shuffle:
                perm = torch.randperm(n, device=self.device, generator=shuffle_gen)
                torch.index_select(xd, 0, perm, out=x_ep)
                torch.index_select(yd, 0, perm, out=y_ep)
            elif use_graph and x_ep is not xd:
                x_ep.copy_(xd)
                y_ep.copy_(yd)
            self.stats.zero_()
            done = 0
            if plan is not None:
                chunk = full_steps if not cl.wants_batch_logs else max(steps_per_graph, 1)
                while done < full_steps:
                    k = min(chunk, full_steps - done)
                    off = done * batch_size
                    self.train_steps_fused(x_ep[off:], y_ep[off:], batch_size, k, plan)
                    done += k
                    if cl.wants_batch_logs:
                        cl.call("on_batch_end", done - 1, {})
                if done < steps:        # ragged last batch
                    off = done * batch_size
                    self.train_steps_fused(x_ep[off:], y_ep[off:], n - off, 1)
                    done += 1
                    if cl.wants_batch_logs:
                        cl.call("on_batch_end", done - 1, {})
            elif use_graph:
                while full_steps - done >= steps_per_graph:
                    off = done * batch_size
                    span = steps_per_graph * batch_size
                    g = self._graphed_steps(x_ep[off:off + span], y_ep[off:off + span], batch_size, steps_per_graph)
                    g.replay()
                    self._host_step += steps_per_graph
                    done += steps_per_graph
                    if cl.wants_batch_logs:
                        cl.call("on_batch_end", done - 1, {})
            for s in range(done, steps):
                xb = x_ep[s * batch_size:(s + 1) * batch_size]
                yb = y_ep[s * batch_size:(s + 1) * batch_size]
                self.train_step(xb, yb)
                if cl.wants_batch_logs:
                    cl.call("on_batch_end", s, {})
            logs = self.read_stats()
            if val is not None:
                ev = self.evaluate(val[0], val[1], batch_size=max(batch_size, 1))
                logs["val_loss"] = ev[0]
                if len(ev) > 1:
                    logs["val_acc"] = ev[1]
            cl.call("on_epoch_end", epoch, logs)
        cl.call("on_train_end", None)
        return history

    def _fit_streamed(self, x, y, epochs, batch_size, validation_data, callbacks, shuffle, verbose, initial_epoch,
                      shuffle_seed):
        from ..input_pipeline import HostPipeline
        pipe = HostPipeline(x, y, batch_size=batch_size, shuffle=shuffle, seed=shuffle_seed, device=self.device)
        try:
            n = pipe.rows
            steps = int(math.ceil(n / float(batch_size)))
            val = None
            if validation_data is not None:
                val = (_as_tensor(validation_data[0], self.device), _as_tensor(validation_data[1], self.device))
            params = {"epochs": epochs, "batch_size": batch_size, "steps": steps, "samples": n,
                      "metrics": ["loss"] + (["acc"] if self.metrics else [])}
            history = cb.History()
            cbs = list(callbacks or []) + ([cb.ProgbarLogger()] if verbose else []) + [history]   # History last, as in Keras: it sees what the others add to logs
            cl = cb.CallbackList(cbs, self, params)
            self.stop_training = False
            cl.call("on_train_begin", None)
            for epoch in range(initial_epoch, epochs):
                if self.stop_training:
                    break
                cl.call("on_epoch_begin", epoch, None)
                self.stats.zero_()
                for s, (xb, yb) in enumerate(pipe.batches(epoch)):
                    self.train_step(xb, yb)
                    if cl.wants_batch_logs:
                        cl.call("on_batch_end", s, {})
                logs = self.read_stats()
                if val is not None:
                    ev = self.evaluate(val[0], val[1], batch_size=max(batch_size, 1))
                    logs["val_loss"] = ev[0]
                    if len(ev) > 1:
                        logs["val_acc"] = ev[1]
                cl.call("on_epoch_end", epoch, logs)
            cl.call("on_train_end", None)
            return history
        finally:
            pipe.close()

    # ------------------------------------------------------------------ learning rate
    def set_learning_rate(self, lr: float):
        """Change the optimizer's learning rate.  Captured hipGraphs hold the old value as a kernel argument, so they
        are dropped and re-captured on their next use."""
        self._check_compiled()
        if float(lr) != self.optimizer.lr:
            self.optimizer.lr = float(lr)
            self._graphs.clear()

    # ------------------------------------------------------------------ weights
    def get_weights(self):
        """The variables as NumPy arrays, in creation order (Keras ``model.get_weights()``)."""
        self._check_built()
        return [v.tensor.detach().cpu().numpy().copy() for v in self.store.variables.values()]

    def set_weights(self, weights):
        self._check_built()
        variables = list(self.store.variables.values())
        if len(weights) != len(variables):
            raise ValueError("set_weights got %d arrays for %d variables" % (len(weights), len(variables)))
        for v, w in zip(variables, weights):
            w = torch.as_tensor(w, dtype=torch.float32)
            if tuple(w.shape) != tuple(v.tensor.shape):
                raise ValueError("shape mismatch for %s: %s vs %s" % (v.name, tuple(w.shape), tuple(v.tensor.shape)))
            v.tensor.copy_(w)

    def save_weights(self, filepath: str, include_optimizer: bool = True) -> str:
        """Write the variables (and, by default, ``global_step`` and the optimizer slots) as a TensorFlow-readable
        TensorBundle checkpoint ``<filepath>.index`` / ``<filepath>.data-00000-of-00001``; returns the prefix."""
        from ..saver import Saver
        self._check_built()
        opt = self.optimizer if (include_optimizer and getattr(self, "optimizer", None) is not None) else None
        if opt is not None:
            opt.create_slots(self.store)
        return Saver(self.store, optimizer=opt, max_to_keep=0).save(filepath, write_state=False)

    def load_weights(self, filepath: str, strict: bool = False) -> int:
        """Restore what ``filepath`` holds (variables; step and optimizer slots when present); returns global_step.
        ``strict=True`` requires every persistent tensor of this model to be in the checkpoint."""
        from ..saver import Saver
        self._check_built()
        opt = getattr(self, "optimizer", None)
        if opt is not None:
            opt.create_slots(self.store)          # so that slots found in the checkpoint have somewhere to go
        step = Saver(self.store, optimizer=opt).restore(filepath, strict=strict)
        self._host_step = step
        return step

    def _check_built(self):
        if not self.built:
            raise RuntimeError("build() or compile() the model first")

    # ------------------------------------------------------------------ misc
    def summary(self) -> str:
        lines = ["Model: %s" % self.name, "%-28s %-20s %10s" % ("Layer (type)", "Output Shape", "Param #")]
        total = 0
        for l in self.layers:
            n = sum(v.numel for v in l.variables)
            total += n
            lines.append("%-28s %-20s %10d" % ("%s (%s)" % (l.name, type(l).__name__), str((None,) + tuple(l.output_shape or ())), n))
        lines.append("Total params: %d" % total)
        return "\n".join(lines)

    def get_config(self):
        """Architecture as plain data; the first layer carries the per-example ``input_shape`` once it is known."""
        layers = [dict(l.get_config()) for l in self.layers]
        if layers and self.layers[0].input_shape is not None:
            layers[0]["input_shape"] = list(self.layers[0].input_shape)
        return {"name": self.name, "layers": layers}

    @classmethod
    def from_config(cls, config: dict) -> "Sequential":
        from ..layers import layer_from_config
        m = cls(name=config.get("name", "sequential"))
        for lc in config["layers"]:
            m.add(layer_from_config(lc))
        return m

    def to_json(self) -> str:
        import json
        return json.dumps(self.get_config(), indent=1)

    def save(self, filepath: str, include_optimizer: bool = True) -> str:
        """Keras ``model.save``: ``<filepath>.json`` (architecture) next to a TensorBundle checkpoint with the
        variables and, by default, the step and optimizer slots.  ``load_model(filepath)`` rebuilds it."""
        import os
        os.makedirs(os.path.dirname(os.path.abspath(filepath)), exist_ok=True)
        with open(filepath + ".json", "w") as f:
            f.write(self.to_json())
        return self.save_weights(filepath, include_optimizer=include_optimizer)


def model_from_json(text: str) -> Sequential:
    import json
    return Sequential.from_config(json.loads(text))


def load_model(filepath: str, compile_args: Optional[dict] = None) -> Sequential:
    """Rebuild a model written by ``Sequential.save``.  With ``compile_args`` (keyword arguments of ``compile``) the
    model is compiled first, so optimizer slots and ``global_step`` in the file are restored too; without them the
    model is only built (on ``compile_args``-less calls the default device) and its variables restored."""
    with open(filepath + ".json") as f:
        m = model_from_json(f.read())
    if compile_args is not None:
        m.compile(**compile_args)
    else:
        m.build()
    m.load_weights(filepath)
    return m

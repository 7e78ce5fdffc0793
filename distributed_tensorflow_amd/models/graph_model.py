"""A model that is a directed acyclic graph of layers and tensor ops, differentiated by reverse-mode autodiff.

``tf.gradients`` behind ``optimizer.minimize`` differentiates whatever graph the script has built
(/root/reference/example.py:146-170).  ``Sequential`` covers chains with hand-scheduled backward passes; this class
takes the symbolic graph of :mod:`distributed_tensorflow_amd.compat` as it is - several placeholders, skip
connections, several losses added up - evaluates it with the autograd wrappers of :mod:`..ops.autograd` (the same
fused HIP kernels) and lets PyTorch's tape produce the gradients.  The variables still live in the flat
``VariableStore`` and the gradients land in its flat gradient buffer, so optimizers, parallel strategies, checkpoints
and sessions work on it exactly as on a ``Sequential``.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import torch

from .. import layers as L
from ..ops import autograd as A
from ..ops import functional as F
from .sequential import Sequential, _as_tensor


def _is_node(x) -> bool:
    return hasattr(x, "op") and hasattr(x, "inputs")


class GraphModel(Sequential):
    supports_gradient_graph = False      # the backward pass is PyTorch's autograd engine, not a fixed launch list

    def __init__(self, loss_node, inputs: Sequence[Any], labels: Sequence[Any], accuracy_node=None, output_node=None,
                 name: str = "graph_model"):
        super().__init__(name=name)
        self.loss_node = loss_node
        self.input_nodes = list(inputs)
        self.label_nodes = list(labels)
        self.accuracy_node = accuracy_node
        self.output_node = output_node
        self.order = self._topological(loss_node)
        self.layers = []
        seen = set()
        for n in self.order:
            if n.op == "layer" and id(n.attrs["layer"]) not in seen:
                seen.add(id(n.attrs["layer"]))
                self.layers.append(n.attrs["layer"])
        self._var_nodes = [n for n in self.order if n.op == "Variable" and n.attrs.get("trainable", True) and n.shape]
        self._leaves: Dict[str, torch.Tensor] = {}
        self._last_env: Optional[dict] = None

    # ------------------------------------------------------------------ structure
    @staticmethod
    def _topological(root) -> List[Any]:
        order, done = [], set()
        stack = [(root, False)]
        while stack:
            node, expanded = stack.pop()
            if not _is_node(node) or id(node) in done:
                continue
            if expanded:
                done.add(id(node))
                order.append(node)
                continue
            stack.append((node, True))
            for i in reversed(node.inputs):
                if _is_node(i) and id(i) not in done:
                    stack.append((i, False))
        return order

    def _infer_shapes(self) -> Dict[int, tuple]:
        """Per-example shapes of every node that needs one (layers are built from them)."""
        shapes: Dict[int, tuple] = {}
        for n in self.order:
            ins = [shapes.get(id(i)) if _is_node(i) else None for i in n.inputs]
            if n.op == "Placeholder":
                shapes[id(n)] = tuple(int(s) for s in (n.shape or ())[1:])
            elif n.op == "Variable":
                shapes[id(n)] = tuple(n.shape or ())
            elif n.op == "layer":
                layer = n.attrs["layer"]
                if ins[0] is None:
                    raise ValueError("layer %s has an input of unknown shape" % n.name)
                if not getattr(layer, "_graph_built", False):
                    layer._graph_out_shape = tuple(layer.build(ins[0], self.store))
                    layer._graph_built = True
                shapes[id(n)] = layer._graph_out_shape
            elif n.op == "matmul":
                w = n.inputs[1]
                shapes[id(n)] = tuple(ins[0][:-1]) + (int(w.shape[1]),)
            elif n.op in ("add", "sub", "mul", "div", "bias_add"):
                shapes[id(n)] = ins[0] if ins[0] is not None and (not _is_node(n.inputs[0]) or n.inputs[0].op != "Variable") \
                    else (ins[1] if len(ins) > 1 else None)
            elif n.op in ("relu", "sigmoid", "tanh", "softmax", "nn_dropout", "round", "square", "cast"):
                shapes[id(n)] = ins[0]
            elif n.op in ("conv2d", "max_pool") and ins[0] is not None and len(ins[0]) == 3:
                from ..ops.functional import conv_geometry
                h, w, c = ins[0]
                kh, kw = (n.inputs[1].shape[:2] if n.op == "conv2d" else n.attrs["ksize"])
                oh, ow = conv_geometry(h, w, int(kh), int(kw), n.attrs["strides"], n.attrs["padding"])[:2]
                shapes[id(n)] = (oh, ow, int(n.inputs[1].shape[3]) if n.op == "conv2d" else c)
            elif n.op == "reshape":
                tail = [int(s) for s in n.attrs["shape"][1:]]
                if -1 in tail and ins[0] is not None:
                    known = 1
                    for s in tail:
                        known *= s if s != -1 else 1
                    total = 1
                    for s in ins[0]:
                        total *= s
                    tail[tail.index(-1)] = total // max(known, 1)
                shapes[id(n)] = tuple(tail) if -1 not in tail else None
            else:
                shapes[id(n)] = None
        return shapes

    def build(self, input_shape=None, device=None, seed: Optional[int] = None, pad_multiple: int = 64):
        if self.built:
            return self
        from ..models.sequential import default_device
        # hand-written variables first (creation order of the script), then the layers' own
        for n in self._var_nodes:
            if n.name not in self.store.variables:
                self.store.create(n.name, n.shape, n.attrs["initial_value"])
        self._infer_shapes()
        if seed is not None:
            for idx, layer in enumerate(self.layers):
                if hasattr(layer, "seed"):
                    layer.seed = (int(seed) * 1000003 + idx * 7919 + 12345) & 0x7FFFFFFF
        self._effective = list(self.layers)
        self.device = torch.device(device) if device is not None else default_device()
        self.store.finalize(self.device, seed=seed, pad_multiple=pad_multiple)
        self.stats = F.new_stats(self.device)
        for v in self.store.trainable:
            leaf = v.tensor.detach().requires_grad_(True)      # shares the flat buffer's memory
            leaf.grad = v.grad                                  # gradients accumulate straight into the flat buffer
            self._leaves[v.name] = leaf
        self.input_shape = tuple(int(s) for s in (self.input_nodes[0].shape or ())[1:]) if self.input_nodes else ()
        self.output_shape = None
        self.built = True
        return self

    def _plan_fusion(self):
        self._effective = list(self.layers)

    def tail_plan(self):
        return None

    def fused_step_plan(self, batch_size: int):
        return None

    def _graphed_steps(self, *a, **k):
        raise RuntimeError("graph models are not captured into hipGraphs (autograd allocates during backward)")

    # ------------------------------------------------------------------ evaluation
    def _leaf(self, var) -> torch.Tensor:
        return self._leaves[var.name]

    def _layer(self, layer, x, training: bool):
        if isinstance(layer, L.Dense):
            b = self._leaf(layer.bias) if layer.bias is not None else None
            y = A.dense(x, self._leaf(layer.kernel), b, layer.act)
            return torch.softmax(y, dim=-1) if layer.softmax else y
        if isinstance(layer, L.Dropout):
            return A.dropout(x, layer.rate, layer.seed, self.store.global_step, training)
        if isinstance(layer, L.Activation):
            if getattr(layer, "activation", None) == "softmax":
                return torch.softmax(x, dim=-1)
            return A.activation(x, layer.activation)
        if isinstance(layer, L.Flatten):
            return x.reshape(x.shape[0], -1)
        if isinstance(layer, L.Conv2D):
            b = self._leaf(layer.bias) if layer.bias is not None else None
            return A.conv2d(x, self._leaf(layer.kernel), b, layer.strides, layer.padding,
                            {0: None, 1: "relu", 2: "sigmoid", 3: "tanh"}[layer.act])
        if isinstance(layer, L.MaxPooling2D):
            return A.max_pool2d(x, layer.pool_size, layer.strides, layer.padding)
        if isinstance(layer, L.AveragePooling2D):
            return A.avg_pool2d(x, layer.pool_size, layer.strides, layer.padding)
        if isinstance(layer, L.GlobalAveragePooling2D):
            return A.avg_pool2d(x, tuple(x.shape[1:3]), tuple(x.shape[1:3]), "valid").reshape(x.shape[0], x.shape[3])
        if isinstance(layer, L.BatchNormalization):
            return A.batch_norm(x, self._leaf(layer.gamma) if layer.gamma is not None else None,
                                self._leaf(layer.beta) if layer.beta is not None else None, layer.moving_mean.tensor,
                                layer.moving_variance.tensor, layer.momentum, layer.epsilon, training)
        raise NotImplementedError("layer %s is not supported in a graph model" % type(layer).__name__)

    def _eval(self, node, env: dict, training: bool):
        if not _is_node(node):
            return node
        key = id(node)
        if key in env:
            return env[key]
        op = node.op
        if op == "Placeholder":
            raise ValueError("placeholder %s must be fed" % node.name)
        if op == "Variable":
            if node.name in self._leaves:
                out = self._leaves[node.name]
            else:
                raise KeyError("variable %s has no value in this model" % node.name)
        elif op == "reduce_mean" and node.attrs.get("axis") is None and _is_node(node.inputs[0]) \
                and node.inputs[0].op in ("keras_mse", "keras_bce", "softmax_xent", "sparse_softmax_xent"):
            # the fused loss kernels: value and gradient from one launch
            inner = node.inputs[0]
            y = self._eval(inner.inputs[0], env, training)
            p = self._eval(inner.inputs[1], env, training)
            if inner.op == "keras_mse":
                out = A.mean_squared_error(p, y.to(torch.float32))
            elif inner.op == "keras_bce":
                out = A.binary_crossentropy(p, y.to(torch.float32))
            else:
                out = A.softmax_cross_entropy(p, y if inner.op == "sparse_softmax_xent" and y.dtype == torch.int64
                                              else (y.long() if inner.op == "sparse_softmax_xent" else y))
        else:
            ins = [self._eval(i, env, training) for i in node.inputs]
            if op == "layer":
                out = self._layer(node.attrs["layer"], ins[0], training)
            elif op == "matmul":
                out = A.dense(ins[0], ins[1], None, None)
            elif op in ("add", "bias_add"):
                out = ins[0] + ins[1]
            elif op == "sub":
                out = ins[0] - ins[1]
            elif op == "mul":
                out = ins[0] * ins[1]
            elif op == "div":
                out = ins[0] / ins[1]
            elif op == "relu":
                out = A.activation(ins[0], "relu")
            elif op == "sigmoid":
                out = A.activation(ins[0], "sigmoid")
            elif op == "tanh":
                out = A.activation(ins[0], "tanh")
            elif op == "conv2d":
                out = A.conv2d(ins[0], ins[1], None, node.attrs["strides"], node.attrs["padding"], None)
            elif op == "max_pool":
                out = A.max_pool2d(ins[0], node.attrs["ksize"], node.attrs["strides"], node.attrs["padding"])
            elif op == "reshape":
                out = ins[0].reshape(node.attrs["shape"])
            elif op == "softmax":
                out = torch.softmax(ins[0], dim=-1)
            elif op == "nn_dropout":
                seed = node.attrs.setdefault("_seed", (hash(node.name) & 0x7FFFFFF) + 1)
                if node.attrs["keep_prob"] is None:      # a fed keep probability: applied as fed (one host read per run)
                    keep = float(ins[1])
                    out = A.dropout(ins[0], 1.0 - keep, seed, self.store.global_step, keep < 1.0)
                else:
                    out = A.dropout(ins[0], 1.0 - node.attrs["keep_prob"], seed, self.store.global_step, training)
            elif op == "square":
                out = ins[0] * ins[0]
            elif op == "round":
                out = torch.round(ins[0])
            elif op == "equal":
                out = ins[0] == ins[1]
            elif op == "cast":
                out = ins[0].to(node.attrs["dtype"])
            elif op == "argmax":
                out = ins[0].argmax(dim=node.attrs["axis"] if node.attrs["axis"] is not None else 0)
            elif op == "reduce_mean":
                ax = node.attrs.get("axis")
                out = ins[0].float().mean() if ax is None else ins[0].float().mean(dim=ax)
            elif op == "reduce_sum":
                ax = node.attrs.get("axis")
                out = ins[0].sum() if ax is None else ins[0].sum(dim=ax)
            elif op == "keras_mse":
                out = ((ins[1] - ins[0]) ** 2).mean(dim=-1)
            elif op == "keras_bce":
                pc = ins[1].clamp(F.BCE_EPS, 1.0 - F.BCE_EPS)
                out = (-(ins[0] * pc.log() + (1.0 - ins[0]) * (1.0 - pc).log())).mean(dim=-1)
            elif op == "softmax_xent":
                out = -(ins[0] * torch.log_softmax(ins[1], dim=1)).sum(dim=1)
            elif op == "sparse_softmax_xent":
                out = torch.nn.functional.cross_entropy(ins[1], ins[0].long(), reduction="none")
            else:
                raise NotImplementedError("op %r (node %s) cannot be differentiated by the graph model" % (op, node.name))
        env[key] = out
        return out

    def _bind(self, nodes, values) -> dict:
        vals = list(values) if isinstance(values, (tuple, list)) else [values]
        if len(vals) != len(nodes):
            raise ValueError("expected %d fed tensors (%s), got %d" % (len(nodes), ", ".join(n.name for n in nodes), len(vals)))
        return {id(n): _as_tensor(v, self.device) for n, v in zip(nodes, vals)}

    # ------------------------------------------------------------------ Sequential protocol
    def forward(self, x, training: bool = False):
        """Feeds the input placeholders (one tensor, or a tuple in placeholder order); returns the designated output."""
        env = self._bind(self.input_nodes, x)
        env["training"] = training
        self._last_env = env
        if self.output_node is None:
            return None
        with torch.no_grad():
            return self._eval(self.output_node, env, training)

    def _loss_and_metrics(self, out, y, want_grad: bool, stats: torch.Tensor):
        """Loss (and accuracy) of the graph for the inputs of the last ``forward`` and the labels ``y``, into ``stats``."""
        env = self._last_env
        env.update(self._bind(self.label_nodes, y))
        training = bool(env.get("training", False))
        with torch.no_grad():
            loss = self._eval(self.loss_node, env, training)
            self._account(env, loss, stats, training)
        return None

    def _account(self, env, loss, stats, training):
        M = float(next(iter(v for k, v in env.items() if isinstance(v, torch.Tensor))).shape[0])
        stats[0] += loss.detach() * M
        stats[2] += M
        if self.accuracy_node is not None:
            with torch.no_grad():
                stats[1] += self._eval(self.accuracy_node, env, training).float() * M

    def compute_gradients(self, x, y, stats: Optional[torch.Tensor] = None):
        """Forward + loss + reverse-mode backward: leaves dL/dvar in ``store.flat_grad``."""
        stats = self.stats if stats is None else stats
        env = self._bind(self.input_nodes, x)
        env.update(self._bind(self.label_nodes, y))
        self.store.flat_grad.zero_()
        loss = self._eval(self.loss_node, env, True)
        loss.backward()
        self._account(env, loss, stats, True)

    def backward(self, dy):
        raise RuntimeError("a graph model differentiates through compute_gradients()")

    def _slices(self, t, i, j):
        return tuple(a[i:j] for a in t) if isinstance(t, (tuple, list)) else t[i:j]

    def evaluate(self, x, y, batch_size: int = 32, verbose: int = 0):
        self._check_compiled()
        st = F.new_stats(self.device)
        n = (x[0] if isinstance(x, (tuple, list)) else x).shape[0]
        for i in range(0, n, batch_size):
            self.forward(self._slices(x, i, i + batch_size), training=False)
            self._loss_and_metrics(None, self._slices(y, i, i + batch_size), False, st)
        logs = self.read_stats(stats=self.strategy.reduce_stats(st))
        return [logs["loss"]] + ([logs["acc"]] if "acc" in logs else [])

    def test_on_batch(self, x, y):
        self._check_compiled()
        st = F.new_stats(self.device)
        self.forward(x, training=False)
        self._loss_and_metrics(None, y, False, st)
        return self.read_stats(stats=st)

    def train_on_batch(self, x, y):
        self._check_compiled()
        self.stats.zero_()
        self.train_step(x, y)
        return self.read_stats()

    def predict(self, x, batch_size: int = 32):
        if self.output_node is None:
            raise RuntimeError("this graph model has no designated output")
        n = (x[0] if isinstance(x, (tuple, list)) else x).shape[0]
        return torch.cat([self.forward(self._slices(x, i, i + batch_size), training=False).clone()
                          for i in range(0, n, batch_size)], dim=0)

    def get_config(self):
        return {"name": self.name, "class": "GraphModel",
                "nodes": [{"name": n.name, "op": n.op, "inputs": [i.name for i in n.inputs if _is_node(i)]} for n in self.order]}

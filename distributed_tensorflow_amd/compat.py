"""Graph-style (TensorFlow-1.x-shaped) front end.

Lets a script written like /root/reference/example.py run almost unchanged::

    from distributed_tensorflow_amd import compat as tf
    from distributed_tensorflow_amd.compat import keras

    with tf.device(device):
        input_ = tf.placeholder(tf.float32, shape=(None, 64))
        x = keras.layers.Dense(128, activation='relu')(input_)
        ...
        loss = tf.reduce_mean(keras.losses.mean_squared_error(labels, preds))
        train_step = tf.train.AdamOptimizer().minimize(loss, global_step=global_step)
    with tf.train.MonitoredTrainingSession(master=target, is_chief=..., checkpoint_dir=..., hooks=hooks) as sess:
        sess.run([accuracy, loss, summ, train_step], feed_dict={input_: bx, labels: by, K.learning_phase(): 1})

There is no tracing compiler behind this: symbolic nodes only *describe* the model.  On the
first session the graph is lowered once onto the kernel-launch engine.  The fast path: a chain
of layer nodes on one placeholder becomes a :class:`~distributed_tensorflow_amd.models.Sequential`,
the loss expression is matched to a fused loss kernel, the accuracy expression to the metric the
same kernel produces - and ``sess.run`` of the training op is the same fixed sequence of HIP
kernel launches as ``model.train_step``.  Any other graph - several placeholders, skip
connections, several losses added up - becomes a
:class:`~distributed_tensorflow_amd.models.graph_model.GraphModel`: the same kernels behind
``torch.autograd.Function`` wrappers, differentiated in reverse mode like ``tf.gradients`` does
behind ``minimize`` (/root/reference/example.py:170).  Other expressions (``tf.round``,
``tf.equal``...) are evaluated eagerly for fetches that are not on the training path.
"""
from __future__ import annotations

import contextlib
import types
from typing import Any, Dict, List, Optional, Sequence

import torch

from . import callbacks as _callbacks
from . import cluster as _cluster
from . import flags as _flags
from . import layers as _layers
from . import losses as _losses
from . import optimizers as _optimizers
from . import saver as _saver
from . import session as _session
from . import summary as _summary
from .engine import ACCURACY, GLOBAL_STEP, LOSS, TRAIN_OP, ModelEngine
from .models.sequential import Sequential as _Sequential
from .models.sequential import _as_tensor

float32 = torch.float32
float64 = torch.float64
int32 = torch.int32
int64 = torch.int64
bool_ = torch.bool
_DTYPES = {"float": torch.float32, "float32": torch.float32, "int32": torch.int32, "int64": torch.int64,
           "bool": torch.bool}


# --------------------------------------------------------------------------- symbolic nodes
class Tensor:
    """A symbolic value.  ``op`` names how it is computed from ``inputs``."""

    def __init__(self, op: str, inputs: Sequence[Any] = (), attrs: Optional[dict] = None, name: Optional[str] = None,
                 shape=None, graph: Optional["Graph"] = None):
        self.op = op
        self.inputs = list(inputs)
        self.attrs = dict(attrs or {})
        self.graph = graph or get_default_graph()
        self.name = self.graph.unique_name(name or op)
        self.shape = shape
        self.device = self.graph.current_device(op)
        self.graph.nodes.append(self)

    def eval(self, feed_dict=None, session=None):
        """``Tensor.eval`` (/root/reference/example.py:225): sugar for ``session.run(self, feed_dict)``."""
        sess = session or _default_session()
        if sess is None:
            raise ValueError("no session: pass session= or enter one with `with sess:`")
        return sess.run(self, feed_dict)

    def __repr__(self):
        return "<Tensor %r op=%s>" % (self.name, self.op)

    __hash__ = object.__hash__

    # arithmetic sugar
    def __sub__(self, o): return Tensor("sub", [self, o])
    def __add__(self, o): return Tensor("add", [self, o])
    def __mul__(self, o): return Tensor("mul", [self, o])
    def __truediv__(self, o): return Tensor("div", [self, o])


class Operation(Tensor):
    """A node run for its effect (training op, initializer); fetching it yields ``None``."""


class InitialValue:
    """What ``tf.random_normal(shape, ...)`` / ``tf.zeros(shape)`` ... return: a shape plus the recipe that fills it
    (drawn from the model's seeded generator when the graph is lowered).  Only meaningful as the initial value of a
    ``tf.Variable``."""

    def __init__(self, shape, kind: str, **params):
        self.shape = tuple(int(s) for s in shape)
        self.kind = kind
        self.params = params

    def __call__(self, shape, generator=None) -> torch.Tensor:
        shape = tuple(shape)
        p = self.params
        if self.kind == "zeros":
            return torch.zeros(shape)
        if self.kind == "ones":
            return torch.ones(shape)
        if self.kind == "constant":
            return torch.full(shape, float(p["value"])) if not isinstance(p["value"], torch.Tensor) \
                else p["value"].to(torch.float32).reshape(shape).clone()
        if self.kind == "random_normal":
            return torch.randn(shape, generator=generator) * p["stddev"] + p["mean"]
        if self.kind == "truncated_normal":       # values beyond two standard deviations are re-drawn
            t = torch.randn(shape, generator=generator)
            for _ in range(8):
                bad = t.abs() > 2.0
                if not bad.any():
                    break
                t = torch.where(bad, torch.randn(shape, generator=generator), t)
            return t.clamp_(-2.0, 2.0) * p["stddev"] + p["mean"]
        if self.kind == "random_uniform":
            return torch.rand(shape, generator=generator) * (p["maxval"] - p["minval"]) + p["minval"]
        if self.kind == "glorot_uniform":
            from .variables import glorot_uniform
            return glorot_uniform(shape, generator)
        raise ValueError("unknown initial value kind %r" % self.kind)


def zeros(shape, dtype=None, name=None): return InitialValue(shape, "zeros")
def ones(shape, dtype=None, name=None): return InitialValue(shape, "ones")
def random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    return InitialValue(shape, "random_normal", mean=float(mean), stddev=float(stddev))
def truncated_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    return InitialValue(shape, "truncated_normal", mean=float(mean), stddev=float(stddev))
def random_uniform(shape, minval=0.0, maxval=1.0, dtype=None, seed=None, name=None):
    return InitialValue(shape, "random_uniform", minval=float(minval), maxval=float(maxval))
def glorot_uniform_initializer(seed=None, dtype=None):
    return lambda shape, dtype=None, partition_info=None: InitialValue(shape, "glorot_uniform")


def constant(value, dtype=None, shape=None, name=None):
    t = torch.as_tensor(value, dtype=torch.float32)
    return InitialValue(tuple(shape) if shape is not None else tuple(t.shape), "constant",
                        value=float(t) if t.dim() == 0 else t)


class VariableNode(Tensor):
    """``tf.Variable``.  Non-trainable scalars (``global_step``) are counters of the engine; a trainable variable
    with an ``InitialValue`` (or array) becomes a weight or bias of the layer that ``tf.matmul`` / ``+`` wire it into."""

    def __init__(self, initial_value, dtype=None, name=None, trainable=True):
        if not isinstance(initial_value, InitialValue) and trainable and not isinstance(initial_value, (int, float)):
            initial_value = constant(initial_value)
        shape = initial_value.shape if isinstance(initial_value, InitialValue) else ()
        super().__init__("Variable", [], {"initial_value": initial_value, "dtype": dtype, "trainable": trainable},
                         name=name or "Variable", shape=shape)


class Graph:
    def __init__(self):
        self.nodes: List[Tensor] = []
        self._names: Dict[str, int] = {}
        self._device_stack: List[Any] = []
        self._scope: List[str] = []
        self._engine: Optional["GraphEngine"] = None
        self.finalized = False
        self.learning_phase: Optional[Tensor] = None
        self.device_setter = None

    def unique_name(self, base: str) -> str:
        full = "/".join(self._scope + [base])
        n = self._names.get(full, 0)
        self._names[full] = n + 1
        return full if n == 0 else "%s_%d" % (full, n)

    def current_device(self, op: str) -> Optional[str]:
        if not self._device_stack:
            return None
        d = self._device_stack[-1]
        if isinstance(d, _cluster.DeviceSetter):
            return d("Variable" if op == "Variable" else "op")
        return d if isinstance(d, str) else None

    def as_graph_def(self) -> dict:
        return {"nodes": [{"name": n.name, "op": n.op, "device": n.device,
                           "inputs": [i.name for i in n.inputs if isinstance(i, Tensor)]} for n in self.nodes]}

    def get_config(self):
        return self.as_graph_def()

    def graph_def_nodes(self):
        """GraphDef nodes: those of the lowered model (variables on the devices the device setter chose, layer ops,
        loss, optimizer) once the graph has an engine, the symbolic nodes as written before that."""
        from . import graph_def as gd
        if self._engine is not None and self._engine.model is not None:
            return self._engine.graph_def_nodes()
        return [gd.Node(n.name, n.op if n.op != "layer" else type(n.attrs["layer"]).__name__,
                        [i.name for i in n.inputs if isinstance(i, Tensor)], n.device) for n in self.nodes]

    def engine(self) -> "GraphEngine":
        if self._engine is None:
            self._engine = GraphEngine(self)
        return self._engine

    def finalize(self):
        self.finalized = True


_default_graph = Graph()
_session_stack: List[Any] = []


def get_default_graph() -> Graph:
    return _default_graph


def reset_default_graph():
    global _default_graph
    _default_graph = Graph()
    _summary.reset_collection()
    return _default_graph


def _default_session():
    return _session_stack[-1] if _session_stack else _KerasBackend._session


@contextlib.contextmanager
def device(device_name_or_function):
    """``tf.device``: records the device function; variable nodes get ``/job:ps/task:k`` strings from a
    ``replica_device_setter`` and the setter's cluster decides the training strategy."""
    g = get_default_graph()
    if isinstance(device_name_or_function, _cluster.DeviceSetter):
        g.device_setter = device_name_or_function
    g._device_stack.append(device_name_or_function)
    try:
        yield
    finally:
        g._device_stack.pop()


@contextlib.contextmanager
def name_scope(name: str):
    g = get_default_graph()
    g._scope.append(name)
    try:
        yield name
    finally:
        g._scope.pop()


def placeholder(dtype=float32, shape=None, name=None) -> Tensor:
    return Tensor("Placeholder", [], {"dtype": dtype}, name=name or "Placeholder", shape=tuple(shape) if shape else None)


def Variable(initial_value, dtype=None, name=None, trainable=True) -> VariableNode:
    return VariableNode(initial_value, dtype=dtype, name=name, trainable=trainable)


def global_variables_initializer() -> Operation:
    return Operation("init", [], name="init")


# elementwise / reductions: symbolic constructors
def round(x, name=None): return Tensor("round", [x], name=name)            # noqa: A001  (half-to-even like tf.round)
def equal(x, y, name=None): return Tensor("equal", [x, y], name=name)
def cast(x, dtype, name=None): return Tensor("cast", [x], {"dtype": _DTYPES.get(dtype, dtype)}, name=name)
def square(x, name=None): return Tensor("square", [x], name=name)
def subtract(x, y, name=None): return Tensor("sub", [x, y], name=name)
def argmax(x, axis=None, name=None): return Tensor("argmax", [x], {"axis": axis}, name=name)
def reduce_mean(x, axis=None, name=None): return Tensor("reduce_mean", [x], {"axis": axis}, name=name)
def reduce_sum(x, axis=None, name=None): return Tensor("reduce_sum", [x], {"axis": axis}, name=name)
def sigmoid(x, name=None): return Tensor("sigmoid", [x], name=name)
def tanh(x, name=None): return Tensor("tanh", [x], name=name)
def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    if transpose_a or transpose_b:
        raise NotImplementedError("tf.matmul: transposed operands are not supported by the graph front end")
    return Tensor("matmul", [a, b], name=name or "MatMul")
def add(x, y, name=None): return Tensor("add", [x, y], name=name)
def _bias_add(value, bias, name=None): return Tensor("bias_add", [value, bias], name=name or "BiasAdd")
def _relu(x, name=None): return Tensor("relu", [x], name=name or "Relu")
def _softmax(x, name=None, **_): return Tensor("softmax", [x], name=name or "Softmax")
def _nhwc4(v, what):
    v = list(v) if isinstance(v, (list, tuple)) else [1, int(v), int(v), 1]
    if len(v) == 2:
        v = [1, v[0], v[1], 1]
    if len(v) != 4 or v[0] != 1 or v[3] != 1:
        raise NotImplementedError("%s must be [1, h, w, 1] (NHWC)" % what)
    return int(v[1]), int(v[2])
def _nn_conv2d(input, filter, strides, padding, data_format="NHWC", name=None, **_):
    """``tf.nn.conv2d`` on an NHWC tensor with a ``[kh, kw, cin, cout]`` filter variable."""
    if data_format != "NHWC":
        raise NotImplementedError("tf.nn.conv2d: NHWC only")
    return Tensor("conv2d", [input, filter], {"strides": _nhwc4(strides, "strides"), "padding": str(padding).lower()},
                  name=name or "Conv2D")
def _nn_max_pool(value, ksize, strides, padding, data_format="NHWC", name=None, **_):
    if data_format != "NHWC":
        raise NotImplementedError("tf.nn.max_pool: NHWC only")
    return Tensor("max_pool", [value], {"ksize": _nhwc4(ksize, "ksize"), "strides": _nhwc4(strides, "strides"),
                                        "padding": str(padding).lower()}, name=name or "MaxPool")
def reshape(tensor, shape, name=None):
    """``tf.reshape`` with a static shape (one -1 allowed, typically the batch dimension)."""
    return Tensor("reshape", [tensor], {"shape": tuple(int(s) for s in shape)}, name=name or "Reshape")
def _nn_dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
    """``tf.nn.dropout``.  A Python ``keep_prob`` follows ``K.learning_phase()`` like the Keras layer (the reference
    script's way of switching); a fed ``keep_prob`` placeholder (the TF1 tutorials' way: 0.5 while training, 1.0 for
    evaluation) is applied as fed, whatever the learning phase - such a graph runs as a graph model."""
    if isinstance(keep_prob, Tensor):
        return Tensor("nn_dropout", [x, keep_prob], {"keep_prob": None}, name=name or "dropout")
    if not isinstance(keep_prob, (int, float)):
        raise NotImplementedError("tf.nn.dropout: keep_prob must be a Python number or a fed placeholder")
    return Tensor("nn_dropout", [x], {"keep_prob": float(keep_prob)}, name=name or "dropout")


def _layers_dense(inputs, units, activation=None, use_bias=True, kernel_initializer=None, name=None, **_):
    """``tf.layers.dense``: the functional spelling of ``keras.layers.Dense(units, activation)(inputs)``."""
    act = {None: None, _relu: "relu", sigmoid: "sigmoid", tanh: "tanh", _softmax: "softmax"}.get(activation, activation)
    return _Dense(units, activation=act, use_bias=use_bias, name=name)(inputs)


def _eval_generic(node, env: Dict[Tensor, torch.Tensor], engine: "GraphEngine", training: bool):
    """Eager evaluation of a node from fed placeholders (metric / debugging path, not the training path)."""
    if not isinstance(node, Tensor):
        return node
    if node in env:
        return env[node]
    low = getattr(engine, "lowered", {}).get(node)
    if low is not None:
        # the last node of a hand-written op group (matmul [+ bias] [+ activation]): evaluate the layer it became
        layer, src = low
        out = layer.forward(_eval_generic(src, env, engine, training), training)
        if isinstance(layer, _layers.Dense) and layer.softmax and layer.emit_logits and node.op == "softmax":
            from .ops import functional as F
            out = F.softmax(out)
        env[node] = out
        return out
    op = node.op
    ins = [_eval_generic(i, env, engine, training) for i in node.inputs]
    if op == "Placeholder":
        raise ValueError("placeholder %s must be fed" % node.name)
    if op == "layer":
        out = node.attrs["layer"].forward(ins[0], training)
        head = node.attrs["layer"]
        if isinstance(head, _layers.Dense) and head.softmax and head.emit_logits:
            from .ops import functional as F
            out = F.softmax(out)
    elif op == "round": out = torch.round(ins[0])
    elif op == "equal": out = ins[0] == ins[1]
    elif op == "cast": out = ins[0].to(node.attrs["dtype"])
    elif op == "square": out = ins[0] * ins[0]
    elif op == "sub": out = ins[0] - ins[1]
    elif op == "add": out = ins[0] + ins[1]
    elif op == "mul": out = ins[0] * ins[1]
    elif op == "div": out = ins[0] / ins[1]
    elif op == "sigmoid": out = torch.sigmoid(ins[0])
    elif op == "tanh": out = torch.tanh(ins[0])
    elif op == "relu": out = torch.relu(ins[0])
    elif op == "softmax": out = torch.softmax(ins[0], dim=-1)
    elif op == "matmul": out = ins[0] @ ins[1]
    elif op == "bias_add": out = ins[0] + ins[1]
    elif op == "nn_dropout": out = ins[0]          # evaluation outside the training path: identity
    elif op == "reshape": out = ins[0].reshape(node.attrs["shape"])
    elif op == "argmax": out = ins[0].argmax(dim=node.attrs["axis"] if node.attrs["axis"] is not None else 0)
    elif op == "reduce_mean":
        out = ins[0].float().mean() if node.attrs["axis"] is None else ins[0].float().mean(dim=node.attrs["axis"])
    elif op == "reduce_sum":
        out = ins[0].sum() if node.attrs["axis"] is None else ins[0].sum(dim=node.attrs["axis"])
    elif op == "keras_mse": out = ((ins[1] - ins[0]) ** 2).mean(dim=-1)
    elif op == "keras_bce": out = _losses.binary_crossentropy(ins[0], ins[1])
    elif op == "softmax_xent":
        out = -(ins[0] * torch.log_softmax(ins[1], dim=1)).sum(dim=1)
    elif op == "sparse_softmax_xent":
        out = torch.nn.functional.cross_entropy(ins[1], ins[0].long(), reduction="none")
    elif op == "Variable":
        out = engine.variable_value(node)
    else:
        raise NotImplementedError("cannot evaluate op %r" % op)
    env[node] = out
    return out


# --------------------------------------------------------------------------- lowering onto the engine
def _match(node, op):
    return isinstance(node, Tensor) and node.op == op


class GraphEngine(ModelEngine):
    """Lowers the default graph onto a compiled ``Sequential`` and serves ``Session.run``."""

    def __init__(self, graph: Graph):
        self.graph = graph
        self.train_op = next((n for n in graph.nodes if n.op == "minimize"), None)
        self.loss_node = self.train_op.inputs[0] if self.train_op is not None else None
        self.preds = self.labels = self.input = None
        self.accuracy_nodes: List[Tensor] = []
        self.model: Optional[_Sequential] = None
        self._last_logs = {}
        self.x_key, self.y_key, self.training_key = "x", "y", "training"
        self.global_step_node = self.train_op.attrs.get("global_step") if self.train_op is not None else None
        self._lower()

    def _lower(self):
        if self.train_op is None:
            raise ValueError("the graph has no training op (optimizer.minimize(loss) was never called)")
        self.dag = False
        self.lowered: Dict[Tensor, Any] = {}
        try:
            loss_name, labels, preds = self._match_loss(self.loss_node)
            self.labels, self.preds = labels, preds
            chain, node = self._lower_chain(preds)
        except NotImplementedError:
            # not a chain on one placeholder with one of the fused losses: differentiate the graph as it is
            return self._lower_dag()
        self.input = node
        if node.shape is None or len(node.shape) < 2:
            raise ValueError("input placeholder needs a (None, features...) shape")
        model = _Sequential(chain, name="graph_model")
        opt_node = self.train_op.attrs["optimizer"]
        strategy = self._pick_strategy(opt_node)
        metrics = []
        for n in self.graph.nodes:
            kind = self._match_accuracy(n)
            if kind:
                self.accuracy_nodes.append(n)
                metrics = ["binary_rounded_accuracy"] if kind == "rounded" else ["accuracy"]
        self._chain_input_shape = tuple(int(s) for s in node.shape[1:])
        model.layers[0].input_shape = self._chain_input_shape
        model.compile(loss=loss_name, optimizer=opt_node.optimizer, metrics=metrics or ["accuracy"],
                      strategy=strategy, device=_pick_device(), seed=self.graph.__dict__.get("seed"))
        self.model = model
        self.graph.finalize()

    def _lower_dag(self):
        """General graphs: every placeholder the loss depends on is an input, except those that only feed a loss op's
        label side (or the subtraction of a squared error), which are labels.  The model is a ``GraphModel``."""
        from .models.graph_model import GraphModel
        self.dag = True
        self.lowered = {}
        order = GraphModel._topological(self.loss_node)
        label_set = []
        for n in order:
            if n.op in ("keras_mse", "keras_bce", "softmax_xent", "sparse_softmax_xent") and _match(n.inputs[0], "Placeholder"):
                label_set.append(n.inputs[0])
            elif n.op == "sub":
                label_set += [i for i in n.inputs if _match(i, "Placeholder")]
        consumers: Dict[int, List[Tensor]] = {}
        for n in order:
            for i in n.inputs:
                if isinstance(i, Tensor):
                    consumers.setdefault(id(i), []).append(n)
        labels = [p for p in dict.fromkeys(label_set)
                  if all(c.op in ("keras_mse", "keras_bce", "softmax_xent", "sparse_softmax_xent", "sub") for c in consumers.get(id(p), []))]
        inputs = [n for n in order if n.op == "Placeholder" and n not in labels and n is not self.graph.learning_phase]
        if not inputs or not labels:
            raise NotImplementedError("the loss must depend on at least one input placeholder and one label placeholder")
        self.input_nodes, self.label_nodes = inputs, labels
        self.input, self.labels = inputs[0], labels[0]
        # the output a metric or a fetch of predictions refers to: the non-label operand of the first loss op
        self.preds = None
        for n in order:
            if n.op in ("keras_mse", "keras_bce", "softmax_xent", "sparse_softmax_xent"):
                self.preds = n.inputs[1]
                break
        acc = None
        for n in self.graph.nodes:
            if self._match_accuracy_dag(n):
                self.accuracy_nodes.append(n)
                acc = n
        model = GraphModel(self.loss_node, inputs, labels, accuracy_node=acc, output_node=self.preds)
        opt_node = self.train_op.attrs["optimizer"]
        strategy = self._pick_strategy(opt_node)
        model.loss = _losses.get("mean_squared_error")         # placeholder entry: the graph computes its own loss
        pad = strategy.var_alignment() if strategy is not None and hasattr(strategy, "var_alignment") else 64
        model.build(device=_pick_device(), seed=self.graph.__dict__.get("seed"), pad_multiple=pad)
        model.optimizer = opt_node.optimizer
        model.metrics = ["accuracy"] if acc is not None else []
        if strategy is not None:
            model.strategy = strategy
        if not getattr(model.strategy, "owns_optimizer_state", False):
            model.optimizer.create_slots(model.store)
        model.strategy.prepare(model)
        model._extra_categorical = False
        model._report_binary = False
        model.matmul_dtype = "fp32"
        model.compiled = True
        self.model = model
        self._chain_input_shape = model.input_shape
        self.graph.finalize()

    def _match_accuracy_dag(self, node) -> bool:
        """``reduce_mean(cast(equal(a, b)))`` over anything: evaluated as written."""
        return _match(node, "reduce_mean") and _match(node.inputs[0], "cast") and _match(node.inputs[0].inputs[0], "equal")

    def _lower_chain(self, preds):
        """Walk from the model output back to its input placeholder and turn what is found into layers: Keras layer
        calls as they are, and hand-written groups ``act(tf.matmul(x, W) + b)`` / ``tf.nn.bias_add`` /
        ``tf.nn.dropout`` / a lone activation into ``Dense`` / ``Dropout`` / ``Activation`` layers that keep the
        script's variable names and initial values.  Returns (layers in forward order, input placeholder)."""
        chain = []
        node = preds
        while not _match(node, "Placeholder"):
            if not isinstance(node, Tensor):
                raise NotImplementedError("the model output must be computed from a placeholder, found %r" % (node,))
            if node.op == "layer":
                chain.append(node.attrs["layer"])
                node = node.inputs[0]
                continue
            if node.op == "nn_dropout":
                if node.attrs["keep_prob"] is None:
                    raise NotImplementedError("a fed keep_prob makes the graph a graph model")
                layer = _layers.Dropout(1.0 - node.attrs["keep_prob"])
                chain.append(layer)
                self.lowered[node] = (layer, node.inputs[0])
                node = node.inputs[0]
                continue
            if node.op not in ("relu", "sigmoid", "tanh", "softmax", "add", "bias_add", "matmul"):
                raise NotImplementedError(
                    "op %r (node %r) is not supported on the model path: the graph front end lowers chains of Keras "
                    "layers, tf.layers.dense, tf.matmul(x, W) [+ b / tf.nn.bias_add] [tf.nn.relu | tf.sigmoid | "
                    "tf.nn.softmax] and tf.nn.dropout applied to one placeholder" % (node.op, node.name))
            top, act, n = node, None, node
            if n.op in ("relu", "sigmoid", "tanh", "softmax"):
                act, n = n.op, n.inputs[0]
            bias = None
            if _match(n, "add") or _match(n, "bias_add"):
                a, b = n.inputs
                if isinstance(b, VariableNode) and len(b.shape) == 1:
                    bias, n = b, a
                elif isinstance(a, VariableNode) and len(a.shape) == 1:
                    bias, n = a, b
                else:
                    raise NotImplementedError("node %r adds two activations: only `matmul + bias variable` is supported "
                                              "(no skip connections)" % n.name)
            if not _match(n, "matmul"):
                if act is None or bias is not None:
                    raise NotImplementedError("node %r: a bias without a tf.matmul in front of it" % top.name)
                layer = _layers.Activation(act)          # a lone activation after some other layer
                chain.append(layer)
                self.lowered[top] = (layer, n)
                node = n
                continue
            x, w = n.inputs
            if not (isinstance(w, VariableNode) and w.attrs["trainable"] and len(w.shape) == 2):
                raise NotImplementedError("tf.matmul(%r, %r): the second operand must be a trainable 2-D tf.Variable"
                                          % (getattr(x, "name", x), getattr(w, "name", w)))
            layer = _layers.Dense(int(w.shape[1]), activation=act, use_bias=bias is not None,
                                  kernel_initializer=w.attrs["initial_value"],
                                  bias_initializer=bias.attrs["initial_value"] if bias is not None else "zeros")
            layer.kernel_name = w.name
            layer.bias_name = bias.name if bias is not None else None
            chain.append(layer)
            self.lowered[top] = (layer, x)
            node = x
        chain.reverse()
        return chain, node

    def _pick_strategy(self, opt_node):
        srv = _cluster.current_server()
        setter = self.graph.device_setter
        if isinstance(opt_node, SyncReplicasOptimizer):
            from .parallel import Communicator, SyncReplicasStrategy, init_process_group
            import torch.distributed as dist
            if not dist.is_initialized():
                if srv is None:
                    return None
                host, port = srv.rendezvous_address()
                init_process_group(None, rank=srv.task_index, world_size=srv.num_workers, master_addr=host,
                                   master_port=port)
            comm = Communicator()
            # every step aggregates exactly one gradient from every replica (a collective has no "first N of M" mode
            # and no stale gradients to drop): any other request cannot be honoured, so it is refused, not ignored
            for what, n in (("replicas_to_aggregate", opt_node.replicas_to_aggregate),
                            ("total_num_replicas", opt_node.total_num_replicas)):
                if n is not None and int(n) != comm.world_size:
                    raise ValueError("SyncReplicasOptimizer(%s=%d) but the cluster has %d workers: synchronous replicas "
                                     "here always aggregate one gradient per worker (leave it unset or pass %d)"
                                     % (what, int(n), comm.world_size, comm.world_size))
            return SyncReplicasStrategy(comm)
        if srv is not None and setter is not None and setter.cluster is not None:
            from .parallel.async_ps import AsyncPSStrategy
            return AsyncPSStrategy(srv, max_staleness=self.graph.__dict__.get("max_staleness"),
                                   placement=setter.strategy.name)
        return None

    def _match_loss(self, node):
        if _match(node, "reduce_mean") and _match(node.inputs[0], "keras_mse"):
            m = node.inputs[0]
            return "mean_squared_error", m.inputs[0], m.inputs[1]
        if _match(node, "reduce_mean") and _match(node.inputs[0], "keras_bce"):
            m = node.inputs[0]
            return "binary_crossentropy", m.inputs[0], m.inputs[1]
        if _match(node, "reduce_mean") and _match(node.inputs[0], "softmax_xent"):
            m = node.inputs[0]
            return "categorical_crossentropy", m.inputs[0], m.inputs[1]
        if _match(node, "reduce_mean") and _match(node.inputs[0], "sparse_softmax_xent"):
            m = node.inputs[0]
            return "sparse_categorical_crossentropy", m.inputs[0], m.inputs[1]
        if _match(node, "reduce_mean") and _match(node.inputs[0], "square") and _match(node.inputs[0].inputs[0], "sub"):
            s = node.inputs[0].inputs[0]
            a, b = s.inputs
            preds, labels = (b, a) if _match(a, "Placeholder") else (a, b)      # the fed operand is the label
            return "mean_squared_error", labels, preds
        raise NotImplementedError(
            "minimize() supports reduce_mean of keras.losses.mean_squared_error / binary_crossentropy, tf.square(a - b), "
            "tf.nn.softmax_cross_entropy_with_logits or tf.nn.sparse_softmax_cross_entropy_with_logits")

    def _match_accuracy(self, node) -> Optional[str]:
        """mean(cast(equal(round(p), round(y)))) -> 'rounded'; mean(cast(equal(argmax(p), argmax(y)))) -> 'argmax'."""
        if not (_match(node, "reduce_mean") and _match(node.inputs[0], "cast") and _match(node.inputs[0].inputs[0], "equal")):
            return None
        a, b = node.inputs[0].inputs[0].inputs
        if _match(a, "round") and _match(b, "round"):
            ends = {a.inputs[0], b.inputs[0]}
            return "rounded" if ends == {self.preds, self.labels} else None
        if _match(a, "argmax") and _match(b, "argmax"):
            ends = {a.inputs[0], b.inputs[0]}
            return "argmax" if ends == {self.preds, self.labels} else None
        return None

    # ---- engine protocol
    def ensure_built(self):
        pass

    def describe(self) -> str:
        import json
        return json.dumps(self.graph.as_graph_def(), indent=1)

    def graph_def_nodes(self):
        from . import graph_def as gd
        if self.dag:
            return [gd.Node(n.name, n.op if n.op != "layer" else type(n.attrs["layer"]).__name__,
                            [i.name for i in n.inputs if isinstance(i, Tensor)], n.device) for n in self.graph.nodes]
        srv = _cluster.current_server()
        wdev = "/job:worker/task:%d" % srv.task_index if srv is not None and srv.job_name == "worker" else ""
        setter = self.graph.device_setter

        placed = {}
        if setter is not None and setter.ps_tasks > 0:
            # creation order as in the reference script: the layers' variables, then global_step
            order = [(v.name, v.numel * 4) for v in self.model.store.trainable] + [("global_step", 8)]
            placed = setter.place_variables(order)

        def vdev(name: str) -> str:
            k = placed.get(name)
            return wdev if k is None or k < 0 else "/job:ps/task:%d" % k
        return gd.model_nodes(self.model, worker_device=wdev, variable_device=vdev,
                              loss_name=getattr(self.model.loss, "name", None),
                              optimizer_name=getattr(self.model.optimizer, "name", None),
                              input_name=self.input.name, labels_name=self.labels.name)

    def variable_value(self, node):
        if node is self.global_step_node or node.name.endswith("global_step"):
            return self.global_step_value()
        v = self.model.store.variables.get(node.name) if self.model is not None else None
        if v is not None:
            return v.tensor
        raise KeyError("variable %s has no value in this graph" % node.name)

    def resolve(self, fetch):
        if fetch is self.train_op:
            return TRAIN_OP
        if fetch is self.loss_node:
            return LOSS
        if fetch in self.accuracy_nodes:
            return ACCURACY
        if fetch is self.global_step_node:
            return GLOBAL_STEP
        if isinstance(fetch, Operation) and fetch.op == "init":
            return lambda: None
        if isinstance(fetch, Tensor):
            return lambda f=fetch: self._eval_node(f)
        return fetch

    def _eval_node(self, node):
        env = dict(self._env)
        out = _eval_generic(node, env, self, self._training)
        if isinstance(out, torch.Tensor):
            return out.item() if out.dim() == 0 else out.detach().cpu().numpy()
        return out

    def run(self, fetch_struct, feed_dict=None):
        feed = {}
        self._env = {}
        self._training = False
        if self.dag:
            return self._run_dag(fetch_struct, feed_dict or {})
        for k, v in (feed_dict or {}).items():
            if k is self.input:
                feed["x"] = v
            elif k is self.labels:
                feed["y"] = v
            elif k is self.graph.learning_phase:
                feed["training"] = bool(v)
                self._training = bool(v)
            elif isinstance(k, Tensor):
                self._env[k] = self._to_device(v)
            else:
                feed[k] = v
        # one transfer per fed value: the device tensors go down to the model engine as they are
        if "x" in feed:
            feed["x"] = self._env[self.input] = self._to_device(feed["x"])
        if "y" in feed:
            feed["y"] = self._env[self.labels] = self._to_device(feed["y"])
        return super().run(fetch_struct, feed)


def _run_dag(self, fetch_struct, feed_dict):
    """``Session.run`` on a general graph: the fed placeholders are bound by identity, in placeholder order."""
    feed = {}
    for k, v in feed_dict.items():
        if k is self.graph.learning_phase:
            feed["training"] = bool(v)
            self._training = bool(v)
        elif isinstance(k, Tensor):
            self._env[k] = self._to_device(v)
        else:
            feed[k] = v
    if all(p in self._env for p in self.input_nodes):
        feed["x"] = tuple(self._env[p] for p in self.input_nodes)
    if all(p in self._env for p in self.label_nodes):
        feed["y"] = tuple(self._env[p] for p in self.label_nodes)
    return ModelEngine.run(self, fetch_struct, feed)


GraphEngine._run_dag = _run_dag


def _pick_device():
    srv = _cluster.current_server()
    if srv is not None:
        return srv.device()
    return None


# --------------------------------------------------------------------------- sessions
class _SessionWrapper:
    """Adds the ``with sess:`` default-session behaviour on top of MonitoredSession."""

    def __init__(self, inner):
        self._inner = inner

    def __enter__(self):
        self._inner.__enter__()
        _session_stack.append(self)
        return self

    def __exit__(self, *exc):
        if _session_stack and _session_stack[-1] is self:
            _session_stack.pop()
        return self._inner.__exit__(*exc)

    def __getattr__(self, name):
        return getattr(self._inner, name)


def _monitored_training_session(*args, **kwargs):
    kwargs.setdefault("engine", get_default_graph().engine())
    return _SessionWrapper(_session.MonitoredTrainingSession(*args, **kwargs))


class Supervisor:
    """``tf.train.Supervisor``, the session manager distributed TF1 scripts used before ``MonitoredTrainingSession``
    (the other way of writing what /root/reference/example.py:189-192 does)::

        sv = tf.train.Supervisor(is_chief=(task_index == 0), logdir=log_dir, global_step=global_step)
        sess = sv.prepare_or_wait_for_session(server.target)        # chief initialises / restores, the others wait
        while not sv.should_stop() and step < last: _, step = sess.run([train_op, global_step], feed)
        sv.stop()

    It is a thin front of the same monitored session: ``logdir`` is its checkpoint directory (the chief saves every
    ``save_model_secs`` and at ``stop``), ``managed_session`` is the ``with`` form.  ``init_op`` / ``saver`` /
    ``summary_op`` arguments are accepted for compatibility: initialisation, the saver and the summaries are the
    session's own."""

    def __init__(self, graph=None, is_chief: bool = True, logdir: Optional[str] = None, init_op=None, saver=None,
                 global_step=None, save_model_secs: float = 600, save_summaries_secs: float = 120,
                 recovery_wait_secs: float = 30, summary_op=None, **_):
        self.is_chief, self.logdir = bool(is_chief), logdir
        self.save_model_secs = save_model_secs
        self._graph = graph
        self._sess = None

    def prepare_or_wait_for_session(self, master: str = "", config=None, **_):
        if self._sess is None:
            g = self._graph or get_default_graph()
            # no summary hook: TF's Supervisor ran summaries from a thread of its own, which cannot feed placeholders;
            # scripts of this style write theirs through an explicit FileWriter
            inner = _session.MonitoredTrainingSession(master=master, is_chief=self.is_chief, checkpoint_dir=self.logdir,
                                                      save_checkpoint_secs=self.save_model_secs or None,
                                                      save_summaries_steps=None, engine=g.engine())
            self._sess = _SessionWrapper(inner)
            self._sess.__enter__()
        return self._sess

    def managed_session(self, master: str = "", config=None, **_):
        sv = self

        class _Managed:
            def __enter__(self_inner):
                return sv.prepare_or_wait_for_session(master)

            def __exit__(self_inner, *exc):
                sv.stop()
                return False
        return _Managed()

    def should_stop(self) -> bool:
        return self._sess is not None and self._sess.should_stop()

    def request_stop(self, ex=None):
        if self._sess is not None:
            self._sess.request_stop()

    def stop(self, *_, **__):
        """Ends the session: the chief's final checkpoint is written here."""
        if self._sess is not None:
            sess, self._sess = self._sess, None
            sess.__exit__(None, None, None)


def Session(target: str = "", graph: Optional[Graph] = None, config=None):
    """Plain ``tf.Session``: a monitored session without hooks or checkpoints."""
    g = graph or get_default_graph()
    return _SessionWrapper(_session.MonitoredSession(g.engine(), is_chief=True, master=target))


# --------------------------------------------------------------------------- optimizers (graph style)
class _GraphOptimizer:
    def __init__(self, optimizer: _optimizers.Optimizer):
        self.optimizer = optimizer

    def minimize(self, loss: Tensor, global_step: Optional[Tensor] = None, name=None) -> Operation:
        return Operation("minimize", [loss], {"optimizer": self, "global_step": global_step}, name=name or "minimize")


class AdamOptimizer(_GraphOptimizer):
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, use_locking=False, name="Adam"):
        super().__init__(_optimizers.Adam(lr=learning_rate, beta_1=beta1, beta_2=beta2, epsilon=epsilon))


class GradientDescentOptimizer(_GraphOptimizer):
    def __init__(self, learning_rate, use_locking=False, name="GradientDescent"):
        super().__init__(_optimizers.SGD(lr=learning_rate))


class MomentumOptimizer(_GraphOptimizer):
    def __init__(self, learning_rate, momentum, use_locking=False, name="Momentum"):
        super().__init__(_optimizers.SGD(lr=learning_rate, momentum=momentum))


class RMSPropOptimizer(_GraphOptimizer):
    def __init__(self, learning_rate, decay=0.9, momentum=0.0, epsilon=1e-10, use_locking=False, centered=False,
                 name="RMSProp"):
        if centered:
            raise NotImplementedError("RMSPropOptimizer(centered=True) is not supported")
        super().__init__(_optimizers.RMSProp(lr=learning_rate, decay=decay, momentum=momentum, epsilon=epsilon))


class AdagradOptimizer(_GraphOptimizer):
    def __init__(self, learning_rate, initial_accumulator_value=0.1, use_locking=False, name="Adagrad"):
        super().__init__(_optimizers.Adagrad(lr=learning_rate, initial_accumulator_value=initial_accumulator_value))


class AdadeltaOptimizer(_GraphOptimizer):
    def __init__(self, learning_rate=0.001, rho=0.95, epsilon=1e-8, use_locking=False, name="Adadelta"):
        super().__init__(_optimizers.Adadelta(lr=learning_rate, rho=rho, epsilon=epsilon))


class SyncReplicasOptimizer(_GraphOptimizer):
    """``tf.train.SyncReplicasOptimizer(opt, replicas_to_aggregate=N)``: gradients of all workers are averaged
    and applied once per global step (reduce-scatter / sharded apply / all-gather)."""

    def __init__(self, opt: _GraphOptimizer, replicas_to_aggregate: Optional[int] = None,
                 total_num_replicas: Optional[int] = None, name="sync_replicas"):
        super().__init__(opt.optimizer)
        self.replicas_to_aggregate = replicas_to_aggregate
        self.total_num_replicas = total_num_replicas

    def make_session_run_hook(self, is_chief, num_tokens=-1):
        return _session.SessionRunHook()


# --------------------------------------------------------------------------- namespaces mirroring tf.*
def _softmax_xent(labels=None, logits=None, name=None, **_):
    return Tensor("softmax_xent", [labels, logits], name=name)


def _sparse_softmax_xent(labels=None, logits=None, name=None, **_):
    return Tensor("sparse_softmax_xent", [labels, logits], name=name)


nn = types.SimpleNamespace(softmax_cross_entropy_with_logits=_softmax_xent,
                           softmax_cross_entropy_with_logits_v2=_softmax_xent,
                           sparse_softmax_cross_entropy_with_logits=_sparse_softmax_xent,
                           sigmoid=sigmoid, tanh=tanh, relu=_relu, softmax=_softmax, bias_add=_bias_add,
                           dropout=_nn_dropout, conv2d=_nn_conv2d, max_pool=_nn_max_pool)
layers = types.SimpleNamespace(dense=_layers_dense)

summary = types.SimpleNamespace(scalar=_summary.scalar, merge_all=_summary.merge_all, merge=_summary.merge,
                                FileWriter=_summary.FileWriter, Summary=_summary.Summary)

train = types.SimpleNamespace(
    AdamOptimizer=AdamOptimizer, GradientDescentOptimizer=GradientDescentOptimizer, MomentumOptimizer=MomentumOptimizer,
    RMSPropOptimizer=RMSPropOptimizer, AdagradOptimizer=AdagradOptimizer, AdadeltaOptimizer=AdadeltaOptimizer,
    SyncReplicasOptimizer=SyncReplicasOptimizer, ClusterSpec=_cluster.ClusterSpec, Server=_cluster.Server,
    replica_device_setter=_cluster.replica_device_setter, StopAtStepHook=_session.StopAtStepHook,
    StepCounterHook=_session.StepCounterHook, SummarySaverHook=_session.SummarySaverHook,
    CheckpointSaverHook=_session.CheckpointSaverHook, LoggingTensorHook=_session.LoggingTensorHook,
    SessionRunHook=_session.SessionRunHook, SessionRunArgs=_session.SessionRunArgs, Scaffold=_session.Scaffold,
    MonitoredTrainingSession=_monitored_training_session, Supervisor=Supervisor, Saver=_saver.Saver,
    latest_checkpoint=_saver.latest_checkpoint, load_checkpoint=_saver.load_checkpoint,
    list_variables=_saver.list_variables)

app = types.SimpleNamespace(flags=_flags)
flags = _flags


# --------------------------------------------------------------------------- keras namespace
class _SymbolicLayerMixin:
    """Makes a layer callable on symbolic tensors: ``Dense(128, activation='relu')(input_)``."""

    def __call__(self, x):
        if isinstance(x, Tensor):
            return Tensor("layer", [x], {"layer": self}, name=type(self).__name__.lower())
        return self.forward(x, training=False)


class _Dense(_SymbolicLayerMixin, _layers.Dense):
    pass


class _Dropout(_SymbolicLayerMixin, _layers.Dropout):
    pass


class _Activation(_SymbolicLayerMixin, _layers.Activation):
    pass


class _Flatten(_SymbolicLayerMixin, _layers.Flatten):
    pass


class _Conv2D(_SymbolicLayerMixin, _layers.Conv2D):
    pass


class _MaxPooling2D(_SymbolicLayerMixin, _layers.MaxPooling2D):
    pass


class _BatchNormalization(_SymbolicLayerMixin, _layers.BatchNormalization):
    pass


class _AveragePooling2D(_SymbolicLayerMixin, _layers.AveragePooling2D):
    pass


class _GlobalAveragePooling2D(_SymbolicLayerMixin, _layers.GlobalAveragePooling2D):
    pass


class _KerasBackend:
    _session = None

    @staticmethod
    def learning_phase() -> Tensor:
        g = get_default_graph()
        if g.learning_phase is None:
            g.learning_phase = Tensor("Placeholder", [], {"dtype": torch.bool}, name="keras_learning_phase", shape=())
        return g.learning_phase

    @staticmethod
    def set_session(sess):
        _KerasBackend._session = sess

    @staticmethod
    def get_session():
        return _KerasBackend._session


def _keras_mse(y_true, y_pred):
    if isinstance(y_true, Tensor) or isinstance(y_pred, Tensor):
        return Tensor("keras_mse", [y_true, y_pred], name="mean_squared_error")
    return _losses.mean_squared_error(y_true, y_pred)


def _keras_bce(y_true, y_pred):
    if isinstance(y_true, Tensor) or isinstance(y_pred, Tensor):
        return Tensor("keras_bce", [y_true, y_pred], name="binary_crossentropy")
    return _losses.binary_crossentropy(y_true, y_pred)


keras = types.SimpleNamespace(
    layers=types.SimpleNamespace(Dense=_Dense, Dropout=_Dropout, Activation=_Activation, Flatten=_Flatten,
                                 Conv2D=_Conv2D, MaxPooling2D=_MaxPooling2D, BatchNormalization=_BatchNormalization,
                                 AveragePooling2D=_AveragePooling2D, GlobalAveragePooling2D=_GlobalAveragePooling2D),
    models=types.SimpleNamespace(Sequential=_Sequential),
    backend=_KerasBackend,
    losses=types.SimpleNamespace(mean_squared_error=_keras_mse, binary_crossentropy=_keras_bce),
    callbacks=types.SimpleNamespace(TensorBoard=_callbacks.TensorBoard, Callback=_callbacks.Callback,
                                    History=_callbacks.History),
    optimizers=types.SimpleNamespace(Adam=_optimizers.Adam, SGD=_optimizers.SGD,
                                     RMSprop=lambda **kw: _optimizers.RMSProp(keras=True, **kw),
                                     Adagrad=lambda **kw: _optimizers.Adagrad(keras=True, **kw),
                                     Adadelta=lambda **kw: _optimizers.Adadelta(**dict({"lr": 1.0}, **kw))),
)

"""``tensorflow.GraphDef`` for a model: what ``FileWriter.add_graph`` puts into the event file (TensorBoard's graph
tab) and what ``graph.pbtxt`` in a checkpoint directory holds (/root/reference/example.py:191,195).

There is no TensorFlow graph behind a model here - a training step is a fixed list of HIP kernel launches - so the
GraphDef is *derived*: one node per variable (``VariableV2`` on the device ``replica_device_setter`` assigned), per
layer op (``MatMul`` / ``BiasAdd`` / ``Relu`` / ``Conv2D`` ...), the placeholders, the loss, the optimizer's apply ops
and ``global_step``, wired like TF would wire them.  The protos are encoded by hand (the same way the event and
checkpoint writers encode theirs)::

    GraphDef   { repeated NodeDef node = 1; VersionDef versions = 4 { int32 producer = 1 } }
    NodeDef    { string name = 1; string op = 2; repeated string input = 3; string device = 4;
                 map<string, AttrValue> attr = 5 }
    AttrValue  { DataType type = 6; TensorShapeProto shape = 7 }
    TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }

``encode_meta_graph_def`` wraps the same GraphDef into the ``MetaGraphDef`` that TF's saver writes next to every
checkpoint as ``model.ckpt-<step>.meta``::

    MetaGraphDef { MetaInfoDef meta_info_def = 1 { repeated string tags = 4; string tensorflow_version = 5 }
                   GraphDef graph_def = 2; SaverDef saver_def = 3; map<string, CollectionDef> collection_def = 4 }
    SaverDef     { string filename_tensor_name = 1; string save_tensor_name = 2; string restore_op_name = 3;
                   int32 max_to_keep = 4; bool sharded = 5; float keep_checkpoint_every_n_hours = 6; version = 7 (V2 = 2) }
    CollectionDef { NodeList node_list = 1 { repeated string value = 1 } | BytesList bytes_list = 2 { repeated bytes value = 1 } }
    VariableDef  { string variable_name = 1; string initializer_name = 2; string snapshot_name = 3; bool trainable = 7 }

Variable interoperability needs only the ``.index`` / ``.data`` files; the ``.meta`` file completes the set of files a
TF checkpoint directory holds and lets tools list the graph and its variable collections.  No TensorFlow is available
to import it with: parity of this file is unpinned, its own decoder (``parse_meta_graph_def``) is what the tests use.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

DT_FLOAT, DT_INT64, DT_BOOL = 1, 9, 10
GRAPH_DEF_PRODUCER = 24          # TensorFlow 1.4's GraphDef version


class Node:
    __slots__ = ("name", "op", "inputs", "device", "dtype", "shape")

    def __init__(self, name: str, op: str, inputs: Sequence[str] = (), device: Optional[str] = None,
                 dtype: Optional[int] = None, shape: Optional[Sequence[int]] = None):
        self.name, self.op, self.inputs, self.device = name, op, list(inputs), device or ""
        self.dtype, self.shape = dtype, (None if shape is None else [(-1 if s is None else int(s)) for s in shape])


# --------------------------------------------------------------------------- wire format
def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while n >= 128:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def _field_bytes(num: int, payload: bytes) -> bytes:
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _field_varint(num: int, value: int) -> bytes:
    return _varint((num << 3) | 0) + _varint(value)


def _attr(key: str, value: bytes) -> bytes:
    """One ``map<string, AttrValue>`` entry: { key = 1; value = 2 }."""
    return _field_bytes(5, _field_bytes(1, key.encode()) + _field_bytes(2, value))


def _shape_proto(shape: Sequence[int]) -> bytes:
    return b"".join(_field_bytes(2, _field_varint(1, int(s))) for s in shape)


def encode_node(n: Node) -> bytes:
    out = _field_bytes(1, n.name.encode()) + _field_bytes(2, n.op.encode())
    for i in n.inputs:
        out += _field_bytes(3, i.encode())
    if n.device:
        out += _field_bytes(4, n.device.encode())
    if n.dtype is not None:
        out += _attr("dtype", _field_varint(6, n.dtype))
    if n.shape is not None:
        out += _attr("shape", _field_bytes(7, _shape_proto(n.shape)))
    return out


def encode_graph_def(nodes: Sequence[Node]) -> bytes:
    """Serialized ``GraphDef``."""
    out = b"".join(_field_bytes(1, encode_node(n)) for n in nodes)
    return out + _field_bytes(4, _field_varint(1, GRAPH_DEF_PRODUCER))


def encode_meta_graph_def(nodes: Sequence[Node], variables: Sequence[Tuple[str, bool]], train_ops: Sequence[str] = (),
                          max_to_keep: int = 5, sharded: bool = True, producer: str = "distributed_tensorflow_amd") -> bytes:
    """Serialized ``MetaGraphDef``.  ``variables``: ``(name, trainable)`` of every saved variable."""
    import struct
    meta_info = _field_bytes(4, b"train") + _field_bytes(5, producer.encode())
    saver = (_field_bytes(1, b"save/Const:0") + _field_bytes(2, b"save/Identity:0" if sharded else b"save/control_dependency:0")
             + _field_bytes(3, b"save/restore_all") + _field_varint(4, int(max_to_keep)) + _field_varint(5, 1 if sharded else 0)
             + _varint((6 << 3) | 5) + struct.pack("<f", 10000.0) + _field_varint(7, 2))

    def variable_def(name: str, trainable: bool) -> bytes:
        out = _field_bytes(1, (name + ":0").encode()) + _field_bytes(2, (name + "/Assign").encode()) \
            + _field_bytes(3, (name + "/read:0").encode())
        return out + (_field_varint(7, 1) if trainable else b"")

    def collection(key: str, value: bytes) -> bytes:
        return _field_bytes(4, _field_bytes(1, key.encode()) + _field_bytes(2, value))

    def bytes_list(items: Sequence[bytes]) -> bytes:
        return _field_bytes(2, b"".join(_field_bytes(1, i) for i in items))

    out = _field_bytes(1, meta_info) + _field_bytes(2, encode_graph_def(nodes)) + _field_bytes(3, saver)
    out += collection("variables", bytes_list([variable_def(n, t) for n, t in variables]))
    out += collection("trainable_variables", bytes_list([variable_def(n, t) for n, t in variables if t]))
    if train_ops:
        out += collection("train_op", _field_bytes(1, b"".join(_field_bytes(1, t.encode()) for t in train_ops)))
    return out


_DT_NAMES = {DT_FLOAT: "DT_FLOAT", DT_INT64: "DT_INT64", DT_BOOL: "DT_BOOL"}


def graph_def_text(nodes: Sequence[Node]) -> str:
    """The same GraphDef in protobuf text format (``graph.pbtxt``)."""
    lines: List[str] = []
    for n in nodes:
        lines.append("node {")
        lines.append('  name: "%s"' % n.name)
        lines.append('  op: "%s"' % n.op)
        for i in n.inputs:
            lines.append('  input: "%s"' % i)
        if n.device:
            lines.append('  device: "%s"' % n.device)
        if n.dtype is not None:
            lines += ["  attr {", '    key: "dtype"', "    value {", "      type: %s" % _DT_NAMES.get(n.dtype, str(n.dtype)),
                      "    }", "  }"]
        if n.shape is not None:
            lines += ["  attr {", '    key: "shape"', "    value {", "      shape {"]
            for s in n.shape:
                lines += ["        dim {", "          size: %d" % s, "        }"]
            lines += ["      }", "    }", "  }"]
        lines.append("}")
    lines += ["versions {", "  producer: %d" % GRAPH_DEF_PRODUCER, "}"]
    return "\n".join(lines) + "\n"


# --------------------------------------------------------------------------- reading back (tests, tools)
def parse_graph_def(data: bytes) -> List[Dict]:
    """Minimal independent decoder: ``[{name, op, input: [...], device, attr: {key: raw bytes}}]``."""
    def fields(buf: bytes):
        pos = 0
        while pos < len(buf):
            key, pos = _read_varint(buf, pos)
            num, wt = key >> 3, key & 7
            if wt == 0:
                v, pos = _read_varint(buf, pos)
            elif wt == 2:
                ln, pos = _read_varint(buf, pos)
                v, pos = buf[pos:pos + ln], pos + ln
            elif wt == 1:
                v, pos = buf[pos:pos + 8], pos + 8
            elif wt == 5:
                v, pos = buf[pos:pos + 4], pos + 4
            else:
                raise ValueError("unsupported wire type %d" % wt)
            yield num, v

    nodes = []
    for num, v in fields(data):
        if num != 1:
            continue
        node = {"name": "", "op": "", "input": [], "device": "", "attr": {}}
        for fnum, fv in fields(v):
            if fnum == 1:
                node["name"] = fv.decode()
            elif fnum == 2:
                node["op"] = fv.decode()
            elif fnum == 3:
                node["input"].append(fv.decode())
            elif fnum == 4:
                node["device"] = fv.decode()
            elif fnum == 5:
                kv = dict(fields(fv))
                node["attr"][kv[1].decode()] = kv.get(2, b"")
        nodes.append(node)
    return nodes


def parse_meta_graph_def(data: bytes) -> Dict:
    """Independent decoder of what ``encode_meta_graph_def`` writes: ``{tags, producer, nodes, saver: {...},
    collections: {key: [variable names] | [node names]}}``."""
    import struct

    def fields(buf: bytes):
        pos = 0
        while pos < len(buf):
            key, pos = _read_varint(buf, pos)
            num, wt = key >> 3, key & 7
            if wt == 0:
                v, pos = _read_varint(buf, pos)
            elif wt == 2:
                ln, pos = _read_varint(buf, pos)
                v, pos = buf[pos:pos + ln], pos + ln
            elif wt == 5:
                v, pos = buf[pos:pos + 4], pos + 4
            else:
                raise ValueError("unsupported wire type %d" % wt)
            yield num, v

    out: Dict = {"tags": [], "producer": "", "nodes": [], "saver": {}, "collections": {}}
    for num, v in fields(data):
        if num == 1:
            for f, fv in fields(v):
                if f == 4:
                    out["tags"].append(fv.decode())
                elif f == 5:
                    out["producer"] = fv.decode()
        elif num == 2:
            out["nodes"] = parse_graph_def(v)
        elif num == 3:
            names = {1: "filename_tensor_name", 2: "save_tensor_name", 3: "restore_op_name", 4: "max_to_keep", 5: "sharded",
                     6: "keep_checkpoint_every_n_hours", 7: "version"}
            for f, fv in fields(v):
                out["saver"][names[f]] = fv.decode() if f <= 3 else (struct.unpack("<f", fv)[0] if f == 6 else fv)
        elif num == 4:
            kv = dict(fields(v))
            key, items = kv[1].decode(), []
            for kind, body in fields(kv[2]):
                for _, item in fields(body):
                    items.append(dict(fields(item))[1].decode() if kind == 2 else item.decode())
            out["collections"][key] = items
    return out


def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


# --------------------------------------------------------------------------- deriving the nodes of a model
_ACT_OPS = {"relu": "Relu", "sigmoid": "Sigmoid", "softmax": "Softmax", "tanh": "Tanh"}


def model_nodes(model, worker_device: str = "", variable_device=None, loss_name: Optional[str] = None,
                optimizer_name: Optional[str] = None, input_name: str = "Placeholder",
                labels_name: str = "Placeholder_1") -> List[Node]:
    """Nodes of a ``Sequential``-like model (``model.layers``): placeholders, per layer the variables and the ops TF
    would create, then - when ``loss_name`` is given - the loss, ``global_step`` and the optimizer's apply ops.

    ``variable_device(name) -> str`` gives the device of a variable (``/job:ps/task:k``); everything else is placed on
    ``worker_device``.
    """
    from . import layers as L
    vdev = variable_device or (lambda name: worker_device)
    nodes: List[Node] = []
    in_shape = getattr(model, "input_shape", None) or (model.layers[0].input_shape if model.layers else None)
    nodes.append(Node(input_name, "Placeholder", device=worker_device, dtype=DT_FLOAT,
                      shape=[None] + list(in_shape or [])))
    cur = input_name
    var_names: List[str] = []
    has_dropout = any(isinstance(l, L.Dropout) for l in model.layers)
    if has_dropout:
        nodes.append(Node("keras_learning_phase", "Placeholder", device=worker_device, dtype=DT_BOOL, shape=[]))
    for layer in model.layers:
        name = layer.name or type(layer).__name__.lower()
        act = "softmax" if getattr(layer, "softmax", False) else getattr(layer, "activation", None)
        if isinstance(layer, (L.Dense, L.Conv2D)):
            kern, bias = getattr(layer, "kernel", None), getattr(layer, "bias", None)
            kname = "%s/kernel" % name
            nodes.append(Node(kname, "VariableV2", device=vdev(kname), dtype=DT_FLOAT,
                              shape=list(kern.shape) if kern is not None else None))
            nodes.append(Node(kname + "/read", "Identity", [kname], device=vdev(kname)))
            var_names.append(kname)
            op = "MatMul" if isinstance(layer, L.Dense) else "Conv2D"
            nodes.append(Node("%s/%s" % (name, op), op, [cur, kname + "/read"], device=worker_device))
            cur = "%s/%s" % (name, op)
            if bias is not None:
                bname = "%s/bias" % name
                nodes.append(Node(bname, "VariableV2", device=vdev(bname), dtype=DT_FLOAT, shape=list(bias.shape)))
                nodes.append(Node(bname + "/read", "Identity", [bname], device=vdev(bname)))
                var_names.append(bname)
                nodes.append(Node("%s/BiasAdd" % name, "BiasAdd", [cur, bname + "/read"], device=worker_device))
                cur = "%s/BiasAdd" % name
            if isinstance(act, str) and act in _ACT_OPS:
                nodes.append(Node("%s/%s" % (name, _ACT_OPS[act]), _ACT_OPS[act], [cur], device=worker_device))
                cur = "%s/%s" % (name, _ACT_OPS[act])
        elif isinstance(layer, L.Dropout):
            nodes.append(Node("%s/cond/dropout/mul" % name, "Mul", [cur, "keras_learning_phase"], device=worker_device))
            cur = "%s/cond/dropout/mul" % name
        elif isinstance(layer, L.MaxPooling2D):
            nodes.append(Node("%s/MaxPool" % name, "MaxPool", [cur], device=worker_device))
            cur = "%s/MaxPool" % name
        elif isinstance(layer, L.Flatten):
            nodes.append(Node("%s/Reshape" % name, "Reshape", [cur], device=worker_device))
            cur = "%s/Reshape" % name
        elif isinstance(layer, L.Activation):
            op = _ACT_OPS.get(str(act), "Identity")
            nodes.append(Node("%s/%s" % (name, op), op, [cur], device=worker_device))
            cur = "%s/%s" % (name, op)
        else:
            nodes.append(Node(name, type(layer).__name__, [cur], device=worker_device))
            cur = name
    if loss_name:
        out_shape = getattr(model, "output_shape", None)
        nodes.append(Node(labels_name, "Placeholder", device=worker_device, dtype=DT_FLOAT,
                          shape=[None] + list(out_shape or [])))
        nodes.append(Node("loss/%s" % loss_name, {"mean_squared_error": "SquaredDifference",
                                                  "binary_crossentropy": "SigmoidCrossEntropyWithLogits"}.get(
            loss_name, "SoftmaxCrossEntropyWithLogits"), [cur, labels_name], device=worker_device))
        nodes.append(Node("loss/Mean", "Mean", ["loss/%s" % loss_name], device=worker_device))
        gs_dev = vdev("global_step")
        nodes.append(Node("global_step", "VariableV2", device=gs_dev, dtype=DT_INT64, shape=[]))
        nodes.append(Node("gradients", "NoOp", ["loss/Mean"], device=worker_device))
        opt = optimizer_name or "Adam"
        updates = []
        for v in var_names:
            slots = {"Adam": ["Adam", "Adam_1"], "SGD": [], "RMSProp": ["RMSProp", "RMSProp_1"],
                     "Adagrad": ["Adagrad"], "Adadelta": ["Adadelta", "Adadelta_1"]}.get(opt, [])
            for s in slots:
                nodes.append(Node("%s/%s" % (v, s), "VariableV2", device=vdev(v), dtype=DT_FLOAT))
            up = "%s/update_%s/Apply%s" % (opt, v, {"SGD": "GradientDescent"}.get(opt, opt))
            nodes.append(Node(up, "Apply%s" % {"SGD": "GradientDescent"}.get(opt, opt),
                              [v] + ["%s/%s" % (v, s) for s in slots] + ["gradients"], device=vdev(v)))
            updates.append(up)
        if opt == "Adam":
            first = vdev(var_names[0]) if var_names else gs_dev
            nodes.append(Node("beta1_power", "VariableV2", device=first, dtype=DT_FLOAT, shape=[]))
            nodes.append(Node("beta2_power", "VariableV2", device=first, dtype=DT_FLOAT, shape=[]))
        nodes.append(Node(opt, "AssignAdd", ["global_step"] + ["^" + u for u in updates], device=gs_dev))
    return nodes


def nodes_for(obj) -> List[Node]:
    """GraphDef nodes for whatever ``add_graph`` was handed: an object with ``graph_def_nodes()`` (compat graphs,
    engines), a ``Sequential``, a list of :class:`Node`, or anything else (one ``NoOp`` node named after its repr)."""
    if isinstance(obj, (list, tuple)) and all(isinstance(n, Node) for n in obj):
        return list(obj)
    fn = getattr(obj, "graph_def_nodes", None)
    if callable(fn):
        return fn()
    if hasattr(obj, "layers"):
        loss = getattr(getattr(obj, "loss", None), "name", None)
        opt = getattr(getattr(obj, "optimizer", None), "name", None)
        return model_nodes(obj, loss_name=loss, optimizer_name=opt)
    if isinstance(obj, dict) and "nodes" in obj:
        out = []
        for n in obj["nodes"]:
            if isinstance(n, dict):
                out.append(Node(str(n.get("name", "node")), str(n.get("op", "NoOp")), n.get("inputs", []), n.get("device")))
            else:
                out.append(Node(str(n), "NoOp"))
        return out
    return [Node("graph", "NoOp")]

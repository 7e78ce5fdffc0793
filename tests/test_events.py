"""Event files: TFRecord framing, Event/Summary protos, async writer, TensorBoard callback."""
import glob
import os
import struct

from distributed_tensorflow_amd import summary as sm
from tests.test_tensor_bundle import crc32c, mask


def test_record_framing_and_contents(tmp_path):
    w = sm.FileWriter(str(tmp_path), filename_suffix=".t")
    w.add_summary(sm.Summary([("loss", 0.25), ("accuracy", 0.5)]), global_step=3.9)   # float step truncates to 3
    w.add_scalar("global_step/sec", 123.0, 100)
    w.add_graph({"nodes": ["a", "b"]})
    w.add_histogram("staleness", [0, 1, 2, 3], [5, 3, 1, 0], global_step=7, minimum=0, maximum=2, total=5.0)
    w.add_summary(sm.Summary([("x", 1.0)]).SerializeToString(), 9)                      # raw serialized Summary
    w.flush()
    name = os.path.basename(w.path)
    assert name.startswith("events.out.tfevents.") and name.endswith(".t")
    raw = open(w.path, "rb").read()
    # first record by hand: u64 len | masked crc(len) | payload | masked crc(payload)
    (n,) = struct.unpack_from("<Q", raw, 0)
    assert struct.unpack_from("<I", raw, 8)[0] == mask(crc32c(raw[:8]))
    payload = raw[12:12 + n]
    assert struct.unpack_from("<I", raw, 12 + n)[0] == mask(crc32c(payload))
    assert payload[0] == 0x09 and payload[9:11] == bytes([0x1A, 13]) and payload[11:] == b"brain.Event:2"
    ev = sm.read_events(w.path)
    assert ev[0]["file_version"] == "brain.Event:2"
    assert ev[1]["step"] == 3 and abs(ev[1]["scalars"]["loss"] - 0.25) < 1e-7 and ev[1]["scalars"]["accuracy"] == 0.5
    assert ev[2]["step"] == 100 and ev[2]["scalars"] == {"global_step/sec": 123.0}
    from distributed_tensorflow_amd.graph_def import parse_graph_def
    assert [n["name"] for n in parse_graph_def(ev[3]["graph_def"])] == ["a", "b"]
    h = ev[4]["histograms"]["staleness"]
    assert h["bucket"] == [5, 3, 1, 0] and h["bucket_limit"] == [0, 1, 2, 3] and h["num"] == 9
    assert ev[5]["step"] == 9 and ev[5]["scalars"] == {"x": 1.0}
    w.close()


def test_many_records_async_and_unique_names(tmp_path):
    a, b = sm.FileWriter(str(tmp_path)), sm.FileWriter(str(tmp_path))
    assert a.path != b.path                      # two writers opened in the same second never share a file
    for i in range(5000):
        a.add_scalar("v", float(i), i)
    a.close()
    b.close()
    ev = sm.read_events(a.path)
    assert len(ev) == 5001 and [e["step"] for e in ev[1:]] == list(range(5000))


def test_merge_all_collection(tmp_path):
    sm.reset_collection()
    vals = {"a": 1.0, "b": 2.0}
    sm.scalar("a", lambda: vals["a"])
    sm.scalar("b", lambda: vals["b"])
    merged = sm.merge_all()
    assert merged.evaluate().value == [("a", 1.0), ("b", 2.0)]
    w = sm.FileWriter(str(tmp_path))
    w.add_summary(merged, 1)
    w.close()
    assert sm.read_events(w.path)[1]["scalars"] == {"a": 1.0, "b": 2.0}
    sm.reset_collection()
    assert sm.merge_all() is None


def test_tensorboard_callback(tmp_path):
    from distributed_tensorflow_amd.callbacks import TensorBoard
    from distributed_tensorflow_amd.data import get_data
    from distributed_tensorflow_amd.models import xor_mlp
    xt, yt, xv, yv = get_data(500, val_set_size=100, seed=0)
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", metrics=["accuracy"], device="cpu", seed=0)
    h = m.fit(xt, yt, epochs=3, batch_size=50, validation_data=(xv, yv), callbacks=[TensorBoard(log_dir=str(tmp_path))],
              verbose=0)
    assert set(h.history) == {"loss", "acc", "val_loss", "val_acc"} and len(h.history["loss"]) == 3
    (path,) = glob.glob(os.path.join(str(tmp_path), "events.out.tfevents.*"))
    ev = sm.read_events(path)
    steps = [e["step"] for e in ev if "scalars" in e and "val_acc" in e["scalars"]]
    assert steps == [0, 1, 2] and any("graph_def" in e for e in ev)


def test_add_graph_writes_a_real_graph_def(tmp_path):
    """``writer.add_graph(sess.graph)`` (/root/reference/example.py:195): the event holds a serialized GraphDef whose
    nodes an independent protobuf decoder finds - the layer variables (on their ps devices), the layer ops, the loss,
    the optimizer - and ``graph.pbtxt`` holds the same graph in text format."""
    from distributed_tensorflow_amd import graph_def as gd
    from distributed_tensorflow_amd.engine import ModelEngine
    from distributed_tensorflow_amd.models import xor_mlp
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", metrics=["accuracy"], device="cpu", seed=0)
    w = sm.FileWriter(str(tmp_path))
    w.add_graph(ModelEngine(m))
    w.close()
    ev = [e for e in sm.read_events(w.path) if "graph_def" in e]
    nodes = {n["name"]: n for n in gd.parse_graph_def(ev[0]["graph_def"])}
    for name, op in (("dense_1/kernel", "VariableV2"), ("dense_1/MatMul", "MatMul"), ("dense_1/BiasAdd", "BiasAdd"),
                     ("dense_1/Relu", "Relu"), ("dense_3/Sigmoid", "Sigmoid"), ("global_step", "VariableV2"),
                     ("dense_2/kernel/Adam_1", "VariableV2"), ("Adam", "AssignAdd"), ("Placeholder", "Placeholder")):
        assert nodes[name]["op"] == op, name
    assert nodes["dense_1/MatMul"]["input"] == ["Placeholder", "dense_1/kernel/read"]
    assert nodes["dense_2/MatMul"]["input"][0] == "dropout_1/cond/dropout/mul"
    assert "dtype" in nodes["dense_1/kernel"]["attr"] and "shape" in nodes["dense_1/kernel"]["attr"]
    # shape attr: AttrValue{ shape(7) { dim(2){size(1)=64} dim{size=128} } }
    assert nodes["dense_1/kernel"]["attr"]["shape"] == bytes([0x3A, 9, 0x12, 2, 0x08, 64, 0x12, 3, 0x08, 0x80, 0x01])
    text = gd.graph_def_text(gd.nodes_for(ModelEngine(m)))
    assert 'name: "dense_1/kernel"' in text and 'op: "VariableV2"' in text and "producer: 24" in text
    assert text.count("node {") == len(nodes)


def test_graph_def_of_the_reference_script_places_variables_round_robin():
    """A compat graph built under ``replica_device_setter`` with 2 ps tasks: variables alternate between the ps tasks
    in creation order (TF's default strategy), ops stay on the worker."""
    from distributed_tensorflow_amd import compat as tf
    from distributed_tensorflow_amd import graph_def as gd
    from distributed_tensorflow_amd import cluster
    from distributed_tensorflow_amd.compat import keras
    tf.reset_default_graph()
    cluster._CURRENT_SERVER = None        # no task runtime in this process: the graph is only described, not trained
    setter = tf.train.replica_device_setter(worker_device="/job:worker/task:0",
                                            cluster={"ps": ["h:1", "h:2"], "worker": ["h:3"]})
    with tf.device(setter):
        x = tf.placeholder(tf.float32, shape=(None, 64))
        h = keras.layers.Dense(128, activation="relu")(x)
        p = keras.layers.Dense(32, activation="sigmoid")(h)
        y = tf.placeholder(tf.float32, shape=(None, 32))
        loss = tf.reduce_mean(keras.losses.mean_squared_error(y, p))
        tf.train.AdamOptimizer().minimize(loss, global_step=tf.Variable(0, dtype=tf.int64, name="global_step", trainable=False))
    g = tf.get_default_graph()
    g.engine()
    nodes = {n["name"]: n for n in gd.parse_graph_def(gd.encode_graph_def(g.graph_def_nodes()))}
    devs = [nodes[k]["device"] for k in ("dense_1/kernel", "dense_1/bias", "dense_2/kernel", "dense_2/bias")]
    assert devs == ["/job:ps/task:0", "/job:ps/task:1", "/job:ps/task:0", "/job:ps/task:1"]
    assert nodes["dense_1/kernel/Adam"]["device"] == "/job:ps/task:0"          # slots colocated with their variable
    tf.reset_default_graph()

"""Graph-style front end: the structure of /root/reference/example.py running on the engine (CPU)."""
import glob
import os

import pytest

from distributed_tensorflow_amd import compat as tf
from distributed_tensorflow_amd.compat import keras
from distributed_tensorflow_amd.data import get_data
from distributed_tensorflow_amd.saver import latest_checkpoint, list_variables
from distributed_tensorflow_amd.summary import read_events

K = keras.backend
Dense, Dropout = keras.layers.Dense, keras.layers.Dropout
BITS = 32


def build_graph():
    tf.reset_default_graph()
    input_ = tf.placeholder(tf.float32, shape=(None, 2 * BITS))
    x = Dense(128, activation="relu")(input_)
    x = Dropout(0.3)(x)
    x = Dense(128, activation="relu")(x)
    x = Dropout(0.3)(x)
    preds = Dense(BITS, activation="sigmoid")(x)
    labels = tf.placeholder(tf.float32, shape=(None, BITS))
    with tf.name_scope("accuracy"):
        accuracy = tf.reduce_mean(tf.cast(tf.equal(tf.round(preds), tf.round(labels)), "float"))
        tf.summary.scalar("accuracy", accuracy)
    with tf.name_scope("loss"):
        loss = tf.reduce_mean(keras.losses.mean_squared_error(labels, preds))
        tf.summary.scalar("loss", loss)
    optimizer = tf.train.AdamOptimizer()
    global_step = tf.Variable(0, dtype=tf.int64, name="global_step", trainable=False)
    train_step = optimizer.minimize(loss, global_step=global_step)
    summ = tf.summary.merge_all()
    return dict(input=input_, labels=labels, preds=preds, accuracy=accuracy, loss=loss, train=train_step, summ=summ,
                global_step=global_step)


def train(g, logdir, last_step, data, steps_per_epoch=20):
    xt, yt, xv, yv = data
    writer = tf.summary.FileWriter(logdir)
    hooks = [tf.train.StopAtStepHook(last_step=last_step)]
    seen = []
    with tf.train.MonitoredTrainingSession(master="", is_chief=True, checkpoint_dir=logdir, hooks=hooks) as sess:
        writer.add_graph(sess.graph)
        start = sess.global_step_value()
        epoch = 0
        while not sess.should_stop():
            epoch += 1
            for i in range(steps_per_epoch):
                if not sess.should_stop():
                    bx, by = xt[i * 50:(i + 1) * 50], yt[i * 50:(i + 1) * 50]
                    acc, loss, s, _ = sess.run([g["accuracy"], g["loss"], g["summ"], g["train"]],
                                               feed_dict={g["input"]: bx, g["labels"]: by, K.learning_phase(): 1})
                    seen.append((acc, loss))
                    writer.add_summary(s, epoch + float(i) / steps_per_epoch)
        val = g["accuracy"].eval({g["input"]: xv, g["labels"]: yv, K.learning_phase(): 0}, session=sess)
        end = sess.global_step_value()
    writer.close()
    return start, end, seen, val


def test_reference_style_training_checkpoint_and_resume(tmp_path):
    data = get_data(1000, seed=3, as_lists=True)
    logdir = str(tmp_path / "logs")
    g = build_graph()
    start, end, seen, val = train(g, logdir, 60, data)
    assert (start, end) == (0, 60)
    assert len(seen) == 60 and all(0.0 <= a <= 1.0 for a, _ in seen)
    assert seen[-1][1] < seen[0][1] + 1e-3          # loss does not blow up
    assert 0.3 < val < 1.0

    # files TF would leave in checkpoint_dir
    names = sorted(os.listdir(logdir))
    assert "checkpoint" in names and "graph.pbtxt" in names
    ck = latest_checkpoint(logdir)
    assert ck.endswith("model.ckpt-60")
    variables = dict(list_variables(ck))
    assert variables["dense_1/kernel"] == (64, 128) and variables["dense_3/bias"] == (32,)
    assert variables["global_step"] == () and "beta1_power" in variables
    assert "dense_2/kernel/Adam" in variables and "dense_2/kernel/Adam_1" in variables
    # ... including the MetaGraphDef next to the checkpoint: graph, SaverDef and the variable collections
    from distributed_tensorflow_amd import graph_def as gd
    assert "model.ckpt-60.meta" in names
    meta = gd.parse_meta_graph_def(open(ck + ".meta", "rb").read())
    assert meta["tags"] == ["train"] and meta["saver"]["version"] == 2 and meta["saver"]["max_to_keep"] == 5
    assert meta["saver"]["restore_op_name"] == "save/restore_all"
    assert {n["name"] for n in meta["nodes"]} >= {"dense_1/kernel", "dense_1/MatMul", "global_step"}
    assert "dense_3/bias:0" in meta["collections"]["trainable_variables"]
    assert "global_step:0" in meta["collections"]["variables"] \
        and "global_step:0" not in meta["collections"]["trainable_variables"]
    assert meta["collections"]["train_op"]

    # event files: the explicit writer (epoch-numbered steps) and the chief's hooks (true global steps)
    events = [read_events(p) for p in glob.glob(os.path.join(logdir, "events.out.tfevents.*"))]
    assert len(events) >= 2 and all(e[0]["file_version"] == "brain.Event:2" for e in events)
    scalars = [ev for e in events for ev in e if "scalars" in ev]
    assert any("loss" in ev["scalars"] and "accuracy" in ev["scalars"] for ev in scalars)
    assert max(ev["step"] for ev in scalars) <= 60

    # a re-run CONTINUES from the restored global_step (absolute last_step semantics)
    g2 = build_graph()
    start2, end2, seen2, _ = train(g2, logdir, 90, data)
    assert (start2, end2) == (60, 90) and len(seen2) == 30
    assert latest_checkpoint(logdir).endswith("model.ckpt-90")
    # and a run whose budget is already spent does nothing
    g3 = build_graph()
    start3, end3, seen3, _ = train(g3, logdir, 90, data)
    assert (start3, end3, len(seen3)) == (90, 90, 0)


def test_unsupported_loss_is_reported():
    tf.reset_default_graph()
    x = tf.placeholder(tf.float32, shape=(None, 4))
    y = Dense(2)(x)
    loss = tf.reduce_sum(y)
    tf.train.AdamOptimizer().minimize(loss)
    with pytest.raises(NotImplementedError):
        tf.get_default_graph().engine()


def test_hand_written_matmul_graph_trains_and_keeps_variable_names(tmp_path):
    """A TF1 script that spells its layers out - tf.Variable / tf.matmul / + / tf.nn.relu / tf.nn.dropout and a linear
    head under softmax cross-entropy - lowers onto the same engine; checkpoints carry the script's variable names."""
    import torch
    from distributed_tensorflow_amd.data import mnist_like
    tf.reset_default_graph()
    tf.get_default_graph().seed = 7
    x = tf.placeholder(tf.float32, shape=(None, 784))
    y = tf.placeholder(tf.float32, shape=(None, 10))
    W1 = tf.Variable(tf.truncated_normal([784, 64], stddev=0.05), name="W1")
    b1 = tf.Variable(tf.zeros([64]), name="b1")
    W2 = tf.Variable(tf.random_normal([64, 32], stddev=0.1), name="W2")
    b2 = tf.Variable(tf.constant(0.1, shape=[32]), name="b2")
    W3 = tf.Variable(tf.random_normal([32, 10], stddev=0.1), name="W3")
    b3 = tf.Variable(tf.zeros([10]), name="b3")
    h1 = tf.nn.dropout(tf.nn.relu(tf.matmul(x, W1) + b1), 0.8)
    h2 = tf.nn.relu(tf.nn.bias_add(tf.matmul(h1, W2), b2))
    logits = tf.add(tf.matmul(h2, W3), b3)
    loss = tf.reduce_mean(tf.nn.softmax_cross_entropy_with_logits(labels=y, logits=logits))
    accuracy = tf.reduce_mean(tf.cast(tf.equal(tf.argmax(logits, 1), tf.argmax(y, 1)), tf.float32))
    global_step = tf.Variable(0, dtype=tf.int64, name="global_step", trainable=False)
    train = tf.train.GradientDescentOptimizer(0.2).minimize(loss, global_step=global_step)
    xs, ys = mnist_like(512, seed=1, one_hot=True)
    losses = []
    with tf.train.MonitoredTrainingSession(checkpoint_dir=str(tmp_path), hooks=[tf.train.StopAtStepHook(last_step=40)]) as sess:
        eng = tf.get_default_graph().engine()
        layers = eng.model.layers
        assert [type(l).__name__ for l in layers] == ["Dense", "Dropout", "Dense", "Dense"]
        assert [l.units for l in layers if hasattr(l, "units")] == [64, 32, 10] and layers[0].activation == "relu"
        assert torch.allclose(eng.model.store.variables["b2"].tensor, torch.full((32,), 0.1))
        assert float(eng.model.store.variables["W1"].tensor.abs().max()) <= 0.1 + 1e-6      # truncated at 2 sigma
        i = 0
        while not sess.should_stop():
            lo = (i * 64) % 448
            l, a, _ = sess.run([loss, accuracy, train], {x: xs[lo:lo + 64], y: ys[lo:lo + 64], K.learning_phase(): 1})
            losses.append(l)
            i += 1
        acc = accuracy.eval({x: xs, y: ys, K.learning_phase(): 0}, session=sess)
        out = sess.run(logits, {x: xs[:8], K.learning_phase(): 0})          # a hand-written node evaluates through its layer
        assert out.shape == (8, 10)
    assert losses[-1] < losses[0] and acc > 0.5
    names = dict(list_variables(latest_checkpoint(str(tmp_path))))
    assert names["W1"] == (784, 64) and names["b3"] == (10,) and names["global_step"] == ()
    tf.reset_default_graph()


def _torch_reference(weights, feeds, loss_fn, steps, lr=1e-3):
    """Plain PyTorch (fp32, autograd, TF-formula Adam) on copies of the model's initial weights."""
    import torch
    ws = {k: v.clone().requires_grad_() for k, v in weights.items()}
    m = {k: torch.zeros_like(v) for k, v in ws.items()}
    v2 = {k: torch.zeros_like(v) for k, v in ws.items()}
    losses = []
    for t in range(1, steps + 1):
        loss = loss_fn(ws, *feeds)
        for w in ws.values():
            w.grad = None
        loss.backward()
        losses.append(float(loss.detach()))
        lr_t = lr * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t)
        with torch.no_grad():
            for k, w in ws.items():
                m[k].mul_(0.9).add_(w.grad, alpha=0.1)
                v2[k].mul_(0.999).addcmul_(w.grad, w.grad, value=0.001)
                w.sub_(lr_t * m[k] / (v2[k].sqrt() + 1e-8))
    return losses


def test_residual_mlp_matches_pytorch_reference(tmp_path):
    """A skip connection and a non-layer op on the model path: the graph is differentiated as written (reverse mode
    over the autograd wrappers of the fused kernels) and follows a plain PyTorch implementation step for step."""
    import torch
    tf.reset_default_graph()
    x = tf.placeholder(tf.float32, shape=(None, 16))
    y = tf.placeholder(tf.float32, shape=(None, 16))
    h = Dense(16, activation="relu")(x)
    h2 = Dense(16, activation="relu")(h + x)               # residual add feeding the next layer
    p = Dense(16, activation="sigmoid")(h2 + h)
    loss = tf.reduce_mean(keras.losses.mean_squared_error(y, p))
    acc = tf.reduce_mean(tf.cast(tf.equal(tf.round(p), tf.round(y)), "float"))
    gs = tf.Variable(0, dtype=tf.int64, name="global_step", trainable=False)
    train_op = tf.train.AdamOptimizer().minimize(loss, global_step=gs)
    eng = tf.get_default_graph().engine()
    assert eng.dag
    names = [v.name for v in eng.model.store.trainable]
    assert names == ["dense_1/kernel", "dense_1/bias", "dense_2/kernel", "dense_2/bias", "dense_3/kernel", "dense_3/bias"]
    w0 = {v.name: v.tensor.clone() for v in eng.model.store.trainable}
    g = torch.Generator().manual_seed(0)
    bx = torch.rand(64, 16, generator=g)
    by = (torch.rand(64, 16, generator=g) > 0.5).float()

    def ref_loss(w, bx, by):
        h = torch.relu(bx @ w["dense_1/kernel"] + w["dense_1/bias"])
        h2 = torch.relu((h + bx) @ w["dense_2/kernel"] + w["dense_2/bias"])
        p = torch.sigmoid((h2 + h) @ w["dense_3/kernel"] + w["dense_3/bias"])
        return ((p - by) ** 2).mean()

    want = _torch_reference(w0, (bx, by), ref_loss, 40)
    got = []
    with tf.train.MonitoredTrainingSession(master="", is_chief=True, checkpoint_dir=str(tmp_path)) as sess:
        for _ in range(40):
            l, a, _ = sess.run([loss, acc, train_op], feed_dict={x: bx.numpy(), y: by.numpy(), K.learning_phase(): 1})
            got.append(l)
        assert sess.run(gs) == 40
        val = acc.eval({x: bx.numpy(), y: by.numpy(), K.learning_phase(): 0}, session=sess)
        assert 0.0 <= val <= 1.0 and abs(val - a) < 0.2
    assert got[-1] < got[0]
    assert max(abs(a - b) for a, b in zip(got, want)) < 2e-5, (got[-3:], want[-3:])
    # sessions, checkpoints and the graph dump work on it as on a chain
    ck = latest_checkpoint(str(tmp_path))
    assert ck is not None and ck.endswith("-40")
    assert {"dense_2/kernel", "dense_2/kernel/Adam_1", "global_step"} <= set(dict(list_variables(ck)))
    assert os.path.exists(os.path.join(str(tmp_path), "graph.pbtxt"))
    tf.reset_default_graph()


def test_two_inputs_two_summed_losses_match_pytorch_reference():
    """Two input placeholders, two branches joined by an add, two heads whose losses are added up."""
    import torch
    tf.reset_default_graph()
    xa = tf.placeholder(tf.float32, shape=(None, 12))
    xb = tf.placeholder(tf.float32, shape=(None, 6))
    ya = tf.placeholder(tf.float32, shape=(None, 5))        # one-hot classes
    yb = tf.placeholder(tf.float32, shape=(None, 3))
    trunk = Dense(24, activation="relu")(xa) + Dense(24, activation="relu")(xb)
    logits = Dense(5)(trunk)
    reg = Dense(3, activation="sigmoid")(trunk)
    loss = tf.reduce_mean(tf.nn.softmax_cross_entropy_with_logits(labels=ya, logits=logits)) \
        + tf.reduce_mean(keras.losses.mean_squared_error(yb, reg))
    train_op = tf.train.AdamOptimizer(learning_rate=0.003).minimize(loss)
    eng = tf.get_default_graph().engine()
    assert eng.dag and [p.name for p in eng.input_nodes] == [xa.name, xb.name] and len(eng.label_nodes) == 2
    w0 = {v.name: v.tensor.clone() for v in eng.model.store.trainable}
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(48, 12, generator=g), torch.randn(48, 6, generator=g)
    cls = torch.nn.functional.one_hot(torch.randint(0, 5, (48,), generator=g), 5).float()
    r = torch.rand(48, 3, generator=g)

    def ref_loss(w, a, b, cls, r):
        t = torch.relu(a @ w["dense_1/kernel"] + w["dense_1/bias"]) + torch.relu(b @ w["dense_2/kernel"] + w["dense_2/bias"])
        lg = t @ w["dense_3/kernel"] + w["dense_3/bias"]
        rg = torch.sigmoid(t @ w["dense_4/kernel"] + w["dense_4/bias"])
        return -(cls * torch.log_softmax(lg, dim=1)).sum(dim=1).mean() + ((rg - r) ** 2).mean()

    want = _torch_reference(w0, (a, b, cls, r), ref_loss, 30, lr=0.003)
    got = []
    with tf.Session() as sess:
        for _ in range(30):
            l, _ = sess.run([loss, train_op], feed_dict={xa: a.numpy(), xb: b.numpy(), ya: cls.numpy(), yb: r.numpy()})
            got.append(l)
    assert got[-1] < 0.8 * got[0]
    assert max(abs(x1 - x2) for x1, x2 in zip(got, want)) < 5e-5, (got[-3:], want[-3:])
    tf.reset_default_graph()


def test_non_layer_op_on_the_model_path():
    """``tf.square`` applied to a layer's output (neither a layer nor a matmul group) trains through the general path."""
    import torch
    tf.reset_default_graph()
    x = tf.placeholder(tf.float32, shape=(None, 8))
    y = tf.placeholder(tf.float32, shape=(None, 4))
    p = tf.square(Dense(4)(x))
    loss = tf.reduce_mean(tf.square(p - y))
    train_op = tf.train.GradientDescentOptimizer(0.05).minimize(loss)
    assert tf.get_default_graph().engine().dag
    g = torch.Generator().manual_seed(2)
    bx, by = torch.randn(32, 8, generator=g).numpy(), torch.rand(32, 4, generator=g).numpy()
    with tf.Session() as sess:
        first = sess.run([loss, train_op], feed_dict={x: bx, y: by})[0]
        for _ in range(50):
            last = sess.run([loss, train_op], feed_dict={x: bx, y: by})[0]
    assert last < 0.7 * first
    tf.reset_default_graph()


def test_tf_layers_dense_and_lone_activation():
    tf.reset_default_graph()
    x = tf.placeholder(tf.float32, shape=(None, 16))
    y = tf.placeholder(tf.float32, shape=(None, 4))
    h = tf.nn.relu(tf.layers.dense(x, 12))             # a lone activation after a layer -> Activation layer
    p = tf.layers.dense(h, 4, activation=tf.sigmoid)
    loss = tf.reduce_mean(tf.square(p - y))
    tf.train.AdamOptimizer().minimize(loss)
    eng = tf.get_default_graph().engine()
    assert [type(l).__name__ for l in eng.model.layers] == ["_Dense", "Activation", "_Dense"]
    tf.reset_default_graph()


def test_sync_replicas_optimizer_refuses_a_partial_aggregate():
    import torch.distributed as dist
    from distributed_tensorflow_amd.parallel import init_process_group
    from tests.test_sync_replicas_cpu import _free_port
    tf.reset_default_graph()
    x = tf.placeholder(tf.float32, shape=(None, 8))
    y = tf.placeholder(tf.float32, shape=(None, 4))
    loss = tf.reduce_mean(tf.square(Dense(4, activation="sigmoid")(x) - y))
    opt = tf.train.SyncReplicasOptimizer(tf.train.AdamOptimizer(), replicas_to_aggregate=3, total_num_replicas=4)
    opt.minimize(loss)
    init_process_group("gloo", rank=0, world_size=1, master_addr="127.0.0.1", master_port=_free_port())
    try:
        with pytest.raises(ValueError, match="replicas_to_aggregate=3"):
            tf.get_default_graph().engine()
    finally:
        dist.destroy_process_group()
        tf.reset_default_graph()


def test_tf1_deep_mnist_tutorial_graph_with_raw_ops():
    _deep_mnist_with_raw_ops()


@pytest.mark.gpu
def test_tf1_deep_mnist_tutorial_graph_with_raw_ops_gpu(gpu_device):
    assert _deep_mnist_with_raw_ops().type == "cuda"


def _deep_mnist_with_raw_ops():
    """The classic TF1 convolutional MNIST graph written with raw ops - ``tf.reshape``, ``tf.nn.conv2d`` +
    bias + ``tf.nn.relu``, ``tf.nn.max_pool``, hand-written ``tf.Variable``s, ``tf.matmul`` - trains through the graph
    model (the convolutions and poolings run on the same kernels as the Keras layers) and matches a PyTorch replica
    of the same graph step for step."""
    import torch
    from distributed_tensorflow_amd.data import mnist_like
    tf.reset_default_graph()
    x = tf.placeholder(tf.float32, [None, 784])
    y_ = tf.placeholder(tf.float32, [None, 10])
    x_image = tf.reshape(x, [-1, 28, 28, 1])
    W1 = tf.Variable(tf.truncated_normal([5, 5, 1, 8], stddev=0.1), name="W_conv1")
    b1 = tf.Variable(tf.constant(0.1, shape=[8]), name="b_conv1")
    h1 = tf.nn.relu(tf.nn.conv2d(x_image, W1, strides=[1, 1, 1, 1], padding="SAME") + b1)
    p1 = tf.nn.max_pool(h1, ksize=[1, 2, 2, 1], strides=[1, 2, 2, 1], padding="SAME")
    W2 = tf.Variable(tf.truncated_normal([3, 3, 8, 16], stddev=0.1), name="W_conv2")
    b2 = tf.Variable(tf.constant(0.1, shape=[16]), name="b_conv2")
    h2 = tf.nn.relu(tf.nn.conv2d(p1, W2, strides=[1, 1, 1, 1], padding="SAME") + b2)
    p2 = tf.nn.max_pool(h2, ksize=[1, 2, 2, 1], strides=[1, 2, 2, 1], padding="SAME")
    flat = tf.reshape(p2, [-1, 7 * 7 * 16])
    Wf = tf.Variable(tf.truncated_normal([7 * 7 * 16, 32], stddev=0.1), name="W_fc1")
    bf = tf.Variable(tf.constant(0.1, shape=[32]), name="b_fc1")
    hf = tf.nn.relu(tf.matmul(flat, Wf) + bf)
    keep_prob = tf.placeholder(tf.float32)                    # fed: 1.0 here, so that the PyTorch replica needs no mask
    hf = tf.nn.dropout(hf, keep_prob)
    Wo = tf.Variable(tf.truncated_normal([32, 10], stddev=0.1), name="W_fc2")
    bo = tf.Variable(tf.constant(0.1, shape=[10]), name="b_fc2")
    logits = tf.matmul(hf, Wo) + bo
    loss = tf.reduce_mean(tf.nn.softmax_cross_entropy_with_logits(labels=y_, logits=logits))
    train = tf.train.GradientDescentOptimizer(0.1).minimize(loss)
    acc = tf.reduce_mean(tf.cast(tf.equal(tf.argmax(logits, 1), tf.argmax(y_, 1)), tf.float32))

    xs, ys = mnist_like(96, seed=3)
    yo = torch.nn.functional.one_hot(ys, 10).float()
    got = []
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        eng = tf.get_default_graph().engine()
        assert eng.dag
        w = {v.name: v.tensor.detach().cpu().clone().double().requires_grad_(True) for v in eng.model.store.trainable}
        assert set(w) == {"W_conv1", "b_conv1", "W_conv2", "b_conv2", "W_fc1", "b_fc1", "W_fc2", "b_fc2"}
        for i in range(6):
            l, _ = sess.run([loss, train], {x: xs.numpy(), y_: yo.numpy(), keep_prob: 1.0})
            got.append(l)
        a = sess.run(acc, {x: xs.numpy(), y_: yo.numpy(), keep_prob: 1.0})
        dropped = sess.run(loss, {x: xs.numpy(), y_: yo.numpy(), keep_prob: 0.5})
        kept = sess.run(loss, {x: xs.numpy(), y_: yo.numpy(), keep_prob: 1.0})
        assert dropped != kept                                # the fed value is what is applied
    # the same graph in PyTorch (fp64), plain gradient descent
    want = []
    fn = torch.nn.functional
    for i in range(6):
        h = xs.double().reshape(-1, 1, 28, 28)
        h = fn.max_pool2d(torch.relu(fn.conv2d(h, w["W_conv1"].permute(3, 2, 0, 1), w["b_conv1"], padding=2)), 2)
        h = fn.max_pool2d(torch.relu(fn.conv2d(h, w["W_conv2"].permute(3, 2, 0, 1), w["b_conv2"], padding=1)), 2)
        h = h.permute(0, 2, 3, 1).reshape(-1, 7 * 7 * 16)
        h = torch.relu(h @ w["W_fc1"] + w["b_fc1"])
        lo = fn.cross_entropy(h @ w["W_fc2"] + w["b_fc2"], ys)
        want.append(lo.item())
        grads = torch.autograd.grad(lo, list(w.values()))
        with torch.no_grad():
            for t, g in zip(w.values(), grads):
                t -= 0.1 * g
    assert max(abs(p - q) for p, q in zip(got, want)) < 2e-4, (got, want)
    assert got[-1] < got[0] and 0.0 <= a <= 1.0
    tf.reset_default_graph()
    return eng.model.device


def test_supervisor_style_script(tmp_path):
    """The older way of writing the reference's session: ``tf.train.Supervisor`` + ``prepare_or_wait_for_session`` +
    ``sv.stop()``, with checkpoints in ``logdir`` and a resume from them."""
    data = get_data(500, seed=1)

    def run(last):
        tf.reset_default_graph()
        g = build_graph()
        sv = tf.train.Supervisor(is_chief=True, logdir=str(tmp_path), global_step=g["global_step"], save_model_secs=600)
        sess = sv.prepare_or_wait_for_session("")
        start = step = sess.run(g["global_step"])
        xt, yt = data[0], data[1]
        while not sv.should_stop() and step < last:
            i = step % 10
            _, step = sess.run([g["train"], g["global_step"]],
                               feed_dict={g["input"]: xt[i * 50:(i + 1) * 50], g["labels"]: yt[i * 50:(i + 1) * 50],
                                          K.learning_phase(): 1})
        sv.stop()
        return start, step

    assert run(30) == (0, 30)
    assert latest_checkpoint(str(tmp_path)).endswith("model.ckpt-30")
    assert run(45) == (30, 45)                    # a second run continues from the checkpoint, as with TF
    tf.reset_default_graph()
    g = build_graph()
    with tf.train.Supervisor(is_chief=True, logdir=None).managed_session("") as sess:      # the `with` form, no logdir
        assert sess.run(g["global_step"]) == 0
    tf.reset_default_graph()

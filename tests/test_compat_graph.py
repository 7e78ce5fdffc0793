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

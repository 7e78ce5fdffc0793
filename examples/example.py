"""Distributed training of the 32-bit XOR MLP, graph/session style.

Launch contract of /root/reference/example.py: no arguments = single-machine training; one
process per task with ``JOB_NAME`` / ``TASK_INDEX`` / ``PS_HOSTS`` / ``WORKER_HOSTS`` in the
environment (or the ``--job_name`` / ``--task_index`` / ``--ps_hosts`` / ``--worker_hosts``
flags) = distributed training::

    H="--ps_hosts=127.0.0.1:2222 --worker_hosts=127.0.0.1:2223,127.0.0.1:2224"
    python examples/example.py --job_name=ps     --task_index=0 $H &
    python examples/example.py --job_name=worker --task_index=0 $H &
    python examples/example.py --job_name=worker --task_index=1 $H

Workers train asynchronously against variables owned by the ps tasks (placed round-robin by
``replica_device_setter``) and share ``global_step``, so stopping at step 30000 is a
cluster-wide budget.  Extras the reference does not have: ``--sync`` (synchronous replicas),
``--max_staleness N`` (bounded asynchrony), ``--last_step``, ``--seed``; and the chief shuts
the ps tasks down when training is complete.
"""
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributed_tensorflow_amd import compat as tf                        # noqa: E402
from distributed_tensorflow_amd.compat import keras                        # noqa: E402
from distributed_tensorflow_amd.data import get_data                       # noqa: E402
from distributed_tensorflow_amd.paths import get_data_path, get_logs_path  # noqa: E402

K = keras.backend

# Hyper-parameters (module constants, as in the reference)
bits = 32
train_batch_size = 50
train_set_size = 30000
val_set_size = 1000
epochs = 50
print_rate = 5

LOCAL_LOGS = os.path.expanduser("~/distributed_tensorflow_amd/logs")
LOCAL_DATA = os.path.expanduser("~/distributed_tensorflow_amd/data")


def define_flags():
    """Cluster flags default to the launcher's environment; all four variables or none (single machine)."""
    env = os.environ
    have_env = all(k in env for k in ("JOB_NAME", "TASK_INDEX", "PS_HOSTS", "WORKER_HOSTS"))
    flags = tf.app.flags
    flags.DEFINE_string("job_name", env["JOB_NAME"] if have_env else None, "job name: worker or ps")
    flags.DEFINE_integer("task_index", env["TASK_INDEX"] if have_env else 0,
                         "Task index (>= 0).  Worker 0 is the chief: it initialises / restores the variables "
                         "and writes the checkpoints")
    flags.DEFINE_string("ps_hosts", env["PS_HOSTS"] if have_env else None, "Comma-separated host:port list")
    flags.DEFINE_string("worker_hosts", env["WORKER_HOSTS"] if have_env else None, "Comma-separated host:port list")
    flags.DEFINE_string("data_dir", get_data_path(dataset_name="", local_root=LOCAL_DATA, local_repo="", path=""),
                        "Dataset directory (unused: the task is synthetic)")
    flags.DEFINE_string("log_dir", get_logs_path(root=LOCAL_LOGS), "Directory for event files and checkpoints")
    flags.DEFINE_boolean("sync", False, "synchronous replicas instead of the asynchronous parameter server")
    flags.DEFINE_integer("max_staleness", -1, "asynchronous mode: bound on worker clock skew (-1 = unbounded)")
    flags.DEFINE_integer("last_step", int(epochs * (train_set_size / train_batch_size)), "global step to stop at")
    flags.DEFINE_float("save_checkpoint_secs", 600.0, "seconds between the chief's periodic checkpoints (TF default 600)")
    flags.DEFINE_integer("seed", -1, "data / initialisation seed (-1 = unseeded, like the reference)")
    return flags.FLAGS


def device_and_target(FLAGS):
    """(device function, session target).  A ps task serves inside this call and exits afterwards."""
    if FLAGS.job_name is None:
        print("Running single-machine training")
        return None, ""
    print("Running distributed training")
    for flag, value in (("task_index", FLAGS.task_index), ("ps_hosts", FLAGS.ps_hosts),
                        ("worker_hosts", FLAGS.worker_hosts)):
        if value is None or value == "":
            raise ValueError("Must specify an explicit `%s`" % flag)
    cluster = tf.train.ClusterSpec({"ps": FLAGS.ps_hosts.split(","), "worker": FLAGS.worker_hosts.split(",")})
    server = tf.train.Server(cluster, job_name=FLAGS.job_name, task_index=FLAGS.task_index)
    if FLAGS.job_name == "ps":
        server.join()
        sys.exit(0)
    setter = tf.train.replica_device_setter(worker_device="/job:worker/task:%d" % FLAGS.task_index, cluster=cluster)
    return setter, server.target


def build_graph(device, FLAGS):
    """64 -> 128 relu -> dropout .3 -> 128 relu -> dropout .3 -> 32 sigmoid, MSE loss, rounded-bit accuracy, Adam."""
    Dense, Dropout = keras.layers.Dense, keras.layers.Dropout
    g = SimpleNamespace()
    with tf.device(device):
        g.input = tf.placeholder(tf.float32, shape=(None, 2 * bits))
        h = Dropout(0.3)(Dense(128, activation="relu")(g.input))
        h = Dropout(0.3)(Dense(128, activation="relu")(h))
        g.preds = Dense(bits, activation="sigmoid")(h)
        g.labels = tf.placeholder(tf.float32, shape=(None, bits))
        with tf.name_scope("accuracy"):
            g.accuracy = tf.reduce_mean(tf.cast(tf.equal(tf.round(g.preds), tf.round(g.labels)), "float"))
            tf.summary.scalar("accuracy", g.accuracy)
        with tf.name_scope("loss"):
            g.loss = tf.reduce_mean(keras.losses.mean_squared_error(g.labels, g.preds))
            tf.summary.scalar("loss", g.loss)
        opt = tf.train.AdamOptimizer()
        if FLAGS.sync:
            opt = tf.train.SyncReplicasOptimizer(opt)
        g.global_step = tf.Variable(0, dtype=tf.int64, name="global_step", trainable=False)
        g.train_step = opt.minimize(g.loss, global_step=g.global_step)
        g.summaries = tf.summary.merge_all()
    graph = tf.get_default_graph()
    graph.max_staleness = None if FLAGS.max_staleness < 0 else FLAGS.max_staleness
    graph.seed = None if FLAGS.seed < 0 else FLAGS.seed + FLAGS.task_index
    return g


def main():
    FLAGS = define_flags()
    device, target = device_and_target(FLAGS)
    g = build_graph(device, FLAGS)
    writer = tf.summary.FileWriter(FLAGS.log_dir)

    # every worker draws its own data set, as in the reference
    data_seed = None if FLAGS.seed < 0 else FLAGS.seed + 7 * FLAGS.task_index
    x_train, y_train, x_val, y_val = get_data(train_set_size, bits=bits, val_set_size=val_set_size, seed=data_seed)
    batches_per_epoch = len(x_train) // train_batch_size

    hooks = [tf.train.StopAtStepHook(last_step=FLAGS.last_step)]
    with tf.train.MonitoredTrainingSession(master=target, is_chief=(FLAGS.task_index == 0),
                                           checkpoint_dir=FLAGS.log_dir, hooks=hooks,
                                           save_checkpoint_secs=FLAGS.save_checkpoint_secs) as sess:
        writer.add_graph(sess.graph)
        epoch = 0
        while not sess.should_stop():
            epoch += 1
            avg_loss = avg_accuracy = 0.0
            for i in range(batches_per_epoch):
                if sess.should_stop():
                    break
                lo, hi = i * train_batch_size, (i + 1) * train_batch_size
                acc, loss_value, summary, _ = sess.run(
                    [g.accuracy, g.loss, g.summaries, g.train_step],
                    feed_dict={g.input: x_train[lo:hi], g.labels: y_train[lo:hi], K.learning_phase(): 1})
                avg_loss += loss_value / batches_per_epoch
                avg_accuracy += acc / batches_per_epoch
                writer.add_summary(summary, epoch + float(i) / batches_per_epoch)
            if (epoch + 1) % print_rate == 0:
                val_acc = g.accuracy.eval({g.input: x_val, g.labels: y_val, K.learning_phase(): 0}, session=sess)
                print("Epoch:", (epoch + 1), "Loss =", "{:.5f}".format(avg_loss), "  Training Accuracy:", avg_accuracy,
                      "  Validation Accuracy:", val_acc)
        print("\nTraining complete!")
    writer.close()

    if FLAGS.job_name is not None:
        strategy = tf.get_default_graph().engine().model.strategy
        if hasattr(strategy, "worker_done"):
            strategy.worker_done()
        if FLAGS.task_index == 0:
            if hasattr(strategy, "shutdown_owners"):
                strategy.shutdown_owners()   # waits for all workers, then ends server.join() in the ps tasks
            else:
                tf.train.Server.current().shutdown_parameter_servers()   # synchronous replicas leave the ps idle


if __name__ == "__main__":
    main()

"""Template: distributed training in the graph/session style.  Copy it, then fill the marked blocks.

The reference ships this file empty (/root/reference/outline_tensorflow.py); its example.py was derived from
such a template ("#TODO define your graph here", /root/reference/example.py:147).  The template below is complete
and runnable as it stands (it trains an MNIST-shaped softmax classifier on synthetic data); the blocks between
``# >>>`` and ``# <<<`` are the ones to replace.

Launch: no arguments = single machine; JOB_NAME / TASK_INDEX / PS_HOSTS / WORKER_HOSTS (or the equally named
flags) = one task of a ps/worker cluster.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributed_tensorflow_amd import compat as tf              # noqa: E402
from distributed_tensorflow_amd.compat import keras              # noqa: E402
from distributed_tensorflow_amd.cluster import device_and_target  # noqa: E402
from distributed_tensorflow_amd.paths import get_logs_path       # noqa: E402

flags = tf.app.flags
env = os.environ
have_env = all(k in env for k in ("JOB_NAME", "TASK_INDEX", "PS_HOSTS", "WORKER_HOSTS"))
flags.DEFINE_string("job_name", env["JOB_NAME"] if have_env else None, "job name: worker or ps")
flags.DEFINE_integer("task_index", env["TASK_INDEX"] if have_env else 0, "task index; worker 0 is the chief")
flags.DEFINE_string("ps_hosts", env["PS_HOSTS"] if have_env else None, "Comma-separated host:port list")
flags.DEFINE_string("worker_hosts", env["WORKER_HOSTS"] if have_env else None, "Comma-separated host:port list")
flags.DEFINE_string("log_dir", get_logs_path(root="./logs"), "event files and checkpoints")
flags.DEFINE_integer("last_step", 300, "global step to stop at")
FLAGS = flags.FLAGS

# ps tasks serve inside this call and exit; workers get (device function, session target, server)
device, target, server = device_and_target(FLAGS.job_name, FLAGS.task_index, FLAGS.ps_hosts, FLAGS.worker_hosts)

with tf.device(device):
    # >>> define your graph here: placeholders, layers, loss, metric ------------------------------------
    inputs = tf.placeholder(tf.float32, shape=(None, 784))
    labels = tf.placeholder(tf.float32, shape=(None, 10))
    hidden = keras.layers.Dense(128, activation="relu")(inputs)
    logits = keras.layers.Dense(10)(hidden)
    loss = tf.reduce_mean(tf.nn.softmax_cross_entropy_with_logits(labels=labels, logits=logits))
    accuracy = tf.reduce_mean(tf.cast(tf.equal(tf.argmax(logits, 1), tf.argmax(labels, 1)), "float"))
    tf.summary.scalar("loss", loss)
    tf.summary.scalar("accuracy", accuracy)
    # <<< ------------------------------------------------------------------------------------------------
    global_step = tf.Variable(0, dtype=tf.int64, name="global_step", trainable=False)
    train_op = tf.train.AdamOptimizer(1e-3).minimize(loss, global_step=global_step)
    summaries = tf.summary.merge_all()

# >>> load your data here (every worker reads its own shard) ------------------------------------------------
from distributed_tensorflow_amd.data import mnist_like  # noqa: E402

x_train, y_train = mnist_like(6000, seed=FLAGS.task_index, one_hot=True)
batch_size = 100
# <<< --------------------------------------------------------------------------------------------------------

hooks = [tf.train.StopAtStepHook(last_step=FLAGS.last_step)]
with tf.train.MonitoredTrainingSession(master=target, is_chief=(FLAGS.task_index == 0), checkpoint_dir=FLAGS.log_dir,
                                       hooks=hooks) as sess:
    step = 0
    while not sess.should_stop():
        lo = (step * batch_size) % (len(x_train) - batch_size + 1)
        acc, l, _ = sess.run([accuracy, loss, train_op], feed_dict={inputs: x_train[lo:lo + batch_size],
                                                                   labels: y_train[lo:lo + batch_size]})
        if step % 100 == 0:
            print("local step %d  loss %.4f  accuracy %.3f" % (step, l, acc))
        step += 1
    print("\nTraining complete!")

if server is not None:
    strategy = tf.get_default_graph().engine().model.strategy
    if hasattr(strategy, "worker_done"):
        strategy.worker_done()
    if FLAGS.task_index == 0:
        strategy.shutdown_owners() if hasattr(strategy, "shutdown_owners") else server.shutdown_parameter_servers()

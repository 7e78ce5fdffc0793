"""Between-graph replicated training written the pre-``MonitoredTrainingSession`` way: ``tf.train.Supervisor`` +
``prepare_or_wait_for_session`` + a hand-written ``while step < train_steps`` loop over raw ``tf.Variable`` /
``tf.matmul`` ops (the other common shape of a TF1 ps/worker script; /root/reference/example.py:124-141,187-228 is
the ``MonitoredTrainingSession`` shape of the same thing)::

    H="--ps_hosts=127.0.0.1:2222 --worker_hosts=127.0.0.1:2223,127.0.0.1:2224"
    python examples/supervisor_replica.py --job_name=ps     --task_index=0 $H &
    python examples/supervisor_replica.py --job_name=worker --task_index=0 $H &
    python examples/supervisor_replica.py --job_name=worker --task_index=1 $H

Workers push gradients to / read variables from the ps tasks asynchronously and share ``global_step``;
``--train_steps`` is the cluster-wide budget; worker 0 is the chief (initialises or restores, checkpoints to
``--train_dir``, ends the ps tasks when every worker is done).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributed_tensorflow_amd import compat as tf        # noqa: E402
from distributed_tensorflow_amd.data import mnist_like      # noqa: E402

flags = tf.app.flags
env = os.environ            # `python -m distributed_tensorflow_amd.launch` describes the task in the environment
flags.DEFINE_string("job_name", env.get("JOB_NAME"), "worker or ps")
flags.DEFINE_integer("task_index", int(env.get("TASK_INDEX", 0)), "index of the task within its job; worker 0 is the chief")
flags.DEFINE_string("ps_hosts", env.get("PS_HOSTS", "127.0.0.1:2222"), "comma-separated host:port list")
flags.DEFINE_string("worker_hosts", env.get("WORKER_HOSTS", "127.0.0.1:2223,127.0.0.1:2224"), "comma-separated host:port list")
flags.DEFINE_string("train_dir", "/tmp/supervisor_replica", "checkpoint directory")
flags.DEFINE_integer("hidden_units", 100, "units of the hidden layer")
flags.DEFINE_integer("train_steps", 200, "global steps to train for (all workers together)")
flags.DEFINE_integer("batch_size", 100, "examples per step and worker")
flags.DEFINE_float("learning_rate", 0.01, "Adam step size")
flags.DEFINE_integer("seed", 0, "data / initialisation seed")
FLAGS = flags.FLAGS

IMAGE_PIXELS, CLASSES = 28, 10


def main():
    if FLAGS.job_name not in ("ps", "worker"):
        raise ValueError("--job_name must be ps or worker")
    cluster = tf.train.ClusterSpec({"ps": FLAGS.ps_hosts.split(","), "worker": FLAGS.worker_hosts.split(",")})
    server = tf.train.Server(cluster, job_name=FLAGS.job_name, task_index=FLAGS.task_index)
    if FLAGS.job_name == "ps":
        server.join()
        return
    is_chief = FLAGS.task_index == 0
    with tf.device(tf.train.replica_device_setter(worker_device="/job:worker/task:%d" % FLAGS.task_index, cluster=cluster)):
        global_step = tf.Variable(0, name="global_step", trainable=False, dtype=tf.int64)
        hid_w = tf.Variable(tf.truncated_normal([IMAGE_PIXELS * IMAGE_PIXELS, FLAGS.hidden_units],
                                                stddev=1.0 / IMAGE_PIXELS), name="hid_w")
        hid_b = tf.Variable(tf.zeros([FLAGS.hidden_units]), name="hid_b")
        sm_w = tf.Variable(tf.truncated_normal([FLAGS.hidden_units, CLASSES], stddev=1.0 / FLAGS.hidden_units ** 0.5),
                           name="sm_w")
        sm_b = tf.Variable(tf.zeros([CLASSES]), name="sm_b")
        x = tf.placeholder(tf.float32, [None, IMAGE_PIXELS * IMAGE_PIXELS])
        y_ = tf.placeholder(tf.float32, [None, CLASSES])
        hid = tf.nn.relu(tf.matmul(x, hid_w) + hid_b)
        logits = tf.matmul(hid, sm_w) + sm_b
        cross_entropy = tf.reduce_mean(tf.nn.softmax_cross_entropy_with_logits(labels=y_, logits=logits))
        train_step = tf.train.AdamOptimizer(FLAGS.learning_rate).minimize(cross_entropy, global_step=global_step)
        init_op = tf.global_variables_initializer()
    tf.get_default_graph().seed = FLAGS.seed + FLAGS.task_index

    sv = tf.train.Supervisor(is_chief=is_chief, logdir=FLAGS.train_dir, init_op=init_op, recovery_wait_secs=1,
                             global_step=global_step)
    print("Worker %d: %s session" % (FLAGS.task_index, "initialising" if is_chief else "waiting for the"))
    sess = sv.prepare_or_wait_for_session(server.target)
    print("Worker %d: session ready" % FLAGS.task_index)

    import torch
    images, labels = mnist_like(FLAGS.batch_size * 50, seed=FLAGS.seed + 7 * FLAGS.task_index)
    onehot = torch.nn.functional.one_hot(labels, CLASSES).float()
    started = time.time()
    local_step, step, loss = 0, 0, float("nan")
    while not sv.should_stop():
        lo = (local_step % 50) * FLAGS.batch_size
        _, step, loss = sess.run([train_step, global_step, cross_entropy],
                                 feed_dict={x: images[lo:lo + FLAGS.batch_size], y_: onehot[lo:lo + FLAGS.batch_size]})
        local_step += 1
        if local_step % 50 == 0:
            print("Worker %d: local step %d, global step %d, loss %.4f" % (FLAGS.task_index, local_step, step, loss))
        if step >= FLAGS.train_steps:
            break
    print("Worker %d: done after %d local steps (global step %d, %.1f s), cross entropy %.4f"
          % (FLAGS.task_index, local_step, step, time.time() - started, loss))
    sv.stop()

    strategy = tf.get_default_graph().engine().model.strategy
    if hasattr(strategy, "worker_done"):
        strategy.worker_done()
    if is_chief:
        if hasattr(strategy, "shutdown_owners"):
            strategy.shutdown_owners()
        else:
            tf.train.Server.current().shutdown_parameter_servers()


if __name__ == "__main__":
    main()

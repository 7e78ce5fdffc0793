"""Keras-style variant: ``Sequential`` -> ``compile`` -> ``fit`` with validation data and a TensorBoard
callback, on the same cluster runtime and launch contract as ``example.py``
(the workflow of /root/reference/example2.py:151-165,189-200).

    python examples/example2.py                         # single machine
    JOB_NAME=worker TASK_INDEX=0 PS_HOSTS=... WORKER_HOSTS=... python examples/example2.py   # one task of a cluster

In a cluster the workers train as synchronous replicas by default (``--mode sync``: gradients
are reduce-scattered, the sharded Adam update runs on each replica's shard, parameters are
all-gathered); ``--mode async`` trains against the parameter-server tasks instead.
"""
import os
import sys
from time import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributed_tensorflow_amd import compat as tf                        # noqa: E402
from distributed_tensorflow_amd.callbacks import ClusterHealth, TensorBoard  # noqa: E402
from distributed_tensorflow_amd.data import get_data                       # noqa: E402
from distributed_tensorflow_amd.engine import ModelEngine                  # noqa: E402
from distributed_tensorflow_amd.layers import Dense, Dropout              # noqa: E402
from distributed_tensorflow_amd.models import Sequential                   # noqa: E402
from distributed_tensorflow_amd.paths import get_logs_path                 # noqa: E402
from distributed_tensorflow_amd.session import MonitoredTrainingSession    # noqa: E402

bits = 32
train_set_size = 30000
val_set_size = 1000

flags = tf.app.flags
env = os.environ
_have_env = all(k in env for k in ("JOB_NAME", "TASK_INDEX", "PS_HOSTS", "WORKER_HOSTS"))
flags.DEFINE_string("job_name", env["JOB_NAME"] if _have_env else None, "job name: worker or ps")
flags.DEFINE_integer("task_index", env["TASK_INDEX"] if _have_env else 0, "task index; worker 0 is the chief")
flags.DEFINE_string("ps_hosts", env["PS_HOSTS"] if _have_env else None, "Comma-separated host:port list")
flags.DEFINE_string("worker_hosts", env["WORKER_HOSTS"] if _have_env else None, "Comma-separated host:port list")
flags.DEFINE_string("log_dir", get_logs_path(root=os.path.expanduser("~/distributed_tensorflow_amd/logs")),
                    "Directory for event files")
flags.DEFINE_string("mode", "sync", "cluster training mode: sync | async")
flags.DEFINE_integer("epochs", 20, "epochs to fit")
flags.DEFINE_integer("batch_size", 50, "mini-batch size")
flags.DEFINE_integer("steps_per_graph", 0, "replay this many steps per captured hipGraph (0 = eager launches)")
flags.DEFINE_integer("seed", -1, "data / initialisation seed (-1 = unseeded)")
FLAGS = flags.FLAGS


def make_strategy():
    """(strategy, session target): (None, "") on a single machine; otherwise the strategy of this worker task and its
    server's target (a ps task serves and exits)."""
    if FLAGS.job_name is None:
        print("Running single-machine training")
        return None, ""
    print("Running distributed training")
    cluster = tf.train.ClusterSpec({"ps": FLAGS.ps_hosts.split(","), "worker": FLAGS.worker_hosts.split(",")})
    server = tf.train.Server(cluster, job_name=FLAGS.job_name, task_index=FLAGS.task_index)
    if FLAGS.job_name == "ps":
        server.join()
        sys.exit(0)
    if FLAGS.mode == "async":
        from distributed_tensorflow_amd.parallel.async_ps import AsyncPSStrategy
        return AsyncPSStrategy(server), server.target
    from distributed_tensorflow_amd.parallel import Communicator, SyncReplicasStrategy, init_process_group
    host, port = server.rendezvous_address()
    init_process_group(None, rank=server.task_index, world_size=server.num_workers, master_addr=host, master_port=port)
    return SyncReplicasStrategy(Communicator()), server.target


def main():
    strategy, target = make_strategy()
    seed = None if FLAGS.seed < 0 else FLAGS.seed

    model = Sequential()
    model.add(Dense(128, input_dim=2 * bits, activation="relu"))
    model.add(Dropout(0.3))
    model.add(Dense(128, activation="relu"))
    model.add(Dropout(0.3))
    model.add(Dense(bits, activation="sigmoid"))
    # 'accuracy' resolves like Keras (argmax match for a non-crossentropy loss with several outputs);
    # 'binary_rounded_accuracy' is the bit-wise metric of example.py
    model.compile(loss="mean_squared_error", optimizer="adam", metrics=["accuracy", "binary_rounded_accuracy"],
                  strategy=strategy, seed=seed)

    x_train, y_train, x_val, y_val = get_data(train_set_size, bits=bits, val_set_size=val_set_size,
                                              seed=None if seed is None else seed + 7 * FLAGS.task_index,
                                              device=model.device)

    is_chief = FLAGS.task_index == 0
    # master=target: in a cluster the session then watches the other tasks' heartbeats, so a lost peer surfaces as
    # ClusterLostError instead of a hang (/root/reference/example2.py:189-190 passes master=target as well)
    with MonitoredTrainingSession(master=target, is_chief=is_chief, engine=ModelEngine(model)) as sess:
        callbacks = [TensorBoard(log_dir=FLAGS.log_dir.format(time()))] if is_chief else []
        if FLAGS.job_name is not None:
            callbacks.append(ClusterHealth(sess))
        history = model.fit(x_train, y_train, epochs=FLAGS.epochs, batch_size=FLAGS.batch_size,
                            validation_data=(x_val, y_val), callbacks=callbacks, verbose=1 if is_chief else 0,
                            steps_per_graph=FLAGS.steps_per_graph)
    if is_chief:
        last = {k: v[-1] for k, v in history.history.items()}
        print("final:", ", ".join("%s=%.4f" % kv for kv in sorted(last.items())))

    if FLAGS.job_name is not None:
        if hasattr(strategy, "worker_done"):
            strategy.worker_done()
        if is_chief:
            if hasattr(strategy, "shutdown_owners"):
                strategy.shutdown_owners()
            else:
                tf.train.Server.current().shutdown_parameter_servers()


if __name__ == "__main__":
    main()

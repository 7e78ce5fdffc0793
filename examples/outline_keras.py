"""Template: Keras-style ``Sequential`` / ``compile`` / ``fit`` training, single GPU or synchronous replicas.

The reference ships this file empty (/root/reference/outline_keras.py).  This template is complete and runnable as
it stands: an MNIST-shaped MLP trained with ``model.fit`` on synthetic data.  Replace the blocks between ``# >>>``
and ``# <<<``.

    python examples/outline_keras.py                                      # one GPU (or CPU)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/outline_keras.py   # 8 replicas
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributed_tensorflow_amd.callbacks import TensorBoard  # noqa: E402
from distributed_tensorflow_amd.layers import Dense, Dropout  # noqa: E402
from distributed_tensorflow_amd.models import Sequential       # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
strategy = None
if world > 1:
    # one process per GPU; gradients are reduce-scattered, the Adam update is sharded, parameters all-gathered
    from distributed_tensorflow_amd.parallel import Communicator, SyncReplicasStrategy, init_process_group
    init_process_group()
    strategy = SyncReplicasStrategy(Communicator())

# >>> define your model here ----------------------------------------------------------------------------------
model = Sequential()
model.add(Dense(128, input_dim=784, activation="relu"))
model.add(Dropout(0.3))
model.add(Dense(128, activation="relu"))
model.add(Dropout(0.3))
model.add(Dense(10, activation="softmax"))
model.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], strategy=strategy,
              seed=0)
# <<< ----------------------------------------------------------------------------------------------------------

# >>> load your data here (each replica its own shard) ---------------------------------------------------------
from distributed_tensorflow_amd.data import mnist_like  # noqa: E402

x_train, y_train = mnist_like(20000, seed=1 + rank, device=model.device)
x_val, y_val = mnist_like(2000, seed=99, device=model.device)
# <<< ----------------------------------------------------------------------------------------------------------

callbacks = [TensorBoard(log_dir=os.environ.get("DTF_LOGS_ROOT", "./logs"))] if rank == 0 else []
history = model.fit(x_train, y_train, epochs=int(os.environ.get("EPOCHS", "5")), batch_size=100,
                    validation_data=(x_val, y_val), callbacks=callbacks, verbose=1 if rank == 0 else 0)
if rank == 0:
    print("val_acc per epoch:", ["%.3f" % a for a in history.history["val_acc"]])

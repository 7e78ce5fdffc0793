"""MI355X-native distributed training framework with the capabilities of the
``distributed_tensorflow`` example repository: ps/worker cluster definition, between-graph
replicated async parameter-server and sync-replica training, a monitored training session
with hooks, a Keras-like ``Sequential``/``compile``/``fit`` API, TensorBoard event files and
TensorFlow-compatible (TensorBundle) checkpoints."""
__version__ = "0.1.0"

from . import callbacks, data, layers, losses, models, optimizers  # noqa: F401
from .layers import (Activation, AveragePooling2D, BatchNormalization, Conv2D, Dense, Dropout, Flatten,  # noqa: F401
                     GlobalAveragePooling2D, MaxPooling2D)
from .models import Sequential  # noqa: F401

"""Ready-made models for the benchmark configurations.

* ``xor_mlp``   - the network of /root/reference/example.py:149-154 and example2.py:151-156
                  (64 -> 128 relu -> dropout .3 -> 128 relu -> dropout .3 -> 32 sigmoid).
* ``mnist_mlp`` - the same body on MNIST-shaped data (784 inputs, 10-way softmax head).
* ``mnist_cnn`` - conv2d + dense classifier on 28x28x1 inputs.
* ``wide_mlp``  - widened MLP whose variables are meant to be sharded across GPUs
                  (>= 1e9 parameters at the default width/depth).
"""
from __future__ import annotations

from ..layers import Dense, Dropout
from .sequential import Sequential


def xor_mlp(bits: int = 32, hidden: int = 128, dropout: float = 0.3) -> Sequential:
    m = Sequential(name="xor_mlp")
    m.add(Dense(hidden, input_dim=2 * bits, activation="relu"))
    m.add(Dropout(dropout))
    m.add(Dense(hidden, activation="relu"))
    m.add(Dropout(dropout))
    m.add(Dense(bits, activation="sigmoid"))
    return m


def mnist_mlp(hidden: int = 128, dropout: float = 0.3, num_classes: int = 10, input_dim: int = 784) -> Sequential:
    m = Sequential(name="mnist_mlp")
    m.add(Dense(hidden, input_dim=input_dim, activation="relu"))
    m.add(Dropout(dropout))
    m.add(Dense(hidden, activation="relu"))
    m.add(Dropout(dropout))
    m.add(Dense(num_classes, activation="softmax"))
    return m


def wide_mlp(width: int = 16384, depth: int = 5, num_classes: int = 10, input_dim: int = 784,
             dropout: float = 0.0) -> Sequential:
    """``depth`` hidden layers of ``width`` units: 784*w + (depth-1)*w^2 + 10*w parameters
    (1.087e9 at the defaults)."""
    m = Sequential(name="wide_mlp")
    m.add(Dense(width, input_dim=input_dim, activation="relu"))
    for _ in range(depth - 1):
        if dropout > 0:
            m.add(Dropout(dropout))
        m.add(Dense(width, activation="relu"))
    m.add(Dense(num_classes, activation="softmax"))
    return m


def mnist_cnn(num_classes: int = 10, dropout: float = 0.3) -> Sequential:
    from ..layers import Conv2D, Flatten, MaxPooling2D
    m = Sequential(name="mnist_cnn")
    m.add(Conv2D(32, (5, 5), padding="same", activation="relu", input_shape=(28, 28, 1)))
    m.add(MaxPooling2D((2, 2)))
    m.add(Conv2D(64, (5, 5), padding="same", activation="relu"))
    m.add(MaxPooling2D((2, 2)))
    m.add(Flatten())
    m.add(Dense(512, activation="relu"))
    m.add(Dropout(dropout))
    m.add(Dense(num_classes, activation="softmax"))
    return m

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
    (1.087e9 at the defaults).  The ReLU layers use He-uniform initial values (the activation scale then does not
    depend on depth); train it with ``wide_mlp_learning_rate(width)``."""
    m = Sequential(name="wide_mlp")
    m.add(Dense(width, input_dim=input_dim, activation="relu", kernel_initializer="he_uniform"))
    for _ in range(depth - 1):
        if dropout > 0:
            m.add(Dropout(dropout))
        m.add(Dense(width, activation="relu", kernel_initializer="he_uniform"))
    m.add(Dense(num_classes, activation="softmax"))
    return m


def wide_mlp_learning_rate(width: int, base_lr: float = 1e-3, base_width: int = 128) -> float:
    """Adam's step per weight is about the learning rate whatever the gradient's size, and a unit with ``width``
    inputs adds up ``width`` such steps: the rate that works at the reference's width 128
    (/root/reference/example.py:150-154,168) has to shrink in proportion for the pre-activations of a wide layer to
    move as they do there (1e-3 at 16384 inputs changes them by ~16 per step and the loss diverges)."""
    return base_lr * min(1.0, float(base_width) / float(width))


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

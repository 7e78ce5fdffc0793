"""Keras' own convolutional MNIST example, as written for Keras 2 (Conv2D 32 / 64 with 3x3 windows, max pooling,
dropout on the feature maps, a 128-unit dense layer, Adadelta, categorical cross-entropy), on MNIST-shaped synthetic
data - the Keras-style counterpart of /root/reference/example2.py for a convolutional network::

    python examples/keras_mnist_cnn.py --epochs 3 --train_size 20000
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch                                                         # noqa: E402

from distributed_tensorflow_amd.compat import keras                  # noqa: E402
from distributed_tensorflow_amd.data import mnist_like               # noqa: E402

Sequential = keras.models.Sequential
Dense, Dropout, Flatten = keras.layers.Dense, keras.layers.Dropout, keras.layers.Flatten
Conv2D, MaxPooling2D = keras.layers.Conv2D, keras.layers.MaxPooling2D


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=128)
    ap.add_argument("--epochs", type=int, default=12)
    ap.add_argument("--train_size", type=int, default=60000)
    ap.add_argument("--test_size", type=int, default=10000)
    ap.add_argument("--log_dir", default="")
    args = ap.parse_args()
    num_classes = 10
    x_train, y_train = mnist_like(args.train_size, seed=1)
    x_test, y_test = mnist_like(args.test_size, seed=2)
    x_train, x_test = x_train.view(-1, 28, 28, 1), x_test.view(-1, 28, 28, 1)
    y_train = torch.nn.functional.one_hot(y_train, num_classes).float()
    y_test = torch.nn.functional.one_hot(y_test, num_classes).float()

    model = Sequential()
    model.add(Conv2D(32, kernel_size=(3, 3), activation="relu", input_shape=(28, 28, 1)))
    model.add(Conv2D(64, (3, 3), activation="relu"))
    model.add(MaxPooling2D(pool_size=(2, 2)))
    model.add(Dropout(0.25))
    model.add(Flatten())
    model.add(Dense(128, activation="relu"))
    model.add(Dropout(0.5))
    model.add(Dense(num_classes, activation="softmax"))
    model.compile(loss="categorical_crossentropy", optimizer=keras.optimizers.Adadelta(), metrics=["accuracy"])

    callbacks = [keras.callbacks.TensorBoard(log_dir=args.log_dir)] if args.log_dir else []
    model.fit(x_train, y_train, batch_size=args.batch_size, epochs=args.epochs, verbose=1,
              validation_data=(x_test, y_test), callbacks=callbacks)
    score = model.evaluate(x_test, y_test, batch_size=1000)
    print("Test loss:", score[0])
    print("Test accuracy:", score[1])


if __name__ == "__main__":
    main()

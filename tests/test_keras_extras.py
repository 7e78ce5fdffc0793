"""Keras conveniences around Sequential: get/set/save/load weights (TensorBundle files) and the ModelCheckpoint /
EarlyStopping / CSVLogger callbacks.  The reference drives checkpoints through MonitoredTrainingSession
(/root/reference/example.py:189-192) and uses only the TensorBoard callback (/root/reference/example2.py:197-200);
these are the neighbouring Keras entry points a user of `model.fit` reaches for."""
import os

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd import callbacks as cb
from distributed_tensorflow_amd.data import get_data
from distributed_tensorflow_amd.models import xor_mlp
from distributed_tensorflow_amd.saver import BundleReader


def _model(seed=0):
    m = xor_mlp(dropout=0.0)
    m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device="cpu", seed=seed)
    return m


def test_get_set_weights_roundtrip():
    a, b = _model(1), _model(2)
    wa = a.get_weights()
    assert [w.shape for w in wa] == [(64, 128), (128,), (128, 128), (128,), (128, 32), (32,)]
    assert not np.allclose(wa[0], b.get_weights()[0])
    b.set_weights(wa)
    torch.testing.assert_close(a.store.flat, b.store.flat)
    with pytest.raises(ValueError):
        b.set_weights(wa[:-1])
    with pytest.raises(ValueError):
        b.set_weights([w.T for w in wa])


def test_save_and_load_weights_keep_variables_step_and_adam_state(tmp_path):
    xt, yt, xv, yv = get_data(600, val_set_size=100, seed=3)
    a = _model(1)
    a.fit(xt, yt, batch_size=50, epochs=2, verbose=0)
    prefix = a.save_weights(str(tmp_path / "w" / "model"))
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    with BundleReader(prefix) as r:
        names = set(r.get_variable_to_shape_map())
    assert {"dense_1/kernel", "dense_3/bias", "global_step", "dense_1/kernel/Adam", "dense_1/kernel/Adam_1",
            "beta1_power"} <= names
    b = _model(9)
    step = b.load_weights(prefix, strict=True)
    assert step == a.global_step == 2 * (xt.shape[0] // 50)
    torch.testing.assert_close(a.store.flat, b.store.flat)
    torch.testing.assert_close(a.store.slot("m"), b.store.slot("m"))
    # training continues identically from the restored state
    a.fit(xt, yt, batch_size=50, epochs=1, shuffle=False, verbose=0)
    b.fit(xt, yt, batch_size=50, epochs=1, shuffle=False, verbose=0)
    torch.testing.assert_close(a.store.flat, b.store.flat)
    # weights only: a fresh optimizer state is fine for a non-strict load
    p2 = a.save_weights(str(tmp_path / "weights_only"), include_optimizer=False)
    c = _model(5)
    c.load_weights(p2)
    torch.testing.assert_close(a.store.flat, c.store.flat)
    with pytest.raises(KeyError):
        _model(6).load_weights(p2, strict=True)


def test_model_checkpoint_early_stopping_and_csv_logger(tmp_path):
    xt, yt, xv, yv = get_data(1000, val_set_size=200, seed=4)
    m = _model(2)
    ck = cb.ModelCheckpoint(str(tmp_path / "ck" / "w.{epoch:02d}"), monitor="val_loss", save_best_only=True)
    csv = cb.CSVLogger(str(tmp_path / "log.csv"))
    h = m.fit(xt, yt, batch_size=50, epochs=4, validation_data=(xv, yv), callbacks=[ck, csv], verbose=0)
    assert ck.saved and all(os.path.exists(p + ".index") for p in ck.saved)
    best_epoch = int(np.argmin(h.history["val_loss"])) + 1
    assert ck.saved[-1].endswith("w.%02d" % best_epoch)
    rows = open(tmp_path / "log.csv").read().strip().splitlines()
    assert rows[0].split(",")[0] == "epoch" and "val_loss" in rows[0] and len(rows) == 5

    # a monitor that cannot improve (min_delta larger than any change) stops after `patience` more epochs
    m2 = _model(3)
    es = cb.EarlyStopping(monitor="val_loss", min_delta=10.0, patience=1, restore_best_weights=True)
    snap = {}
    grab = cb.LambdaCallback(on_epoch_end=lambda e, logs: snap.setdefault(e, m2.store.flat.clone()))
    h2 = m2.fit(xt, yt, batch_size=50, epochs=10, validation_data=(xv, yv), callbacks=[grab, es], verbose=0)
    assert len(h2.history["loss"]) == 3 and es.stopped_epoch == 2        # best = epoch 0, waited through 1 and 2
    torch.testing.assert_close(m2.store.flat, snap[0])                   # best epoch's variables are back


def test_learning_rate_callbacks(tmp_path):
    xt, yt, xv, yv = get_data(500, val_set_size=100, seed=5)
    # a schedule equals training the epochs by hand with the same rates
    a, b = _model(4), _model(4)
    rates = [1e-3, 5e-4, 2.5e-4]
    h = a.fit(xt, yt, batch_size=50, epochs=3, shuffle=False, verbose=0,
              callbacks=[cb.LearningRateScheduler(lambda epoch, lr: rates[epoch])])
    assert h.history["lr"] == rates
    for e, lr in enumerate(rates):
        b.set_learning_rate(lr)
        b.fit(xt, yt, batch_size=50, epochs=e + 1, initial_epoch=e, shuffle=False, verbose=0)
    torch.testing.assert_close(a.store.flat, b.store.flat)

    m = _model(6)
    rl = cb.ReduceLROnPlateau(monitor="val_loss", factor=0.5, patience=1, min_delta=10.0, min_lr=2e-4)
    m.fit(xt, yt, batch_size=50, epochs=5, validation_data=(xv, yv), callbacks=[rl], verbose=0)
    assert m.optimizer.lr == pytest.approx(2e-4)           # halved after epochs 1, 2 and 3, then held at the floor
    with pytest.raises(ValueError):
        cb.ReduceLROnPlateau(factor=1.5)


def test_model_save_and_load_model_rebuild_architecture_and_state(tmp_path):
    from distributed_tensorflow_amd.models import load_model, mnist_cnn, model_from_json
    xt, yt, _, _ = get_data(300, val_set_size=50, seed=6)
    a = _model(7)
    a.fit(xt, yt, batch_size=50, epochs=1, verbose=0)
    path = a.save(str(tmp_path / "saved" / "xor"))
    assert os.path.exists(path + ".json") and os.path.exists(path + ".index")
    b = load_model(path, compile_args=dict(loss="mean_squared_error", optimizer="adam", device="cpu"))
    assert [type(l).__name__ for l in b.layers] == [type(l).__name__ for l in a.layers]
    assert b.global_step == a.global_step
    torch.testing.assert_close(a.store.flat, b.store.flat)
    torch.testing.assert_close(a.predict(xt[:20]), b.predict(xt[:20]))
    c = load_model(path)                              # not compiled: variables only
    torch.testing.assert_close(a.store.flat[:a.store.num_params], c.store.flat[:a.store.num_params])
    # every layer type survives the JSON round trip
    cnn = mnist_cnn()
    twin = model_from_json(cnn.to_json())
    assert twin.get_config()["layers"] == cnn.get_config()["layers"]
    twin.build(device="cpu")
    assert twin.store.num_params == 5 * 5 * 32 + 32 + 5 * 5 * 32 * 64 + 64 + 3136 * 512 + 512 + 5120 + 10


def test_fit_validation_split_holds_out_the_tail():
    import torch
    from distributed_tensorflow_amd.data import get_data
    from distributed_tensorflow_amd.models import xor_mlp
    x, y, _, _ = get_data(1000, val_set_size=0, seed=2)
    seen = {}

    from distributed_tensorflow_amd.callbacks import Callback

    class Spy(Callback):
        def on_train_begin(self, logs=None):
            seen["samples"] = self.params["samples"]

    m = xor_mlp(dropout=0.0)
    m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device="cpu", seed=1)
    h = m.fit(x, y, epochs=2, batch_size=50, validation_split=0.2, verbose=0, shuffle=False, callbacks=[Spy()])
    assert seen["samples"] == 800 and m.global_step == 2 * 16
    assert "val_loss" in h.history and len(h.history["val_loss"]) == 2
    # the held-out part is the LAST fifth of the data
    want = m.evaluate(x[800:], y[800:], batch_size=200)[0]
    assert abs(h.history["val_loss"][-1] - want) < 1e-6
    import pytest
    with pytest.raises(ValueError):
        m.fit(x, y, epochs=1, batch_size=50, validation_split=1.5, verbose=0)


def test_inspect_checkpoint_tool(tmp_path, capsys):
    from distributed_tensorflow_amd.models import xor_mlp
    from distributed_tensorflow_amd.saver import Saver
    from distributed_tensorflow_amd.tools import inspect_checkpoint
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", device="cpu", seed=1)
    Saver(m.store, m.optimizer).save(str(tmp_path / "model.ckpt"), global_step=7)
    assert inspect_checkpoint.main([str(tmp_path)]) == 0                      # a directory: its latest checkpoint
    out = capsys.readouterr().out
    assert "model.ckpt-7" in out and "dense_1/kernel" in out and "(64, 128)" in out and "dense_3/bias/Adam_1" in out
    assert "32768 bytes" in out
    assert inspect_checkpoint.main([str(tmp_path / "model.ckpt-7"), "--tensor_name", "global_step"]) == 0
    assert "int64" in capsys.readouterr().out
    assert inspect_checkpoint.main([str(tmp_path / "model.ckpt-7"), "--tensor_name", "nope"]) == 1
    (tmp_path / "empty").mkdir()
    assert inspect_checkpoint.main([str(tmp_path / "empty")]) == 1


def _keras_mnist_cnn(device):
    """The architecture and compile line of Keras' own mnist_cnn example."""
    from distributed_tensorflow_amd.layers import Conv2D, Dense, Dropout, Flatten, MaxPooling2D
    from distributed_tensorflow_amd.models import Sequential
    m = Sequential([Conv2D(32, (3, 3), activation="relu", input_shape=(28, 28, 1)), Conv2D(64, (3, 3), activation="relu"),
                    MaxPooling2D((2, 2)), Dropout(0.25), Flatten(), Dense(128, activation="relu"), Dropout(0.5),
                    Dense(10, activation="softmax")])
    m.compile(loss="categorical_crossentropy", optimizer="adadelta", metrics=["accuracy"], device=device, seed=3)
    return m


def _fit_keras_mnist_cnn(device, n, epochs):
    import torch
    from distributed_tensorflow_amd.data import mnist_like
    x, y = mnist_like(n, seed=4)
    x = x.view(n, 28, 28, 1)
    y1 = torch.nn.functional.one_hot(y, 10).float()
    m = _keras_mnist_cnn(device)
    l0 = m.evaluate(x, y1, batch_size=n // 2)[0]
    m.fit(x, y1, epochs=epochs, batch_size=64, verbose=0)
    l1, acc = m.evaluate(x, y1, batch_size=n // 2)[:2]
    return l0, l1, acc, m


def test_keras_mnist_cnn_example_runs_as_written():
    l0, l1, acc, m = _fit_keras_mnist_cnn("cpu", 256, 3)
    assert m.optimizer.name == "Adadelta" and m.optimizer.lr == 1.0
    assert l1 < 0.9 * l0 and acc > 0.3            # twelve steps: it has started to learn


import pytest  # noqa: E402


@pytest.mark.gpu
def test_keras_mnist_cnn_example_on_the_gpu(gpu_device):
    l0, l1, acc, m = _fit_keras_mnist_cnn(gpu_device, 2048, 3)
    print("keras mnist_cnn on the GPU:", l0, l1, acc)
    assert l1 < 0.5 * l0 and acc > 0.8

"""AveragePooling2D / GlobalAveragePooling2D: forward and gradient against PyTorch (TensorFlow's rule for windows over
the border: average over the in-image positions = ``count_include_pad=False``), on the CPU reference path and - marked
gpu - on the HIP kernels; a ResNet-style head (conv, batch norm, global average pooling) trains."""
import pytest
import torch

from distributed_tensorflow_amd.layers import (AveragePooling2D, BatchNormalization, Conv2D, Dense, Flatten,
                                               GlobalAveragePooling2D)
from distributed_tensorflow_amd.models import Sequential
from distributed_tensorflow_amd.ops import autograd as A
from distributed_tensorflow_amd.ops import functional as F

CASES = [((2, 2), None, "valid", (3, 8, 8, 8)), ((3, 3), (2, 2), "same", (2, 9, 7, 4)), ((2, 3), (1, 2), "same", (2, 5, 6, 3)),
         ((3, 3), (1, 1), "valid", (2, 7, 7, 12)), ((7, 7), (7, 7), "valid", (4, 7, 7, 16))]


def _torch_ref(x, pool, strides, padding):
    strides = strides or pool
    H, W = x.shape[1:3]
    OH, OW, pt, pl = F.pool_geometry(H, W, pool[0], pool[1], strides, padding)
    pb = max((OH - 1) * strides[0] + pool[0] - H - pt, 0)
    pr = max((OW - 1) * strides[1] + pool[1] - W - pl, 0)
    xp = x.permute(0, 3, 1, 2)
    ones = torch.ones_like(xp)
    pad = lambda t: torch.nn.functional.pad(t, (pl, pr, pt, pb))
    s = torch.nn.functional.avg_pool2d(pad(xp), pool, strides, divisor_override=1)
    n = torch.nn.functional.avg_pool2d(pad(ones), pool, strides, divisor_override=1)
    return (s / n).permute(0, 2, 3, 1)


def check(device):
    g = torch.Generator().manual_seed(1)
    for pool, strides, padding, shape in CASES:
        x = torch.randn(*shape, generator=g)
        xr = x.double().requires_grad_(True)
        want = _torch_ref(xr, pool, strides, padding)
        w = torch.randn(*want.shape, generator=g)
        (want * w.double()).sum().backward()
        xa = x.clone().to(device).requires_grad_(True)
        got = A.avg_pool2d(xa, pool, strides, padding)
        (got * w.to(device)).sum().backward()
        torch.testing.assert_close(got.detach().cpu().double(), want.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(xa.grad.cpu().double(), xr.grad, rtol=1e-5, atol=1e-6)


def test_cpu():
    check("cpu")


@pytest.mark.gpu
def test_gpu(gpu_device):
    check(gpu_device)


def _head(device):
    m = Sequential([Conv2D(8, (3, 3), padding="same", activation="relu", input_shape=(8, 8, 4)), AveragePooling2D((2, 2)),
                    Conv2D(16, (3, 3), padding="same"), BatchNormalization(momentum=0.8), GlobalAveragePooling2D(), Dense(3, activation="softmax")])
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device=device, seed=2)
    return m


def check_model(device):
    g = torch.Generator().manual_seed(3)
    y = torch.randint(0, 3, (96,), generator=g)
    x = torch.randn(96, 8, 8, 4, generator=g) + y.view(-1, 1, 1, 1).float()         # class = mean level
    m = _head(device)
    assert m.layers[1].output_shape == (4, 4, 8) and m.layers[4].output_shape == (16,)
    l0 = m.evaluate(x, y, batch_size=96)[0]
    m.fit(x, y, epochs=30, batch_size=32, verbose=0)
    l1, acc = m.evaluate(x, y, batch_size=96)[:2]
    assert l1 < 0.6 * l0 and acc > 0.7
    cfg = m.get_config()
    assert [l["class"] for l in Sequential.from_config(cfg).get_config()["layers"]][1::3] == ["AveragePooling2D", "GlobalAveragePooling2D"]


def test_model_cpu():
    check_model("cpu")


@pytest.mark.gpu
def test_model_gpu(gpu_device):
    check_model(gpu_device)


def test_keras_namespace():
    from distributed_tensorflow_amd import compat as tf
    assert issubclass(tf.keras.layers.AveragePooling2D, AveragePooling2D)
    assert issubclass(tf.keras.layers.GlobalAveragePooling2D, GlobalAveragePooling2D)

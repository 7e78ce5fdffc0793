"""HIP convolution-support kernels and the CNN model on the GPU against the PyTorch fp32 CPU reference."""
import pytest
import torch

from distributed_tensorflow_amd.data import mnist_like
from distributed_tensorflow_amd.models import mnist_cnn
from distributed_tensorflow_amd.ops import functional as F

from .test_conv_cpu import _autograd_reference, conv_ref, small_cnn

pytestmark = pytest.mark.gpu

GEOMS = [
    dict(B=3, H=8, W=8, C=3, KH=3, KW=3, strides=(1, 1), padding="same"),
    dict(B=5, H=28, W=28, C=1, KH=5, KW=5, strides=(1, 1), padding="same"),
    dict(B=4, H=14, W=14, C=32, KH=5, KW=5, strides=(1, 1), padding="same"),
    dict(B=2, H=9, W=8, C=4, KH=3, KW=2, strides=(2, 2), padding="same"),
    dict(B=2, H=11, W=9, C=8, KH=4, KW=4, strides=(2, 1), padding="valid"),
]


@pytest.mark.parametrize("g", GEOMS)
def test_im2col_col2im(gpu_device, g):
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(g["B"], g["H"], g["W"], g["C"], generator=gen)
    cols_ref = F.im2col(x, g["KH"], g["KW"], g["strides"], g["padding"])
    cols = F.im2col(x.to(gpu_device), g["KH"], g["KW"], g["strides"], g["padding"])
    assert torch.equal(cols.cpu(), cols_ref)
    c = torch.randn(cols_ref.shape, generator=gen)
    yprev = torch.randn(x.shape, generator=gen)
    for yp in (None, yprev):
        ref = F.col2im(c, x.shape, g["KH"], g["KW"], g["strides"], g["padding"], yprev=yp)
        got = F.col2im(c.to(gpu_device), x.shape, g["KH"], g["KW"], g["strides"], g["padding"],
                       yprev=None if yp is None else yp.to(gpu_device))
        torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("g", GEOMS[:3])
def test_conv_as_gemm_matches_torch_conv2d(gpu_device, g):
    gen = torch.Generator().manual_seed(1)
    Fo = 16
    x = torch.randn(g["B"], g["H"], g["W"], g["C"], generator=gen)
    w = torch.randn(g["KH"], g["KW"], g["C"], Fo, generator=gen) * 0.2
    b = torch.randn(Fo, generator=gen)
    ref = torch.relu(conv_ref(x, w, b, g["strides"], g["padding"]))
    cols = F.im2col(x.to(gpu_device), g["KH"], g["KW"], g["strides"], g["padding"])
    y = F.dense_forward(cols, w.view(-1, Fo).to(gpu_device), b.to(gpu_device), F.ACT_RELU)
    torch.testing.assert_close(y.cpu().view(ref.shape), ref, rtol=1e-4, atol=1e-4)


def test_maxpool(gpu_device):
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(3, 9, 10, 5, generator=gen)
    y_ref, a_ref = F.maxpool_forward(x, (2, 2))
    y, a = F.maxpool_forward(x.to(gpu_device), (2, 2))
    assert torch.equal(y.cpu(), y_ref) and torch.equal(a.cpu(), a_ref)
    dy = torch.randn(y_ref.shape, generator=gen)
    for xr in (None, torch.relu(x)):
        ref = F.maxpool_backward(dy, a_ref, x.shape, (2, 2), x_relu=xr)
        got = F.maxpool_backward(dy.to(gpu_device), a, x.shape, (2, 2), x_relu=None if xr is None else xr.to(gpu_device))
        assert torch.equal(got.cpu(), ref)


def test_cnn_gradients_on_gpu_match_autograd(gpu_device):
    m = small_cnn()
    m.compile(loss="sparse_categorical_crossentropy", optimizer="sgd", metrics=["accuracy"], device=gpu_device, seed=3)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(6, 8, 8, 2, generator=gen)
    y = torch.randint(0, 5, (6,), generator=gen)
    m.compute_gradients(x.to(gpu_device), y.to(gpu_device))
    host = small_cnn()
    host.compile(loss="sparse_categorical_crossentropy", optimizer="sgd", device="cpu", seed=3)
    host.store.flat.copy_(m.store.flat.cpu())
    ref_loss, params = _autograd_reference(host, x, y)
    assert m.read_stats()["loss"] == pytest.approx(ref_loss, rel=1e-4)
    for v, p in zip(m.store.trainable, params):
        torch.testing.assert_close(v.grad.cpu(), p.grad, rtol=2e-3, atol=2e-5, msg=v.name)


def test_mnist_cnn_learns(gpu_device):
    x, y = mnist_like(4096, seed=0, device=gpu_device)
    xt, yt = mnist_like(1024, seed=1, device=gpu_device)
    m = mnist_cnn()
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device=gpu_device, seed=0)
    h = m.fit(x.view(-1, 28, 28, 1), y, epochs=3, batch_size=128, validation_data=(xt.view(-1, 28, 28, 1), yt), verbose=0)
    assert h.history["loss"][-1] < h.history["loss"][0]
    assert h.history["val_acc"][-1] > 0.5

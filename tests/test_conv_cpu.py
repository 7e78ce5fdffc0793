"""Convolution / pooling on the CPU reference path against torch.nn.functional and autograd."""
import pytest
import torch

from distributed_tensorflow_amd.layers import Conv2D, Dense, Dropout, Flatten, MaxPooling2D
from distributed_tensorflow_amd.models import Sequential, mnist_cnn
from distributed_tensorflow_amd.ops import functional as F


def conv_ref(x, w, b, strides, padding):
    """NHWC convolution through torch's NCHW conv2d with TensorFlow padding rules."""
    H, W = x.shape[1:3]
    KH, KW = w.shape[:2]
    OH, OW, pt, pl, pb, pr = F.conv_geometry(H, W, KH, KW, strides, padding)
    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = torch.nn.functional.conv2d(xp, w.permute(3, 2, 0, 1), b, stride=strides)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("geom", [
    dict(H=8, W=8, C=3, KH=3, KW=3, strides=(1, 1), padding="same"),
    dict(H=9, W=7, C=4, KH=5, KW=5, strides=(1, 1), padding="same"),
    dict(H=9, W=8, C=2, KH=3, KW=2, strides=(2, 2), padding="same"),
    dict(H=10, W=10, C=8, KH=3, KW=3, strides=(1, 1), padding="valid"),
    dict(H=11, W=9, C=1, KH=4, KW=4, strides=(2, 1), padding="valid"),
])
def test_im2col_matmul_is_convolution_and_col2im_is_its_adjoint(geom):
    g = torch.Generator().manual_seed(0)
    B, Fo = 3, 5
    x = torch.randn(B, geom["H"], geom["W"], geom["C"], generator=g)
    w = torch.randn(geom["KH"], geom["KW"], geom["C"], Fo, generator=g)
    b = torch.randn(Fo, generator=g)
    cols = F.im2col(x, geom["KH"], geom["KW"], geom["strides"], geom["padding"])
    y = (cols @ w.view(-1, Fo) + b)
    ref = conv_ref(x, w, b, geom["strides"], geom["padding"])
    torch.testing.assert_close(y.view(ref.shape), ref, rtol=1e-4, atol=1e-5)
    c = torch.randn(cols.shape, generator=g)
    back = F.col2im(c, x.shape, geom["KH"], geom["KW"], geom["strides"], geom["padding"])
    assert float((cols * c).sum()) == pytest.approx(float((x * back).sum()), rel=1e-4)


def test_maxpool_matches_torch():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 7, 6, 3, generator=g).requires_grad_(True)
    y, arg = F.maxpool_forward(x.detach(), (2, 2))
    ref = torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    torch.testing.assert_close(y, ref.detach())
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    dx = F.maxpool_backward(dy, arg, x.shape, (2, 2))
    torch.testing.assert_close(dx, x.grad)
    relu_in = torch.relu(x.detach())
    dx2 = F.maxpool_backward(dy, F.maxpool_forward(relu_in, (2, 2))[1], x.shape, (2, 2), x_relu=relu_in)
    x2 = x.detach().clone().requires_grad_(True)
    torch.nn.functional.max_pool2d(torch.relu(x2).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).backward(dy)
    torch.testing.assert_close(dx2, x2.grad)


def _autograd_reference(m, x, y):
    params = [v.tensor.clone().requires_grad_(True) for v in m.store.trainable]
    h, pi = x, 0
    for layer in m.layers:
        if isinstance(layer, Conv2D):
            h = conv_ref(h, params[pi], params[pi + 1], layer.strides, layer.padding)
            pi += 2
            if layer.activation == "relu":
                h = torch.relu(h)
        elif isinstance(layer, MaxPooling2D):
            h = torch.nn.functional.max_pool2d(h.permute(0, 3, 1, 2), layer.pool_size).permute(0, 2, 3, 1)
        elif isinstance(layer, Flatten):
            h = h.reshape(h.shape[0], -1)
        elif isinstance(layer, Dense):
            h = h @ params[pi] + params[pi + 1]
            pi += 2
            if layer.activation == "relu":
                h = torch.relu(h)
    loss = torch.nn.functional.cross_entropy(h, y)
    loss.backward()
    return loss.item(), params


def small_cnn():
    return Sequential([
        Conv2D(4, (3, 3), padding="same", activation="relu", input_shape=(8, 8, 2)),
        MaxPooling2D((2, 2)),
        Conv2D(8, (3, 3), padding="valid", activation="relu"),
        Conv2D(8, (1, 1), activation="relu"),
        Conv2D(6, (2, 2), strides=(1, 1), padding="same"),
        Flatten(),
        Dense(16, activation="relu"),
        Dense(5, activation="softmax"),
    ])


def test_cnn_gradients_match_autograd():
    m = small_cnn()
    m.compile(loss="sparse_categorical_crossentropy", optimizer="sgd", metrics=["accuracy"], device="cpu", seed=3)
    conv1, pool, conv2, conv3 = m.layers[:4]
    assert pool.fold_relu_grad and conv1.grad_is_dz and conv3.fuse_prev_relu and conv2.grad_is_dz
    assert conv1.kernel.shape == (3, 3, 2, 4) and conv1.kernel.name == "conv2d_1/kernel"
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 8, 8, 2, generator=g)
    y = torch.randint(0, 5, (6,), generator=g)
    m.compute_gradients(x, y)
    ref_loss, params = _autograd_reference(m, x, y)
    assert m.read_stats()["loss"] == pytest.approx(ref_loss, rel=1e-5)
    for v, p in zip(m.store.trainable, params):
        torch.testing.assert_close(v.grad, p.grad, rtol=1e-4, atol=1e-6, msg=v.name)


def test_mnist_cnn_shape_and_one_step():
    m = mnist_cnn()
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device="cpu", seed=0)
    assert m.store.num_params == 832 + 51264 + 3136 * 512 + 512 + 5130
    assert [tuple(l.output_shape) for l in m.layers[:5]] == [(28, 28, 32), (14, 14, 32), (14, 14, 64), (7, 7, 64), (3136,)]
    x = torch.rand(4, 28, 28, 1)
    y = torch.tensor([1, 2, 3, 4])
    w0 = m.store.flat.clone()
    logs = m.train_on_batch(x, y)
    assert logs["loss"] > 0 and not torch.equal(w0, m.store.flat) and m.global_step == 1
    assert m.predict(x).shape == (4, 10)
    with pytest.raises(ValueError):
        Conv2D(4, 3, padding="full")
    assert not MaxPooling2D((2, 2), strides=(1, 1)).tiling and MaxPooling2D((2, 2)).tiling     # overlapping windows: general kernels
    with pytest.raises(ValueError):
        MaxPooling2D((2, 2), padding="full")


def test_masked_implicit_tables_switch_off_above_2_to_30_elements():
    """The unpadded (bit-set) gather tables hold 30-bit offsets: tensors of 2^30 elements or more take the padded path."""
    from distributed_tensorflow_amd.layers import Conv2D
    fit = Conv2D._masked_tables_fit
    assert fit(4096, 14, 14, 32, 14, 14, 64, 5, 5, True)
    assert not fit(4096, 14, 14, 32, 14, 14, 64, 7, 7, True)                  # 49 window positions
    assert not fit(85600, 14, 14, 64, 14, 14, 64, 5, 5, False)                # input: 85600*14*14*64 >= 2^30
    assert fit(85000, 14, 14, 32, 14, 14, 64, 5, 5, False)
    assert not fit(85600, 14, 14, 32, 14, 14, 64, 5, 5, True)                 # dgrad gathers the 64-channel output gradient


import pytest as _pytest


def _pool_reference(x, pool, strides, padding):
    """fp64 PyTorch: TensorFlow-style SAME padding (extra at the bottom / right, padded positions never win)."""
    import torch
    xr = x.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    H, W = x.shape[1], x.shape[2]
    xin = xr
    if padding == "same":
        oh, ow = -(-H // strides[0]), -(-W // strides[1])
        th, tw = max((oh - 1) * strides[0] + pool[0] - H, 0), max((ow - 1) * strides[1] + pool[1] - W, 0)
        xin = torch.nn.functional.pad(xr, (tw // 2, tw - tw // 2, th // 2, th - th // 2), value=float("-inf"))
    y = torch.nn.functional.max_pool2d(xin, pool, stride=strides)
    return xr, y


POOL_CASES = [((3, 3), (2, 2), "same"), ((3, 3), (1, 1), "valid"), ((2, 2), (2, 2), "same"), ((3, 2), (2, 1), "same"),
              ((2, 2), (3, 3), "valid")]


@_pytest.mark.parametrize("pool,strides,padding", POOL_CASES)
def test_general_max_pooling_matches_pytorch(pool, strides, padding):
    """Overlapping windows, strides larger than the window, SAME padding on odd sizes: values, arg-max routing and the
    summed gradient of overlapping windows against PyTorch autograd."""
    import torch
    from distributed_tensorflow_amd.ops import functional as F
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 9, 7, 5, generator=g)
    y, arg = F.maxpool_strided_forward(x, pool, strides, padding)
    xr, yr = _pool_reference(x, pool, strides, padding)
    assert tuple(y.shape) == tuple(yr.permute(0, 2, 3, 1).shape)
    torch.testing.assert_close(y.double(), yr.detach().permute(0, 2, 3, 1))
    dy = torch.randn(y.shape, generator=g)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    dx = F.maxpool_strided_backward(dy, arg, x.shape, pool, strides, padding)
    torch.testing.assert_close(dx.double(), xr.grad.permute(0, 2, 3, 1), rtol=1e-6, atol=1e-6)


def test_cnn_with_overlapping_same_pooling_trains():
    import torch
    from distributed_tensorflow_amd.layers import Conv2D, Dense, Flatten, MaxPooling2D
    from distributed_tensorflow_amd.models import Sequential
    m = Sequential([Conv2D(8, (3, 3), padding="same", activation="relu", input_shape=(9, 9, 1)),
                    MaxPooling2D((3, 3), strides=(2, 2), padding="same"), Flatten(), Dense(4, activation="softmax")])
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device="cpu", seed=1)
    assert m.layers[1].output_shape == (5, 5, 8) and not m.layers[1].tiling
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 9, 9, 1, generator=g)
    y = (x.mean(dim=(1, 2, 3)) > 0).long() + 2 * (x[:, :4].mean(dim=(1, 2, 3)) > 0).long()
    first = m.train_on_batch(x, y)["loss"]
    for _ in range(60):
        last = m.train_on_batch(x, y)["loss"]
    assert last < 0.7 * first

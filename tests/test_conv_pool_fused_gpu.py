"""First-layer convolution fused with its 2x2 max pooling (csrc/kernels/conv_pool.hip) against plain PyTorch
(conv2d -> bias -> relu -> max_pool2d and autograd for the filter / bias gradients), and the layer pair inside a model
against the unfused layer chain."""
import pytest
import torch

from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.layers import ACT_CODES, Conv2D, MaxPooling2D
from distributed_tensorflow_amd.ops import functional as F

pytestmark = pytest.mark.gpu


def _reference(x, w, b, act, padding, dpool):
    """x NHWC, w [KH,KW,C,F] (TF layout), fp64 on the CPU."""
    x = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(False)
    w = w.double().cpu().permute(3, 2, 0, 1).clone().requires_grad_(True)      # -> [F, C, KH, KW]
    b = b.double().cpu().clone().requires_grad_(True)
    kh, kw = w.shape[2], w.shape[3]
    if padding == "same":            # TF rule: extra padding goes to the bottom / right
        ph, pw = kh - 1, kw - 1
        x = torch.nn.functional.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    z = torch.nn.functional.conv2d(x, w) + b.view(1, -1, 1, 1)
    if act == "relu":
        z = torch.relu(z)
    y = torch.nn.functional.max_pool2d(z, 2)
    y.backward(dpool.double().cpu().permute(0, 3, 1, 2))
    return y.detach().permute(0, 2, 3, 1), w.grad.permute(2, 3, 1, 0), b.grad


@pytest.mark.parametrize("staged", [1, 2, 0], ids=["staged_in_lds", "staged_mfma_wgrad", "direct_from_global"])
@pytest.mark.parametrize("B,H,W,C,KH,KW,Fo,padding,act", [
    (5, 28, 28, 1, 5, 5, 32, "same", "relu"),        # the MNIST CNN's first layer; groups of two images, the last of one
    (3, 12, 10, 3, 3, 3, 64, "valid", "linear"),     # 27 taps, two channel groups, no activation, ragged last tile
    (1, 6, 6, 2, 2, 2, 32, "same", "relu"),          # 9 pooled pixels: one full tile and one nearly empty one
    (37, 28, 28, 1, 5, 5, 32, "same", "relu"),       # an odd number of images: a short last group
])
def test_fused_conv_pool_matches_pytorch(gpu_device, B, H, W, C, KH, KW, Fo, padding, act, staged):
    from distributed_tensorflow_amd.ops import _native as nv
    nv.kernels().dtf_conv_pool_set_staged(staged)
    try:
        _check_fused_conv_pool(gpu_device, B, H, W, C, KH, KW, Fo, padding, act, staged)
    finally:
        nv.kernels().dtf_conv_pool_set_staged(1)


def _check_fused_conv_pool(gpu_device, B, H, W, C, KH, KW, Fo, padding, act, staged):
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, H, W, C, generator=g).to(gpu_device)
    w = (torch.randn(KH, KW, C, Fo, generator=g) * 0.3).to(gpu_device)
    b = (torch.randn(Fo, generator=g) * 0.1).to(gpu_device)
    assert F.conv_pool_supported(x.shape, KH, KW, Fo, (1, 1), padding, (2, 2))
    OH, OW, pt, pl = F.conv_geometry(H, W, KH, KW, (1, 1), padding)[:4]
    assert bool(F.conv_pool_staged_fits(B, H, W, C, KH, KW, pt, pl, OH // 2, OW // 2, Fo)) == bool(staged)
    ypool = torch.empty(B, OH // 2, OW // 2, Fo, device=gpu_device)
    arg = torch.empty(B, OH // 2, OW // 2, Fo, dtype=torch.uint8, device=gpu_device)
    code = ACT_CODES[act]
    F.conv_pool_forward(x, w.view(-1, Fo), b, code, KH, KW, padding, ypool, arg)
    dpool = torch.randn(ypool.shape, generator=g).to(gpu_device)
    ry, rdw, rdb = _reference(x, w, b, act, padding, dpool)
    torch.testing.assert_close(ypool.double().cpu(), ry, rtol=1e-5, atol=1e-5)
    assert int(arg.max()) <= 3

    dw = torch.full((KH * KW * C, Fo), float("nan"), device=gpu_device)
    db = torch.full((Fo,), float("nan"), device=gpu_device)
    ws = torch.empty(F.conv_pool_workspace_floats(C, KH, KW, Fo), device=gpu_device)
    F.conv_pool_wgrad(x, dpool, ypool, arg, dw, db, code, KH, KW, padding, ws)
    torch.testing.assert_close(dw.double().cpu().view(KH, KW, C, Fo), rdw, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.double().cpu(), rdb, rtol=1e-4, atol=1e-4)


def test_shapes_outside_the_fused_kernels_are_reported(gpu_device):
    assert not F.conv_pool_supported((4, 28, 28, 32), 5, 5, 64)           # 800 taps: the implicit-GEMM path's job
    assert not F.conv_pool_supported((4, 28, 28, 1), 5, 5, 10)            # 10 filters: not a multiple of 32
    assert not F.conv_pool_supported((4, 27, 27, 1), 5, 5, 32, padding="same")    # odd conv output: windows do not tile it
    assert not F.conv_pool_supported((4, 28, 28, 1), 5, 5, 32, strides=(2, 2))


def test_model_step_with_the_fused_pair_matches_the_unfused_chain(gpu_device):
    g = torch.Generator().manual_seed(4)
    x = torch.rand(256, 28, 28, 1, generator=g).to(gpu_device)
    y = torch.randint(0, 10, (256,), generator=g).to(gpu_device)
    grads, losses = [], []
    for fused in (True, False):
        m = models.mnist_cnn(dropout=0.0)
        m.compile(loss="sparse_categorical_crossentropy", optimizer="sgd", metrics=["accuracy"], device=gpu_device, seed=9)
        conv, pool = m._effective[0], m._effective[1]
        assert isinstance(conv, Conv2D) and isinstance(pool, MaxPooling2D) and conv.fused_pool is pool
        if not fused:
            conv.fused_pool, pool.fused_into = None, None
        hist = m.fit(x, y, batch_size=128, epochs=2, shuffle=False, verbose=0)
        assert conv._fused_ran == fused
        losses.append(hist.history["loss"])
        grads.append(m.store.flat.clone())
    for a, c in zip(*losses):
        assert abs(a - c) < 1e-4
    torch.testing.assert_close(grads[0], grads[1], rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize("C", [1, 3, 8])
def test_conv_pad_any_channel_count(gpu_device, C):
    """Zero padding of an NHWC image (the fused first layer pads a 'same' input once): float4 kernel for channel
    counts divisible by 4, scalar kernel otherwise, against torch.nn.functional.pad."""
    g = torch.Generator().manual_seed(C)
    x = torch.randn(3, 9, 7, C, generator=g).to(gpu_device)
    out = F.conv_pad(x, 2, 1, 9 + 2 + 3, 7 + 1 + 2)
    ref = torch.nn.functional.pad(x, (0, 0, 1, 2, 2, 3))
    assert torch.equal(out, ref)


def test_same_padded_three_channel_first_layer_through_the_layer_pair(gpu_device):
    """Conv2D(3 -> 32, 3x3, same) + MaxPooling2D as layers: the fused pair (pad once, VALID kernels) against the
    unfused chain, forward output and filter / bias gradients."""
    from distributed_tensorflow_amd.models import Sequential
    from distributed_tensorflow_amd.layers import Dense, Flatten
    g = torch.Generator().manual_seed(8)
    x = torch.randn(16, 12, 12, 3, generator=g).to(gpu_device)
    y = torch.randint(0, 10, (16,), generator=g).to(gpu_device)
    flats = []
    for fused in (True, False):
        m = Sequential()
        m.add(Conv2D(32, (3, 3), padding="same", activation="relu", input_shape=(12, 12, 3)))
        m.add(MaxPooling2D((2, 2)))
        m.add(Flatten())
        m.add(Dense(10, activation="softmax"))
        m.compile(loss="sparse_categorical_crossentropy", optimizer="sgd", device=gpu_device, seed=4)
        if not fused:
            m._effective[0].fused_pool, m._effective[1].fused_into = None, None
        for _ in range(3):
            m.train_step(x, y)
        assert m._effective[0]._fused_ran == fused
        flats.append(m.store.flat.clone())
    torch.testing.assert_close(flats[0], flats[1], rtol=1e-4, atol=1e-6)

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


# scalar and 4-channel vector kernels; 2x2 windows tiling the image exactly take a kernel of their own backwards
@pytest.mark.parametrize("C,pool,H", [(5, (2, 2), 9), (8, (2, 2), 9), (32, (3, 2), 9), (8, (2, 2), 8), (64, (2, 2), 14)])
def test_maxpool(gpu_device, C, pool, H):
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(3, H, 10, C, generator=gen)
    y_ref, a_ref = F.maxpool_forward(x, pool)
    y, a = F.maxpool_forward(x.to(gpu_device), pool)
    assert torch.equal(y.cpu(), y_ref) and torch.equal(a.cpu(), a_ref)
    dy = torch.randn(y_ref.shape, generator=gen)
    for xr in (None, torch.relu(x)):
        ref = F.maxpool_backward(dy, a_ref, x.shape, pool, x_relu=xr)
        got = F.maxpool_backward(dy.to(gpu_device), a, x.shape, pool, x_relu=None if xr is None else xr.to(gpu_device))
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


@pytest.mark.parametrize("g", [
    dict(B=4, H=14, W=14, C=32, N=64, KH=5, KW=5, padding="same"),
    dict(B=3, H=9, W=11, C=64, N=32, KH=3, KW=3, padding="valid"),
    dict(B=2, H=8, W=8, C=32, N=96, KH=2, KW=3, padding="same"),
    dict(B=70, H=14, W=14, C=32, N=64, KH=5, KW=5, padding="same"),
    # channel counts that are multiples of 4 only: a k tile spans several window positions, K has a tail
    dict(B=5, H=9, W=10, C=4, N=8, KH=3, KW=3, padding="same"),
    dict(B=3, H=12, W=12, C=12, N=20, KH=5, KW=5, padding="valid"),
    dict(B=4, H=7, W=9, C=40, N=24, KH=2, KW=3, padding="same"),
    dict(B=9, H=6, W=6, C=8, N=4, KH=3, KW=1, padding="same"),
])
def test_implicit_gemm_conv_matches_patch_matrix_path(gpu_device, g):
    """Forward, weight gradient and data gradient with the patches gathered inside the GEMM against the same
    products on a materialised patch matrix (and, through that, torch conv2d)."""
    gen = torch.Generator().manual_seed(4)
    B, H, W, C, N, KH, KW = (g[k] for k in ("B", "H", "W", "C", "N", "KH", "KW"))
    assert F.conv_implicit_supported(C, N, (1, 1))
    x = torch.randn(B, H, W, C, generator=gen).to(gpu_device)
    w = (torch.randn(KH, KW, C, N, generator=gen) * 0.1).to(gpu_device)
    b = torch.randn(N, generator=gen).to(gpu_device)
    OH, OW, pt, pl, pb, pr = F.conv_geometry(H, W, KH, KW, (1, 1), g["padding"])
    M, K = B * OH * OW, KH * KW * C
    # reference: explicit patch matrix
    cols = F.im2col(x, KH, KW, (1, 1), g["padding"])
    y_ref = F.dense_forward(cols, w.view(K, N), b, F.ACT_RELU)
    dz = torch.randn(M, N, generator=gen).to(gpu_device)
    dw_ref, db_ref = torch.empty(K, N, device=gpu_device), torch.empty(N, device=gpu_device)
    F.dense_backward_dw(cols, dz, dw_ref, db_ref)
    yprev = torch.randn(B, H, W, C, generator=gen).to(gpu_device)
    dx_ref = F.col2im(F.dense_backward_dx(dz, w.view(K, N)), x.shape, KH, KW, (1, 1), g["padding"], yprev=yprev)
    # implicit
    Hp, Wp = H + pt + pb, W + pl + pr
    xpad = F.conv_pad(x, pt, pl, Hp, Wp)
    assert torch.equal(xpad[:, pt:pt + H, pl:pl + W, :], x) and float(xpad.sum()) == pytest.approx(float(x.sum()), rel=1e-4)
    rowtab, shift = F.conv_tables(B, OH, OW, Hp, Wp, KH, KW, C, gpu_device)
    y = F.conv_gemm_forward(xpad, rowtab, shift, w.view(K, N), b, F.ACT_RELU, torch.empty(M, N, device=gpu_device))
    torch.testing.assert_close(y, y_ref, rtol=1e-4, atol=1e-4)
    dw, db = torch.full((K, N), float("nan"), device=gpu_device), torch.full((N,), float("nan"), device=gpu_device)
    ws = torch.empty(max(F.dense_dw_workspace_floats(M, N, K), 1), device=gpu_device)
    F.conv_gemm_wgrad(xpad, rowtab, shift, dz, dw, db, ws)
    torch.testing.assert_close(dw, dw_ref, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(db, db_ref, rtol=1e-3, atol=1e-3)
    top, left = KH - 1 - pt, KW - 1 - pl
    Hq, Wq = H + KH - 1, W + KW - 1
    dzpad = F.conv_pad(dz.view(B, OH, OW, N), top, left, Hq, Wq)
    rowtab_d, shift_d = F.conv_tables(B, H, W, Hq, Wq, KH, KW, N, gpu_device)
    wflip = F.conv_flip_weights(w)
    dx = F.conv_gemm_dgrad(dzpad, rowtab_d, shift_d, wflip, torch.empty(B * H * W, C, device=gpu_device), yprev=yprev.view(-1, C))
    torch.testing.assert_close(dx.view(B, H, W, C), dx_ref, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("c0,c1,c2", [(32, 32, 64), (4, 12, 20)])
def test_cnn_with_implicit_convs_matches_autograd(gpu_device, c0, c1, c2):
    """A model whose convolutions take the implicit-GEMM path (4-channel multiples, stride 1): every gradient against
    PyTorch autograd on the CPU."""
    from distributed_tensorflow_amd.layers import Conv2D, Dense, Flatten, MaxPooling2D
    from distributed_tensorflow_amd.models import Sequential

    def build():
        return Sequential([
            Conv2D(c1, (3, 3), padding="same", activation="relu", input_shape=(8, 8, c0)),
            Conv2D(c2, (3, 3), padding="valid", activation="relu"),
            MaxPooling2D((2, 2)),
            Conv2D(c1, (2, 2), padding="same", activation="relu"),
            Flatten(),
            Dense(5, activation="softmax"),
        ])
    m = build()
    m.compile(loss="sparse_categorical_crossentropy", optimizer="sgd", metrics=["accuracy"], device=gpu_device, seed=3)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(6, 8, 8, c0, generator=gen)
    y = torch.randint(0, 5, (6,), generator=gen)
    m.compute_gradients(x.to(gpu_device), y.to(gpu_device))
    convs = [l for l in m.layers if isinstance(l, Conv2D)]
    assert [c._used_implicit for c in convs] == [True, True, True]
    host = build()
    host.compile(loss="sparse_categorical_crossentropy", optimizer="sgd", device="cpu", seed=3)
    host.store.flat.copy_(m.store.flat.cpu())
    ref_loss, params = _autograd_reference(host, x, y)
    assert m.read_stats()["loss"] == pytest.approx(ref_loss, rel=1e-4)
    for v, p in zip(m.store.trainable, params):
        torch.testing.assert_close(v.grad.cpu(), p.grad, rtol=2e-3, atol=2e-5, msg=v.name)


@pytest.mark.parametrize("g", [
    dict(B=4, H=14, W=14, C=32, N=64, KH=5, KW=5, padding="same"),
    dict(B=3, H=10, W=12, C=64, N=32, KH=3, KW=3, padding="valid"),
    dict(B=2, H=8, W=8, C=32, N=96, KH=2, KW=3, padding="same"),
    dict(B=3, H=8, W=10, C=12, N=20, KH=3, KW=3, padding="same"),
    dict(B=2, H=6, W=6, C=4, N=8, KH=5, KW=5, padding="same"),
])
def test_unpadded_gather_and_pooling_epilogue(gpu_device, g):
    """The image is NOT zero-padded: per-pixel tap masks make the loaders read border taps as zero.  Forward, weight
    gradient and data gradient must equal the padded-copy path bit for bit (same products, same order); with
    pool-ordered tables the forward GEMM's epilogue must produce exactly what conv -> ReLU -> 2x2 max pooling does,
    arg-max included, and the pooled-mask pooling gradient what the full-resolution mask gives."""
    gen = torch.Generator().manual_seed(9)
    B, H, W, C, N, KH, KW = (g[k] for k in ("B", "H", "W", "C", "N", "KH", "KW"))
    x = torch.randn(B, H, W, C, generator=gen).to(gpu_device)
    w = (torch.randn(KH, KW, C, N, generator=gen) * 0.1).to(gpu_device)
    b = torch.randn(N, generator=gen).to(gpu_device)
    OH, OW, pt, pl, pb, pr = F.conv_geometry(H, W, KH, KW, (1, 1), g["padding"])
    M, K = B * OH * OW, KH * KW * C
    Hp, Wp = H + pt + pb, W + pl + pr
    xpad = F.conv_pad(x, pt, pl, Hp, Wp)
    rt_p, sh_p = F.conv_tables(B, OH, OW, Hp, Wp, KH, KW, C, gpu_device)
    rt, sh, mk = F.conv_tables_unpadded(B, OH, OW, H, W, KH, KW, pt, pl, C, gpu_device)
    y_ref = F.conv_gemm_forward(xpad, rt_p, sh_p, w.view(K, N), b, F.ACT_RELU, torch.empty(M, N, device=gpu_device))
    y = F.conv_gemm_forward(x, rt, sh, w.view(K, N), b, F.ACT_RELU, torch.empty(M, N, device=gpu_device), masktab=mk)
    assert torch.equal(y, y_ref)
    dz = torch.randn(M, N, generator=gen).to(gpu_device)
    ws = torch.empty(max(F.dense_dw_workspace_floats(M, N, K), 1), device=gpu_device)
    dw_ref, db_ref = torch.empty(K, N, device=gpu_device), torch.empty(N, device=gpu_device)
    F.conv_gemm_wgrad(xpad, rt_p, sh_p, dz, dw_ref, db_ref, ws)
    dw, db = torch.full((K, N), float("nan"), device=gpu_device), torch.full((N,), float("nan"), device=gpu_device)
    F.conv_gemm_wgrad(x, rt, sh, dz, dw, db, ws, masktab=mk)
    assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
    top, left = KH - 1 - pt, KW - 1 - pl
    dzpad = F.conv_pad(dz.view(B, OH, OW, N), top, left, H + KH - 1, W + KW - 1)
    rtd_p, shd_p = F.conv_tables(B, H, W, H + KH - 1, W + KW - 1, KH, KW, N, gpu_device)
    rtd, shd, mkd = F.conv_tables_unpadded(B, H, W, OH, OW, KH, KW, top, left, N, gpu_device)
    wflip = F.conv_flip_weights(w)
    dx_ref = F.conv_gemm_dgrad(dzpad, rtd_p, shd_p, wflip, torch.empty(B * H * W, C, device=gpu_device))
    dx = F.conv_gemm_dgrad(dz.view(B, OH, OW, N), rtd, shd, wflip, torch.empty(B * H * W, C, device=gpu_device), masktab=mkd)
    assert torch.equal(dx, dx_ref)
    if OH % 2 == 0 and OW % 2 == 0:
        rt2, sh2, mk2 = F.conv_tables_unpadded(B, OH, OW, H, W, KH, KW, pt, pl, C, gpu_device, pool_order=True)
        yp = torch.full((B * (OH // 2) * (OW // 2), N), float("nan"), device=gpu_device)
        arg = torch.full((B, OH // 2, OW // 2, N), 255, dtype=torch.uint8, device=gpu_device)
        F.conv_gemm_forward(x, rt2, sh2, w.view(K, N), b, F.ACT_RELU, yp, masktab=mk2, pool_arg=arg)
        yp_ref, arg_ref = F.maxpool_forward(y_ref.view(B, OH, OW, N), (2, 2))
        assert torch.equal(yp.view(yp_ref.shape), yp_ref) and torch.equal(arg, arg_ref)
        dyp = torch.randn(yp_ref.shape, generator=gen).to(gpu_device)
        full = F.maxpool_backward(dyp, arg_ref, (B, OH, OW, N), (2, 2), x_relu=y_ref.view(B, OH, OW, N))
        pooled = F.maxpool_backward(dyp, arg, (B, OH, OW, N), (2, 2), x_relu=yp.view(yp_ref.shape), mask_pooled=True)
        assert torch.equal(pooled, full)
        # CPU reference of the pooled-mask variant
        cpu = F.maxpool_backward(dyp.cpu(), arg.cpu(), (B, OH, OW, N), (2, 2), x_relu=yp.view(yp_ref.shape).cpu(), mask_pooled=True)
        assert torch.equal(cpu, full.cpu())


@pytest.mark.parametrize("pool,strides,padding", [((3, 3), (2, 2), "same"), ((3, 3), (1, 1), "valid"), ((2, 2), (2, 2), "same"),
                                                  ((3, 2), (2, 1), "same"), ((2, 2), (3, 3), "valid")])
def test_general_max_pooling_kernels_match_the_cpu_path(gpu_device, pool, strides, padding):
    """The strided / SAME-padded pooling kernels against the CPU implementation (itself checked against PyTorch autograd
    in tests/test_conv_cpu.py): values, arg-max and the gradient summed over overlapping windows."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 13, 10, 6, generator=g)
    y, arg = F.maxpool_strided_forward(x, pool, strides, padding)
    yg, argg = F.maxpool_strided_forward(x.to(gpu_device), pool, strides, padding)
    assert torch.equal(yg.cpu(), y) and torch.equal(argg.cpu(), arg)
    dy = torch.randn(y.shape, generator=g)
    xr = torch.relu(x)
    for relu in (None, xr):
        dx = F.maxpool_strided_backward(dy, arg, x.shape, pool, strides, padding, x_relu=relu)
        dxg = F.maxpool_strided_backward(dy.to(gpu_device), argg, x.shape, pool, strides, padding,
                                         x_relu=None if relu is None else relu.to(gpu_device))
        torch.testing.assert_close(dxg.cpu(), dx, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("g", [
    dict(B=3, H=9, W=11, C=8, N=12, KH=3, KW=3, strides=(2, 2), padding="same"),
    dict(B=2, H=12, W=10, C=32, N=16, KH=5, KW=5, strides=(2, 1), padding="valid"),
    dict(B=4, H=8, W=8, C=4, N=8, KH=2, KW=2, strides=(2, 2), padding="valid"),
    dict(B=2, H=13, W=13, C=12, N=20, KH=7, KW=7, strides=(3, 3), padding="same"),     # 49 taps: the padded-copy tables
])
def test_strided_implicit_gemm_conv_matches_patch_matrix_path(gpu_device, g):
    """Strided convolutions gather their patches inside the GEMM too (forward and filter gradient): window origins step
    by the strides, in the unpadded (bit set of in-image taps) and in the padded-copy form of the tables."""
    gen = torch.Generator().manual_seed(8)
    B, H, W, C, N, KH, KW = (g[k] for k in ("B", "H", "W", "C", "N", "KH", "KW"))
    st, pad = g["strides"], g["padding"]
    assert F.conv_implicit_supported(C, N, st)
    x = torch.randn(B, H, W, C, generator=gen).to(gpu_device)
    w = (torch.randn(KH, KW, C, N, generator=gen) * 0.1).to(gpu_device)
    b = torch.randn(N, generator=gen).to(gpu_device)
    OH, OW, pt, pl, pb, pr = F.conv_geometry(H, W, KH, KW, st, pad)
    M, K = B * OH * OW, KH * KW * C
    cols = F.im2col(x, KH, KW, st, pad)
    y_ref = F.dense_forward(cols, w.view(K, N), b, F.ACT_RELU)
    dz = torch.randn(M, N, generator=gen).to(gpu_device)
    dw_ref, db_ref = torch.empty(K, N, device=gpu_device), torch.empty(N, device=gpu_device)
    F.dense_backward_dw(cols, dz, dw_ref, db_ref)
    ws = torch.empty(max(F.dense_dw_workspace_floats(M, N, K), 1), device=gpu_device)
    forms = []
    if KH * KW <= 32:
        rowtab, shift, mask = F.conv_tables_unpadded(B, OH, OW, H, W, KH, KW, pt, pl, C, gpu_device, strides=st)
        forms.append((x, rowtab, shift, mask))
    Hp, Wp = H + pt + pb, W + pl + pr
    rowtab, shift = F.conv_tables(B, OH, OW, Hp, Wp, KH, KW, C, gpu_device, strides=st)
    forms.append((F.conv_pad(x, pt, pl, Hp, Wp), rowtab, shift, None))
    for src, rowtab, shift, mask in forms:
        y = F.conv_gemm_forward(src, rowtab, shift, w.view(K, N), b, F.ACT_RELU, torch.empty(M, N, device=gpu_device),
                                masktab=mask)
        torch.testing.assert_close(y, y_ref, rtol=1e-4, atol=1e-4)
        dw, db = torch.full((K, N), float("nan"), device=gpu_device), torch.full((N,), float("nan"), device=gpu_device)
        F.conv_gemm_wgrad(src, rowtab, shift, dz, dw, db, ws, masktab=mask)
        torch.testing.assert_close(dw, dw_ref, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(db, db_ref, rtol=1e-3, atol=1e-3)


def test_cnn_with_strided_convs_matches_autograd(gpu_device):
    """Strided layers in a model: implicit forward / filter gradient, data gradient through col2im; against autograd."""
    from distributed_tensorflow_amd.layers import Conv2D, Dense, Flatten
    from distributed_tensorflow_amd.models import Sequential

    def build():
        return Sequential([
            Conv2D(8, (3, 3), strides=(2, 2), padding="same", activation="relu", input_shape=(11, 11, 4)),
            Conv2D(12, (3, 3), strides=(2, 1), padding="valid", activation="relu"),
            Conv2D(8, (2, 2), padding="same", activation="relu"),
            Flatten(),
            Dense(5, activation="softmax"),
        ])
    m = build()
    m.compile(loss="sparse_categorical_crossentropy", optimizer="sgd", metrics=["accuracy"], device=gpu_device, seed=3)
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(6, 11, 11, 4, generator=gen)
    y = torch.randint(0, 5, (6,), generator=gen)
    m.compute_gradients(x.to(gpu_device), y.to(gpu_device))
    from distributed_tensorflow_amd.layers import Conv2D as C2
    assert [c._used_implicit for c in m.layers if isinstance(c, C2)] == [True, True, True]
    host = build()
    host.compile(loss="sparse_categorical_crossentropy", optimizer="sgd", device="cpu", seed=3)
    host.store.flat.copy_(m.store.flat.cpu())
    ref_loss, params = _autograd_reference(host, x, y)
    assert m.read_stats()["loss"] == pytest.approx(ref_loss, rel=1e-4)
    for v, p in zip(m.store.trainable, params):
        torch.testing.assert_close(v.grad.cpu(), p.grad, rtol=2e-3, atol=2e-5, msg=v.name)

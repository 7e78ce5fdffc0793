"""Sequential on the CPU reference path: the fused execution plan must equal plain PyTorch autograd."""
import pytest
import torch

from distributed_tensorflow_amd.data import get_data, mnist_like
from distributed_tensorflow_amd.layers import Activation, Dense, Dropout, Flatten
from distributed_tensorflow_amd.models import Sequential, mnist_mlp, xor_mlp
from distributed_tensorflow_amd.ops import functional as F


def torch_reference_step(m, x, y, loss, masks):
    """Forward/backward of the same stack with autograd; returns flat gradients in store layout."""
    params = [v.tensor.clone().requires_grad_(True) for v in m.store.trainable]
    h, pi, mi = x, 0, 0
    for layer in m.layers:
        if isinstance(layer, Dense):
            h = h @ params[pi] + params[pi + 1]
            pi += 2
            if layer.activation == "relu":
                h = torch.relu(h)
            elif layer.activation == "sigmoid":
                h = torch.sigmoid(h)
        elif isinstance(layer, Dropout):
            h = h * masks[mi]
            mi += 1
    if loss == "mse":
        l = ((h - y) ** 2).mean()
    else:
        l = torch.nn.functional.cross_entropy(h, y)
    l.backward()
    return l.item(), params


def test_fusion_plan_flags():
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", device="cpu", seed=0)
    d1, dr1, d2, dr2, d3 = m.layers
    assert d1.fused_keep == pytest.approx(0.7) and dr1.fused_into_previous and dr2.fused_into_previous
    assert d2.fuse_prev_grad and d3.fuse_prev_grad and d1.grad_is_dz and d2.grad_is_dz and d3.grad_is_dz
    assert not d1.need_input_grad and d3.prev_scale == pytest.approx(1 / 0.7)
    assert [v.name for v in m.store.trainable] == ["dense_1/kernel", "dense_1/bias", "dense_2/kernel", "dense_2/bias",
                                                   "dense_3/kernel", "dense_3/bias"]
    assert m.store.num_params == 28960


@pytest.mark.parametrize("which", ["xor", "mnist"])
def test_gradients_match_autograd(which):
    if which == "xor":
        m = xor_mlp()
        m.compile(loss="mean_squared_error", optimizer="adam", device="cpu", seed=1)
        x, y = get_data(64, val_set_size=0, seed=2)[:2]
        loss = "mse"
    else:
        m = mnist_mlp()
        m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", device="cpu", seed=1)
        x, y = mnist_like(64, seed=2)
        loss = "xent"
    m.compute_gradients(x, y)
    # the masks the fused epilogue used are a pure function of (seed, step, index)
    masks = []
    for layer in m.layers:
        if isinstance(layer, Dense) and layer.fused_keep < 1.0:
            u = F.dropout_uniform(F.dropout_key(layer.seed, 0), 64 * layer.units).view(64, layer.units)
            masks.append((u < layer.fused_keep).float() / layer.fused_keep)
    ref_loss, params = torch_reference_step(m, x, y, loss, masks)
    logs = m.read_stats()
    assert abs(logs["loss"] - ref_loss) < 1e-5
    for v, p in zip(m.store.trainable, params):
        torch.testing.assert_close(v.grad, p.grad, rtol=1e-4, atol=1e-7)


def test_unfused_layers_match_fused_plan():
    """Dense+Activation+Dropout as separate layers computes the same as the folded form."""
    torch.manual_seed(0)
    x, y = get_data(32, val_set_size=0, seed=5)[:2]
    fused = Sequential([Dense(16, input_dim=64, activation="relu"), Dense(32, activation="sigmoid")])
    fused.compile(loss="mean_squared_error", optimizer="sgd", device="cpu", seed=3)
    split = Sequential([Dense(16, input_dim=64), Activation("relu"), Dense(32), Activation("sigmoid")])
    split.compile(loss="mean_squared_error", optimizer="sgd", device="cpu", seed=3)
    split.store.flat.copy_(fused.store.flat)
    fused.compute_gradients(x, y)
    split.compute_gradients(x, y)
    torch.testing.assert_close(fused.store.flat_grad, split.store.flat_grad, rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(fused.predict(x), split.predict(x))


def test_adam_step_matches_tf_formula_not_torch_adam():
    m = Sequential([Dense(4, input_dim=3)])
    m.compile(loss="mean_squared_error", optimizer="adam", device="cpu", seed=0)
    w0 = m.store.flat.clone()
    x, y = torch.randn(8, 3), torch.randn(8, 4)
    m.train_step(x, y)
    g = m.store.flat_grad
    # after one step m=.1g, v=.001g^2, lr_t = lr*sqrt(1-b2)/(1-b1): update = lr_t * .1 g / (sqrt(.001) |g| + eps)
    lr_t = 1e-3 * (1 - 0.999) ** 0.5 / (1 - 0.9)
    expect = w0 - lr_t * 0.1 * g / ((0.001 * g * g).sqrt() + 1e-8)
    torch.testing.assert_close(m.store.flat, expect, rtol=1e-5, atol=1e-9)
    assert m.global_step == 1


def test_xor_converges_on_cpu():
    xt, yt, xv, yv = get_data(6000, val_set_size=500, seed=11)
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device="cpu", seed=4)
    h = m.fit(xt, yt, epochs=12, batch_size=50, validation_data=(xv, yv), verbose=0)
    # 1440 steps: well past chance (0.5) and still climbing; the full 30000-step run is a GPU test
    assert h.history["val_acc"][-1] > 0.58 and h.history["val_acc"][-1] > h.history["val_acc"][0]
    assert h.history["loss"][-1] < h.history["loss"][0]


def test_keras_accuracy_rule_and_evaluate_predict():
    xt, yt, xv, yv = get_data(200, val_set_size=50, seed=1)
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", metrics=["accuracy", "binary_rounded_accuracy"], device="cpu", seed=0)
    logs = m.train_on_batch(xt[:50], yt[:50])
    assert set(logs) == {"loss", "acc", "binary_rounded_accuracy"}
    p = m.predict(xv)
    assert p.shape == (50, 32) and float(p.min()) >= 0 and float(p.max()) <= 1
    ev = m.evaluate(xv, yv, batch_size=16)
    assert ev[1] == pytest.approx((p.argmax(1) == yv.argmax(1)).float().mean().item(), abs=1e-6)   # categorical, as Keras
    assert ev[0] == pytest.approx(((p - yv) ** 2).mean().item(), rel=1e-4)
    c = mnist_mlp()
    c.compile(loss="categorical_crossentropy", optimizer="sgd", metrics=["accuracy"], device="cpu", seed=0)
    x, y = mnist_like(40, seed=0, one_hot=True)
    probs = c.predict(x)
    torch.testing.assert_close(probs.sum(1), torch.ones(40))
    with pytest.raises(ValueError):
        Sequential([Dense(3, input_dim=2, activation="relu")]).compile(loss="categorical_crossentropy", device="cpu")
    with pytest.raises(ValueError):
        Sequential([Flatten(input_shape=(2, 2)), Dense(3), Dense(2, input_dim=3)]).compile(loss="nope", device="cpu")


def test_data_generators():
    a = get_data(100, seed=3)
    b = get_data(100, seed=3)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    xt, yt, xv, yv = a
    assert xt.shape == (100, 64) and yt.shape == (100, 32) and xv.shape == (1000, 64)
    assert torch.equal(yt, (xt[:, :32] != xt[:, 32:]).float())
    lists = get_data(10, val_set_size=5, seed=3, as_lists=True)
    assert isinstance(lists[0], list) and lists[0][0][0] in (0, 1) and len(lists[2]) == 5
    assert not torch.equal(get_data(100)[0], get_data(100)[0])          # unseeded like the reference
    x, y = mnist_like(256, seed=1)
    assert x.shape == (256, 784) and y.dtype == torch.int64 and 0 <= int(y.min()) and int(y.max()) <= 9
    assert len(torch.unique(y)) >= 5


def test_matmul_dtype_switch_is_validated_and_harmless_on_cpu():
    """compile(matmul_dtype=...) drives a process-wide switch of the GPU GEMM kernels; on the CPU reference path the
    three settings compute the same fp32 numbers."""
    from distributed_tensorflow_amd.ops import functional as F
    x = torch.rand(64, 64)
    y = (torch.rand(64, 32) > 0.5).float()
    outs = []
    try:
        for mode in ("fp32", "fp32_split", "bf16"):
            m = xor_mlp(dropout=0.0)
            m.compile(loss="mean_squared_error", optimizer="adam", device="cpu", seed=1, matmul_dtype=mode)
            m.fit(x, y, batch_size=32, epochs=2, shuffle=False, verbose=0)
            outs.append(m.store.flat.clone())
        torch.testing.assert_close(outs[0], outs[1], rtol=0, atol=0)
        torch.testing.assert_close(outs[0], outs[2], rtol=0, atol=0)
        with pytest.raises(ValueError):
            xor_mlp().compile(matmul_dtype="fp16")
    finally:
        F.set_gemm_split_products(0)


def test_he_uniform_and_device_side_initialisation():
    """He-uniform bounds; the counter-based initialiser that large models use on their device gives U(-l, l) with the
    same bounds, is a pure function of (seed, variable, element) and leaves biases at zero."""
    import math
    from distributed_tensorflow_amd.layers import Dense
    from distributed_tensorflow_amd.models import Sequential
    from distributed_tensorflow_amd.ops import functional as F
    from distributed_tensorflow_amd.variables import he_uniform
    w = he_uniform((600, 50), torch.Generator().manual_seed(0))
    lim = math.sqrt(6.0 / 600)
    assert w.abs().max() <= lim and w.abs().max() > 0.95 * lim and abs(w.mean()) < 0.01
    a = F.init_uniform_(torch.empty(100000), 7, -0.5, 0.5)
    b = F.init_uniform_(torch.empty(100000), 7, -0.5, 0.5)
    c = F.init_uniform_(torch.empty(100000), 8, -0.5, 0.5)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert a.min() >= -0.5 and a.max() < 0.5 and abs(a.mean()) < 0.01 and abs(a.std() - 1 / math.sqrt(12)) < 0.01
    # neighbouring elements are uncorrelated
    assert abs(torch.corrcoef(torch.stack([a[:-1], a[1:]]))[0, 1]) < 0.02

    def make():
        m = Sequential([Dense(64, input_dim=32, activation="relu", kernel_initializer="he_uniform"), Dense(10, activation="softmax")])
        m.build(device="cpu", seed=3)
        return m
    m1 = make()
    assert not m1.store.device_initialised
    from distributed_tensorflow_amd.variables import VariableStore
    st = VariableStore()
    k = st.create("dense_1/kernel", (32, 64), "he_uniform")
    bia = st.create("dense_1/bias", (64,), "zeros")
    st.finalize("cpu", seed=3, device_init=True)
    assert st.device_initialised
    assert k.tensor.abs().max() <= math.sqrt(6.0 / 32) and k.tensor.std() > 0.2 and float(bia.tensor.abs().sum()) == 0.0
    st2 = VariableStore()
    k2 = st2.create("dense_1/kernel", (32, 64), "he_uniform")
    st2.create("dense_1/bias", (64,), "zeros")
    st2.finalize("cpu", seed=3, device_init=True)
    assert torch.equal(k.tensor, k2.tensor)

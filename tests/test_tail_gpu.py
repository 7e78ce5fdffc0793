"""Fused network-tail kernel (csrc/kernels/mlp_tail.hip) against the chain of layer-wise HIP kernels and against
PyTorch autograd on the CPU."""
import pytest
import torch

from distributed_tensorflow_amd.ops import functional as F
from distributed_tensorflow_amd.ops._native import ACT_NONE, ACT_RELU

pytestmark = pytest.mark.gpu


def _inputs(M, C, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    h1 = torch.relu(torch.randn(M, 128, generator=g))            # a first layer's ReLU output (zeros included)
    w2 = torch.randn(128, 128, generator=g) * 0.1
    b2 = torch.randn(128, generator=g) * 0.1
    w3 = torch.randn(128, C, generator=g) * 0.1
    b3 = torch.randn(C, generator=g) * 0.1
    y = torch.randint(0, C, (M,), generator=g)
    return [t.to(dev) for t in (h1, w2, b2, w3, b3, y)]


def _run_tail(h1, w2, b2, w3, b3, y, keep2, seed2, step, scale1):
    dev = h1.device
    M, C = h1.shape[0], w3.shape[1]
    dz1 = torch.full((M, 128), float("nan"), device=dev)
    dw2, db2 = torch.full((128, 128), float("nan"), device=dev), torch.full((128,), float("nan"), device=dev)
    dw3, db3 = torch.full((128, C), float("nan"), device=dev), torch.full((C,), float("nan"), device=dev)
    ws = torch.empty(F.mlp_tail_workspace_floats(M, C), device=dev)
    stats = F.new_stats(dev)
    F.mlp_tail(h1, w2, b2, w3, b3, y, dz1, dw2, db2, dw3, db3, ws, keep2, seed2, step, scale1, 1.0 / M, stats)
    return dz1, dw2, db2, dw3, db3, stats


def _run_layerwise(h1, w2, b2, w3, b3, y, keep2, seed2, step, scale1):
    dev = h1.device
    M, C = h1.shape[0], w3.shape[1]
    h2 = F.dense_forward(h1, w2, b2, ACT_RELU, keep2, seed2, step)
    z = F.dense_forward(h2, w3, b3, ACT_NONE)
    stats = F.new_stats(dev)
    dz3 = torch.empty_like(z)
    F.softmax_xent(z, y, dz3, 1.0 / M, stats)
    dw3, db3 = torch.empty(128, C, device=dev), torch.empty(C, device=dev)
    F.dense_backward_dw(h2, dz3, dw3, db3)
    dz2 = F.dense_backward_dx(dz3, w3, h2, ACT_RELU, 1.0 / keep2)
    dw2, db2 = torch.empty(128, 128, device=dev), torch.empty(128, device=dev)
    F.dense_backward_dw(h1, dz2, dw2, db2)
    dz1 = F.dense_backward_dx(dz2, w2, h1, ACT_RELU, scale1)
    return dz1, dw2, db2, dw3, db3, stats


@pytest.mark.parametrize("M,C,keep2", [(64, 10, 1.0), (50, 10, 0.7), (1000, 10, 0.7), (32768, 10, 0.7), (4097, 16, 0.5),
                                       (300, 3, 1.0), (40000, 10, 0.7)])
def test_tail_matches_layerwise_kernels(gpu_device, M, C, keep2):
    args = _inputs(M, C, gpu_device)
    step = torch.tensor([7], dtype=torch.int64, device=gpu_device)
    got = _run_tail(*args, keep2, 1234, step, 1.0 / 0.7)
    ref = _run_layerwise(*args, keep2, 1234, step, 1.0 / 0.7)
    names = ["dz1", "dw2", "db2", "dw3", "db3"]
    for n, a, b in zip(names, got[:5], ref[:5]):
        assert torch.isfinite(a).all(), n
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-6, msg=lambda m, n=n: "%s: %s" % (n, m))
    gs, rs = got[5].tolist(), ref[5].tolist()
    assert gs[2] == rs[2] == M
    assert gs[0] == pytest.approx(rs[0], rel=1e-4) and gs[1] == rs[1]


def test_tail_matches_autograd_without_dropout(gpu_device):
    M, C = 200, 10
    args = _inputs(M, C, gpu_device, seed=3)
    step = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    dz1, dw2, db2, dw3, db3, stats = _run_tail(*args, 1.0, 0, step, 1.0)
    h1, w2, b2, w3, b3, y = [t.cpu() for t in args]
    pre = torch.randn(M, 128, generator=torch.Generator().manual_seed(9))
    h1 = h1.clone().requires_grad_(True)
    w2, b2, w3, b3 = [t.clone().requires_grad_(True) for t in (w2, b2, w3, b3)]
    z = torch.relu(h1 @ w2 + b2) @ w3 + b3
    loss = torch.nn.functional.cross_entropy(z, y)
    loss.backward()
    torch.testing.assert_close(dw2.cpu(), w2.grad, rtol=1e-3, atol=1e-6)
    torch.testing.assert_close(db2.cpu(), b2.grad, rtol=1e-3, atol=1e-6)
    torch.testing.assert_close(dw3.cpu(), w3.grad, rtol=1e-3, atol=1e-6)
    torch.testing.assert_close(db3.cpu(), b3.grad, rtol=1e-3, atol=1e-6)
    # dz1 includes the first layer's ReLU gradient: positions where h1 == 0 are zero
    torch.testing.assert_close(dz1.cpu(), h1.grad * (h1 > 0).float(), rtol=1e-3, atol=1e-7)
    assert stats[0].item() / M == pytest.approx(loss.item(), rel=1e-4)


def test_model_training_with_and_without_tail_kernel(gpu_device):
    from distributed_tensorflow_amd.data import mnist_like
    from distributed_tensorflow_amd.models import mnist_mlp, xor_mlp
    x, y = mnist_like(8 * 600, seed=0, device=gpu_device)
    ms = []
    for use in (True, False):
        m = mnist_mlp()
        m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device=gpu_device, seed=0)
        m.use_tail_kernel = use
        for i, l in enumerate(m.layers):
            if hasattr(l, "seed"):
                l.seed = 900 + i
        ms.append(m)
    a, b = ms
    assert a.tail_plan() is not None and b.tail_plan() is None
    for i in range(8):
        for m in ms:
            m.train_step(x[i * 600:(i + 1) * 600], y[i * 600:(i + 1) * 600])
    torch.testing.assert_close(a.store.flat, b.store.flat, rtol=1e-3, atol=2e-6)
    la, lb = a.read_stats(), b.read_stats()
    assert la["loss"] == pytest.approx(lb["loss"], rel=1e-4) and la["acc"] == pytest.approx(lb["acc"], abs=1e-3)
    # graph capture of a step that contains the tail kernel
    g = a._graphed_steps(x[:1200], y[:1200], 600, 2)
    before = a.global_step
    g.replay()
    torch.cuda.synchronize()
    assert a.global_step == before + 2
    # models that do not match keep the layer-wise path
    other = xor_mlp()
    other.compile(loss="mean_squared_error", optimizer="adam", device=gpu_device, seed=0)
    assert other.tail_plan() is None

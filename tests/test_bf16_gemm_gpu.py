"""bf16-operand MFMA products for wide layers (opt-in mixed precision): the HIP kernels against a plain PyTorch
reference that rounds the operands to bf16 and multiplies / accumulates in fp32 (fp64 for the check)."""
import pytest
import torch

from distributed_tensorflow_amd.ops import functional as F

pytestmark = pytest.mark.gpu


def _r(t):
    return t.to(torch.bfloat16).double()


@pytest.mark.parametrize("M,K,N", [(2048, 2048, 2048), (1536, 2052, 2300)])       # whole tiles, and ragged edges
def test_three_products_match_the_rounded_operand_reference(gpu_device, M, K, N):
    assert F.dense_bf16_supported(M, N, K)
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(M, K, generator=g) - 0.5).to(gpu_device)
    w = ((torch.rand(K, N, generator=g) - 0.5) * 0.1).to(gpu_device)
    b = torch.rand(N, generator=g).to(gpu_device)
    dz = (torch.rand(M, N, generator=g) - 0.5).to(gpu_device)

    y = F.dense_forward(x, w, b, 1, bf16=True)
    ref = torch.relu(_r(x) @ _r(w) + b.double())
    torch.testing.assert_close(y.double(), ref, rtol=1e-4, atol=1e-4)
    # it really is the reduced-precision product: the exact fp32 result differs by the operand rounding
    exact = torch.relu(x.double() @ w.double() + b.double())
    assert (y.double() - exact).abs().max() > 10 * (y.double() - ref).abs().max()

    dx = F.dense_backward_dx(dz, w, yprev=x, act_prev=1, scale_prev=1.25, bf16=True)
    refx = torch.where(x > 0, (_r(dz) @ _r(w).t()) * 1.25, torch.zeros((), dtype=torch.float64, device=gpu_device))
    torch.testing.assert_close(dx.double(), refx, rtol=1e-4, atol=1e-4)

    dw = torch.empty(K, N, device=gpu_device)
    db = torch.empty(N, device=gpu_device)
    F.dense_backward_dw(x, dz, dw, db, bf16=True)
    torch.testing.assert_close(dw.double(), _r(x).t() @ _r(dz), rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(db.double(), dz.double().sum(0), rtol=1e-5, atol=1e-4)      # bias gradient stays fp32


def test_dropout_mask_is_the_fp32_kernel_mask(gpu_device):
    M, K, N = 2048, 2048, 2048
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(M, K, generator=g) - 0.5).to(gpu_device)
    w = ((torch.rand(K, N, generator=g) - 0.5) * 0.1).to(gpu_device)
    step = torch.tensor([11], dtype=torch.int64, device=gpu_device)
    a = F.dense_forward(x, w, None, 0, keep=0.7, seed=5, step=step, bf16=True)
    c = F.dense_forward(x, w, None, 0, keep=0.7, seed=5, step=step)
    assert torch.equal(a == 0, c == 0)


def test_small_layers_are_refused(gpu_device):
    assert not F.dense_bf16_supported(50, 128, 64)
    x = torch.zeros(50, 64, device=gpu_device)
    w = torch.zeros(64, 128, device=gpu_device)
    with pytest.raises(Exception):
        F.dense_forward(x, w, None, 0, bf16=True)


def test_wide_model_trains_in_mixed_precision(gpu_device):
    """compile(matmul_dtype="bf16"): the wide layers run on the bf16 kernels, gradients stay close to the fp32 run's
    and the loss goes down."""
    from distributed_tensorflow_amd import models
    from distributed_tensorflow_amd.data import mnist_like
    x, y = mnist_like(2048, seed=0, device=gpu_device)
    grads, losses = {}, {}
    for dt in ("fp32", "bf16"):
        m = models.wide_mlp(width=2048, depth=3)
        m.compile(loss="sparse_categorical_crossentropy", optimizer="sgd", metrics=["accuracy"], device=gpu_device, seed=4,
                  matmul_dtype=dt)
        m.stats.zero_()
        m.train_step(x, y)
        grads[dt] = m.store.flat_grad.clone()
        used = [getattr(l, "_bf16_used", False) for l in m.layers if hasattr(l, "units")]
        assert any(used) == (dt == "bf16") and not used[-1]          # the 10-unit head is never a bf16 product
        ls = []
        for _ in range(15):
            m.stats.zero_()
            m.train_step(x, y)
            ls.append(m.read_stats()["loss"])
        losses[dt] = ls
    cos = torch.nn.functional.cosine_similarity(grads["fp32"], grads["bf16"], dim=0)
    assert cos > 0.995, cos
    assert not torch.equal(grads["fp32"], grads["bf16"])
    assert losses["bf16"][-1] < losses["bf16"][0]
    assert abs(losses["bf16"][-1] - losses["fp32"][-1]) < 0.1 * abs(losses["fp32"][0])

"""fp32 dense-layer products on the bf16 matrix cores (three exact bf16 limbs per operand, 6 / 8 / 9 limb products
accumulated in fp32): the HIP kernel against a plain PyTorch fp64 product of the same fp32 operands, held to the error
of the native fp32 MFMA kernel, and a training run against the fp32 run."""
import pytest
import torch

from distributed_tensorflow_amd.models import zoo
from distributed_tensorflow_amd.ops import functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture
def split_mode():
    yield F.set_gemm_split_products
    F.set_gemm_split_products(0)


def _three_products(x, w, b, dz):
    M, K = x.shape
    N = w.shape[1]
    y = F.dense_forward(x, w, b, 1)
    dx = F.dense_backward_dx(dz, w, yprev=x, act_prev=1, scale_prev=1.25)
    dw = torch.empty(K, N, device=x.device)
    db = torch.empty(N, device=x.device)
    ws = torch.empty(max(F.dense_dw_workspace_floats(M, N, K), 1), device=x.device)
    F.dense_backward_dw(x, dz, dw, db, workspace=ws)
    return y.double(), dx.double(), dw.double(), db.double()


def _rel_err(a, ref):
    return ((a - ref).abs().max() / ref.abs().max()).item()


# whole tiles; ragged M and K (multiples of 4, as the kernel requires) with split-batch wgrad; the headline first layer
@pytest.mark.parametrize("M,K,N", [(2048, 1024, 1024), (4100, 788, 516), (8192, 784, 128)])
@pytest.mark.parametrize("products", [6, 8, 9])
def test_split_products_are_as_accurate_as_the_fp32_mfma(gpu_device, split_mode, M, K, N, products):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, K, generator=g).to(gpu_device)
    w = (torch.randn(K, N, generator=g) * 0.05).to(gpu_device)
    b = torch.randn(N, generator=g).to(gpu_device)
    dz = torch.randn(M, N, generator=g).to(gpu_device)
    refs = (torch.relu(x.double() @ w.double() + b.double()),
            torch.where(x > 0, (dz.double() @ w.double().t()) * 1.25, torch.zeros((), dtype=torch.float64, device=gpu_device)),
            x.double().t() @ dz.double(),
            dz.double().sum(0))
    split_mode(0)
    native = _three_products(x, w, b, dz)
    split_mode(products)
    assert F.gemm_split_products() == products
    split = _three_products(x, w, b, dz)
    for name, s, n, r in zip(("fwd", "dx", "dw", "db"), split, native, refs):
        es, en = _rel_err(s, r), _rel_err(n, r)
        assert es < 5e-6, (name, es)
        assert es <= 1.5 * en + 1e-7, (name, es, en)          # no worse than the fp32 matrix-core kernel
    # it is a different kernel, not a silent fallback: the rounding order differs from the fp32 MFMA's
    assert not torch.equal(split[0], native[0])


def test_shapes_the_split_kernel_cannot_take_fall_back_to_fp32_mfma(gpu_device, split_mode):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(512, 130, generator=g).to(gpu_device)        # K not a multiple of 4
    w = torch.randn(130, 10, generator=g).to(gpu_device)          # 10 output columns
    split_mode(0)
    a = F.dense_forward(x, w, None, 0)
    split_mode(6)
    c = F.dense_forward(x, w, None, 0)
    assert torch.equal(a, c)


def test_training_with_split_products_tracks_the_fp32_run(gpu_device, split_mode):
    losses = {}
    for mode in ("fp32", "fp32_split"):
        torch.manual_seed(0)
        model = zoo.mnist_mlp(dropout=0.0)
        model.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device=gpu_device,
                      seed=3, matmul_dtype=mode)
        g = torch.Generator().manual_seed(1)
        x = torch.rand(8192, 784, generator=g).to(gpu_device)
        y = torch.randint(0, 10, (8192,), generator=g).to(gpu_device)
        hist = model.fit(x, y, batch_size=4096, epochs=6, shuffle=False, verbose=0)
        losses[mode] = hist.history["loss"]
        assert F.gemm_split_products() == (6 if mode == "fp32_split" else 0)
    for a, c in zip(losses["fp32"], losses["fp32_split"]):
        assert abs(a - c) < 2e-4 * max(1.0, abs(a)), losses
    assert losses["fp32_split"][-1] < losses["fp32_split"][0]

"""HIP kernel numerics against plain PyTorch fp32/fp64 references (needs an MI355X)."""
import pytest
import torch

from distributed_tensorflow_amd.ops import functional as F
from distributed_tensorflow_amd.ops._native import ACT_NONE, ACT_RELU, ACT_SIGMOID

pytestmark = pytest.mark.gpu

SHAPES = [  # (M, K, N)
    (50, 64, 128), (50, 128, 128), (50, 128, 32), (1000, 64, 128), (1000, 128, 32),
    (256, 784, 128), (4096, 784, 128), (4096, 128, 10), (333, 100, 77), (1, 5, 3), (129, 33, 65),
    (8192, 128, 128), (70, 784, 10),
]


def _rand(*shape, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1).to(device)


def _tol(K):
    return dict(rtol=2e-4, atol=2e-5 * max(K, 16) ** 0.5)


@pytest.mark.parametrize("M,K,N", SHAPES)
@pytest.mark.parametrize("act", [ACT_NONE, ACT_RELU, ACT_SIGMOID])
def test_dense_forward(gpu_device, M, K, N, act):
    x, w, b = _rand(M, K, device=gpu_device, seed=1), _rand(K, N, device=gpu_device, seed=2), _rand(N, device=gpu_device, seed=3)
    y = F.dense_forward(x, w, b, act)
    ref = x.double() @ w.double() + b.double()
    ref = torch.relu(ref) if act == ACT_RELU else torch.sigmoid(ref) if act == ACT_SIGMOID else ref
    torch.testing.assert_close(y, ref.float(), **_tol(K))


def test_dense_forward_identity_asymmetric(gpu_device):
    """A = I with an asymmetric B catches a transposed C-write or a swapped operand map."""
    n = 128
    eye = torch.eye(n, device=gpu_device)
    b = torch.arange(n * n, dtype=torch.float32, device=gpu_device).view(n, n) / 1000.0
    torch.testing.assert_close(F.dense_forward(eye, b, None, ACT_NONE), b)
    torch.testing.assert_close(F.dense_forward(b, eye, None, ACT_NONE), b)


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_dense_backward_dx(gpu_device, M, K, N):
    dz, w = _rand(M, N, device=gpu_device, seed=4), _rand(K, N, device=gpu_device, seed=5)
    yprev = _rand(M, K, device=gpu_device, seed=6)
    dx = F.dense_backward_dx(dz, w)
    ref = (dz.double() @ w.double().t())
    torch.testing.assert_close(dx, ref.float(), **_tol(N))
    dx2 = F.dense_backward_dx(dz, w, yprev, ACT_RELU, 1.0 / 0.7)
    ref2 = torch.where(yprev > 0, ref / 0.7, torch.zeros_like(ref))
    torch.testing.assert_close(dx2, ref2.float(), **_tol(N))
    p = torch.sigmoid(yprev)
    dx3 = F.dense_backward_dx(dz, w, p, ACT_SIGMOID, 1.0)
    torch.testing.assert_close(dx3, (ref * p.double() * (1 - p.double())).float(), **_tol(N))


@pytest.mark.parametrize("M,K,N", SHAPES + [(65536, 784, 128), (20000, 128, 128)])
def test_dense_backward_dw(gpu_device, M, K, N):
    x, dz = _rand(M, K, device=gpu_device, seed=7), _rand(M, N, device=gpu_device, seed=8)
    dw = torch.full((K, N), float("nan"), device=gpu_device)
    db = torch.full((N,), float("nan"), device=gpu_device)
    F.dense_backward_dw(x, dz, dw, db)
    tol = dict(rtol=2e-4, atol=3e-5 * max(M, 16) ** 0.5)
    torch.testing.assert_close(dw, (x.double().t() @ dz.double()).float(), **tol)
    torch.testing.assert_close(db, dz.double().sum(0).float(), **tol)


@pytest.mark.parametrize("M,K,N", [(65536, 784, 128), (32768, 800, 64), (40000, 1056, 256), (30000, 200, 128)])
def test_dense_backward_dw_partly_filled_last_row_tile_and_dirty_workspace(gpu_device, M, K, N):
    """Weight gradients whose last row tile is at most half full (784 = 12 x 64 + 16, 800 = 12 x 64 + 32, ...), computed
    in a caller-provided workspace that holds NaN: every slab element the reducer reads was written by a block."""
    x, dz = _rand(M, K, device=gpu_device, seed=17), _rand(M, N, device=gpu_device, seed=18)
    dw = torch.full((K, N), float("nan"), device=gpu_device)
    db = torch.full((N,), float("nan"), device=gpu_device)
    ws = torch.full((max(F.dense_dw_workspace_floats(M, N, K), 1),), float("nan"), device=gpu_device)
    F.dense_backward_dw(x, dz, dw, db, workspace=ws)
    tol = dict(rtol=2e-4, atol=3e-5 * max(M, 16) ** 0.5)
    torch.testing.assert_close(dw, (x.double().t() @ dz.double()).float(), **tol)
    torch.testing.assert_close(db, dz.double().sum(0).float(), **tol)


def test_fused_dropout_matches_cpu_mask(gpu_device):
    M, K, N = 300, 64, 128
    x, w, b = _rand(M, K, device=gpu_device, seed=1), _rand(K, N, device=gpu_device, seed=2), _rand(N, device=gpu_device, seed=3)
    step = torch.tensor([41], dtype=torch.int64, device=gpu_device)
    y = F.dense_forward(x, w, b, ACT_RELU, keep=0.7, seed=99, step=step)
    ref = F.dense_forward(x.cpu(), w.cpu(), b.cpu(), ACT_RELU, keep=0.7, seed=99, step=step.cpu())
    torch.testing.assert_close(y.cpu(), ref, rtol=2e-4, atol=1e-4)
    frac = (y == 0).float().mean().item()
    assert 0.55 < frac < 0.75  # relu kills ~half, dropout 30% of the rest
    step += 1
    y2 = F.dense_forward(x, w, b, ACT_RELU, keep=0.7, seed=99, step=step)
    assert not torch.equal(y, y2)


def test_standalone_dropout(gpu_device):
    x = _rand(1000, 37, device=gpu_device, seed=1) + 2.0
    step = torch.tensor([7], dtype=torch.int64, device=gpu_device)
    y = F.dropout(x, 0.7, 5, step)
    ref = F.dropout(x.cpu(), 0.7, 5, step.cpu())
    torch.testing.assert_close(y.cpu(), ref)
    assert abs((y != 0).float().mean().item() - 0.7) < 0.02


def test_act_grad_and_activation(gpu_device):
    dy, y = _rand(513, 77, device=gpu_device, seed=1), _rand(513, 77, device=gpu_device, seed=2)
    for act in (ACT_RELU, ACT_SIGMOID, ACT_NONE):
        torch.testing.assert_close(F.act_grad(dy, y, act, 1.3).cpu(), F.act_grad(dy.cpu(), y.cpu(), act, 1.3))
        torch.testing.assert_close(F.activation_forward(y, act).cpu(), F.activation_forward(y.cpu(), act),
                                   rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,N", [(50, 32), (1000, 32), (4097, 10)])
@pytest.mark.parametrize("through", [False, True])
def test_mse_loss(gpu_device, M, N, through):
    p = torch.sigmoid(_rand(M, N, device=gpu_device, seed=1) * 3)
    y = (_rand(M, N, device=gpu_device, seed=2) > 0).float()
    g, st = torch.empty_like(p), F.new_stats(gpu_device)
    F.mse_loss(p, y, g, None, through, st)
    gc, sc = torch.empty(M, N), F.new_stats("cpu")
    F.mse_loss(p.cpu(), y.cpu(), gc, None, through, sc)
    torch.testing.assert_close(g.cpu(), gc, rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(st.cpu(), sc, rtol=1e-4, atol=1e-4)
    # autograd cross-check of the fused gradient
    pz = p.cpu().double().clone().requires_grad_(True)
    loss = ((pz - y.cpu().double()) ** 2).mean()
    loss.backward()
    ref = pz.grad * (p.cpu().double() * (1 - p.cpu().double()) if through else 1.0)
    torch.testing.assert_close(g.cpu().double(), ref, rtol=1e-4, atol=1e-10)


def test_round_half_to_even(gpu_device):
    p = torch.tensor([[0.5, 0.5, 0.50001, 0.49999]], device=gpu_device)
    y = torch.tensor([[0.0, 1.0, 1.0, 0.0]], device=gpu_device)
    st = F.new_stats(gpu_device)
    F.mse_loss(p, y, None, None, False, st)
    assert abs(st[1].item() - 0.75) < 1e-6  # round(0.5) == 0 (half to even), as tf.round


@pytest.mark.parametrize("M,C", [(50, 10), (4096, 10), (1000, 37)])
@pytest.mark.parametrize("sparse", [True, False])
def test_softmax_xent(gpu_device, M, C, sparse):
    z = _rand(M, C, device=gpu_device, seed=1) * 4
    lab = torch.randint(0, C, (M,), generator=torch.Generator().manual_seed(3)).to(gpu_device)
    labels = lab if sparse else torch.nn.functional.one_hot(lab, C).float()
    g, st = torch.empty_like(z), F.new_stats(gpu_device)
    F.softmax_xent(z, labels, g, None, st)
    zd = z.cpu().double().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(zd, lab.cpu(), reduction="mean")
    loss.backward()
    torch.testing.assert_close(g.cpu().double(), zd.grad, rtol=1e-4, atol=1e-8)
    assert abs(st[0].item() / M - loss.item()) < 1e-4 * max(1.0, loss.item())
    assert abs(st[1].item() - (z.argmax(1) == lab).sum().item()) < 0.5
    assert st[2].item() == M
    torch.testing.assert_close(F.softmax(z).cpu(), torch.softmax(z.cpu(), 1), rtol=1e-5, atol=1e-7)


def test_adam_tf_formula(gpu_device):
    n = 10007
    var, g = _rand(n, device=gpu_device, seed=1), _rand(n, device=gpu_device, seed=2)
    m, v = torch.zeros(n, device=gpu_device), torch.zeros(n, device=gpu_device)
    step = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    pw = torch.ones(2, device=gpu_device)
    vr, mr, vvr = var.cpu().double(), torch.zeros(n, dtype=torch.double), torch.zeros(n, dtype=torch.double)
    for t in range(1, 6):
        F.adam_apply(var, m, v, g, step)
        F.step_advance(step, pw, 0.9, 0.999)
        gd = g.cpu().double()
        lr_t = 1e-3 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t)
        mr = 0.9 * mr + 0.1 * gd
        vvr = 0.999 * vvr + 0.001 * gd * gd
        vr = vr - lr_t * mr / (vvr.sqrt() + 1e-8)
    assert step.item() == 5
    torch.testing.assert_close(var.cpu().double(), vr, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pw.cpu(), torch.tensor([0.9 ** 5, 0.999 ** 5]), rtol=1e-5, atol=0)


def test_sgd(gpu_device):
    n = 5003
    var, g = _rand(n, device=gpu_device, seed=1), _rand(n, device=gpu_device, seed=2)
    mom = torch.zeros(n, device=gpu_device)
    ref, refm = var.clone().cpu(), torch.zeros(n)
    for _ in range(3):
        F.sgd_apply(var, g, 0.1, mom, 0.9, 0.5)
        F.sgd_apply(ref, g.cpu(), 0.1, refm, 0.9, 0.5)
    torch.testing.assert_close(var.cpu(), ref, rtol=1e-5, atol=1e-6)
    v2, r2 = var.clone(), var.clone().cpu()
    F.sgd_apply(v2, g, 0.1)
    F.sgd_apply(r2, g.cpu(), 0.1)
    torch.testing.assert_close(v2.cpu(), r2, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("inside", [True, False])
def test_rmsprop_and_adagrad_kernels(gpu_device, inside):
    """The accumulator rules on the GPU against the fp32 PyTorch reference of the same op (odd length: scalar tail),
    with and without the momentum slot, epsilon inside / outside the root, and the fused global_step increment."""
    n = 10007
    g0 = torch.Generator().manual_seed(5)
    var = torch.randn(n, generator=g0)
    step = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    for mom_on in (True, False):
        v, ms, mo = var.to(gpu_device), torch.ones(n, device=gpu_device), (torch.zeros(n, device=gpu_device) if mom_on else None)
        r, rms, rmo = var.clone(), torch.ones(n), (torch.zeros(n) if mom_on else None)
        for t in range(4):
            g = torch.randn(n, generator=g0)
            F.rmsprop_apply(v, ms, g.to(gpu_device), 0.01, 0.9, mo, 0.7, 1e-7, inside, grad_scale=0.5, advance=step)
            F.rmsprop_apply(r, rms, g, 0.01, 0.9, rmo, 0.7, 1e-7, inside, grad_scale=0.5)
        torch.testing.assert_close(v.cpu(), r, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(ms.cpu(), rms, rtol=1e-5, atol=1e-7)
        if mom_on:
            torch.testing.assert_close(mo.cpu(), rmo, rtol=1e-5, atol=1e-7)
    assert step.item() == 8
    v, acc, r, racc = var.to(gpu_device), torch.full((n,), 0.1, device=gpu_device), var.clone(), torch.full((n,), 0.1)
    for t in range(4):
        g = torch.randn(n, generator=g0)
        F.adagrad_apply(v, acc, g.to(gpu_device), 0.05, 1e-8, inside, advance=step)
        F.adagrad_apply(r, racc, g, 0.05, 1e-8, inside)
    torch.testing.assert_close(v.cpu(), r, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(acc.cpu(), racc, rtol=1e-6, atol=1e-7)
    assert step.item() == 12
    v, a, u = var.to(gpu_device), torch.zeros(n, device=gpu_device), torch.zeros(n, device=gpu_device)
    r, ra, ru = var.clone(), torch.zeros(n), torch.zeros(n)
    for t in range(4):
        g = torch.randn(n, generator=g0)
        F.adadelta_apply(v, a, u, g.to(gpu_device), 1.0, 0.95, 1e-8, advance=step)
        F.adadelta_apply(r, ra, ru, g, 1.0, 0.95, 1e-8)
    torch.testing.assert_close(v.cpu(), r, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(u.cpu(), ru, rtol=1e-4, atol=1e-9)
    assert step.item() == 16


def test_xor_generator_matches_cpu(gpu_device):
    x, y = F.xor_data(5000, 32, 77, gpu_device)
    xc, yc = F.xor_data(5000, 32, 77, "cpu")
    assert torch.equal(x.cpu(), xc) and torch.equal(y.cpu(), yc)
    assert torch.equal(y, (x[:, :32] != x[:, 32:]).float())
    assert abs(x.mean().item() - 0.5) < 0.01


def test_dense_forward_splits_a_long_k_with_few_output_tiles(gpu_device):
    """1024 x 3136 -> 512 (the CNN's first dense layer): K is split into slabs, bias and ReLU follow in the reducer."""
    M, K, N = 1024, 3136, 512
    need = F.dense_fwd_workspace_floats(M, N, K)
    assert need >= 2 * M * N
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(M, K, generator=g) - 0.5).to(gpu_device)
    w = ((torch.rand(K, N, generator=g) - 0.5) * 0.1).to(gpu_device)
    b = torch.rand(N, generator=g).to(gpu_device)
    ws = torch.empty(need, device=gpu_device)
    for act in (0, 1, 2):
        y = F.dense_forward(x, w, b, act, workspace=ws)
        ref = x.double() @ w.double() + b.double()
        ref = torch.relu(ref) if act == 1 else (torch.sigmoid(ref) if act == 2 else ref)
        torch.testing.assert_close(y.double(), ref, rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(y, F.dense_forward(x, w, b, act), rtol=1e-5, atol=1e-5)     # unsplit path agrees
    step = torch.tensor([7], dtype=torch.int64, device=gpu_device)
    yd = F.dense_forward(x, w, b, 1, keep=0.7, seed=31, step=step, workspace=ws)
    torch.testing.assert_close(yd, F.dense_forward(x, w, b, 1, keep=0.7, seed=31, step=step), rtol=1e-5, atol=1e-5)
    assert 0.25 < (yd == 0).float().mean() < 0.9                       # ReLU zeros plus 30 % dropped
    assert F.dense_fwd_workspace_floats(65536, 128, 784) == 0          # plenty of tiles: never split


def test_device_initialiser_matches_the_cpu_generator():
    """``init_uniform_``: the HIP kernel and the CPU implementation of the counter-based generator agree bit for bit
    (also past element 2^32 is covered by the same formula; here a range with an odd length)."""
    from distributed_tensorflow_amd.ops import functional as F
    n = 1_000_003
    g = F.init_uniform_(torch.empty(n, device="cuda"), 12345, -0.037, 0.037)
    c = F.init_uniform_(torch.empty(n), 12345, -0.037, 0.037)
    assert torch.equal(g.cpu(), c)
    assert float(g.min()) >= -0.037 and float(g.max()) < 0.037 and abs(float(g.mean())) < 1e-3

"""The ``torch.autograd.Function`` wrappers of ``ops/autograd.py`` against fp64 PyTorch autograd of the same math:
values and every gradient.  CPU runs check the wiring with the reference implementations; the GPU runs are the
gradient check of the HIP kernels under PyTorch's reverse-mode engine (arbitrary graphs: a residual block, two
summed losses, an ordinary ``torch.nn.Module`` trained with ``torch.optim``)."""
import pytest
import torch

from distributed_tensorflow_amd.ops import autograd as A

DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def _ref(t):
    return t.detach().double().cpu().requires_grad_(t.requires_grad)


def _close(a, b, tol=2e-4):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = max(1.0, float(b.abs().max()))
    assert float((a - b).abs().max()) <= tol * scale, (float((a - b).abs().max()), scale)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("act", [None, "relu", "sigmoid"])
@pytest.mark.parametrize("shape", [(50, 64, 128), (130, 100, 36)])
def test_dense_values_and_gradients(device, act, shape):
    M, K, N = shape
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(device).requires_grad_()
    w = (torch.randn(K, N, generator=g) * 0.2).to(device).requires_grad_()      # asymmetric operand
    b = torch.randn(N, generator=g).to(device).requires_grad_()
    up = torch.randn(M, N, generator=g).to(device)
    y = A.dense(x, w, b, act)
    (y * up).sum().backward()
    xr, wr, br = _ref(x), _ref(w), _ref(b)
    z = xr @ wr + br
    yr = {None: z, "relu": torch.relu(z), "sigmoid": torch.sigmoid(z)}[act]
    (yr * up.double().cpu()).sum().backward()
    _close(y, yr)
    _close(x.grad, xr.grad)
    _close(w.grad, wr.grad)
    _close(b.grad, br.grad)


@pytest.mark.parametrize("device", DEVICES)
def test_dense_with_fused_dropout_uses_one_mask_both_ways(device):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 32, generator=g).to(device).requires_grad_()
    w = (torch.randn(32, 48, generator=g) * 0.3).to(device).requires_grad_()
    step = torch.zeros(1, dtype=torch.int64, device=device)
    y = A.dense(x, w, None, "relu", dropout=0.3, seed=11, step=step)
    kept = (y != 0)
    plain = A.dense(x.detach(), w.detach(), None, "relu")
    # surviving activations are scaled by 1 / keep; about 30 % of the positive ones are dropped
    _close(y[kept], plain[kept] / 0.7)
    pos = plain > 0
    assert 0.2 < float((pos & ~kept).sum()) / float(pos.sum()) < 0.4
    y.sum().backward()
    # gradient w.r.t. x: only the kept, positive units pass, scaled the same way
    expect = (kept.float() / 0.7) @ w.detach().t()
    _close(x.grad, expect)


@pytest.mark.parametrize("device", DEVICES)
def test_losses(device):
    g = torch.Generator().manual_seed(5)
    z = torch.randn(40, 10, generator=g).to(device).requires_grad_()
    labels = torch.randint(0, 10, (40,), generator=g).to(device)
    loss = A.softmax_cross_entropy(z, labels)
    (loss * 3.0).backward()
    zr = _ref(z)
    lr = torch.nn.functional.cross_entropy(zr, labels.cpu())
    (lr * 3.0).backward()
    _close(loss, lr)
    _close(z.grad, zr.grad)
    p = torch.rand(40, 32, generator=g).to(device).requires_grad_()
    y = (torch.rand(40, 32, generator=g) > 0.5).float().to(device)
    m = A.mean_squared_error(p, y)
    m.backward()
    pr = _ref(p)
    mr = ((pr - y.double().cpu()) ** 2).mean()
    mr.backward()
    _close(m, mr)
    _close(p.grad, pr.grad)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("cfg", [dict(C=1, N=8, k=5, s=(1, 1), pad="same"), dict(C=4, N=12, k=3, s=(2, 2), pad="valid"),
                                 dict(C=3, N=6, k=3, s=(1, 1), pad="valid")])
def test_conv2d_and_pooling(device, cfg):
    g = torch.Generator().manual_seed(cfg["k"] + cfg["C"])
    B, H, W = 3, 12, 12
    x = torch.randn(B, H, W, cfg["C"], generator=g).to(device).requires_grad_()
    w = (torch.randn(cfg["k"], cfg["k"], cfg["C"], cfg["N"], generator=g) * 0.3).to(device).requires_grad_()
    b = torch.randn(cfg["N"], generator=g).to(device).requires_grad_()
    y = A.conv2d(x, w, b, cfg["s"], cfg["pad"], "relu")
    pooled = A.max_pool2d(y, (2, 2)) if y.shape[1] % 2 == 0 and y.shape[2] % 2 == 0 else y
    up = torch.randn(pooled.shape, generator=g).to(device)
    (pooled * up).sum().backward()
    xr, wr, br = _ref(x), _ref(w), _ref(b)
    xin = xr.permute(0, 3, 1, 2)
    if cfg["pad"] == "same":      # TensorFlow's rule: out = ceil(in / stride), the extra padding goes bottom / right
        oh = -(-H // cfg["s"][0])
        total = max((oh - 1) * cfg["s"][0] + cfg["k"] - H, 0)
        lo, hi = total // 2, total - total // 2
        xin = torch.nn.functional.pad(xin, (lo, hi, lo, hi))
    yr = torch.relu(torch.nn.functional.conv2d(xin, wr.permute(3, 2, 0, 1), br, stride=cfg["s"]))
    pr = torch.nn.functional.max_pool2d(yr, 2) if pooled is not y else yr
    pr = pr.permute(0, 2, 3, 1)
    (pr * up.double().cpu()).sum().backward()
    _close(pooled, pr)
    _close(x.grad, xr.grad)
    _close(w.grad, wr.grad)
    _close(b.grad, br.grad)


class _ResidualNet(torch.nn.Module):
    """Two inputs, a skip connection, two heads with summed losses: nothing a sequential chain can express."""

    def __init__(self):
        super().__init__()
        self.inp_a = A.Dense(24, 32, "relu")
        self.inp_b = A.Dense(8, 32, "relu")
        self.block = A.Dense(32, 32, "relu")
        self.head_cls = A.Dense(32, 5)
        self.head_reg = A.Dense(32, 3, "sigmoid")

    def forward(self, a, b):
        h = self.inp_a(a) + self.inp_b(b)
        h = h + self.block(h)                      # residual add
        return self.head_cls(h), self.head_reg(h)


@pytest.mark.parametrize("device", DEVICES)
def test_ordinary_module_with_skip_connection_trains_and_matches_fp64(device):
    torch.manual_seed(0)
    net = _ResidualNet().to(device)
    ref = _ResidualNet().double()
    ref.load_state_dict({k: v.detach().double().cpu() for k, v in net.state_dict().items()})
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(64, 24, generator=g), torch.randn(64, 8, generator=g)
    cls, reg = torch.randint(0, 5, (64,), generator=g), torch.rand(64, 3, generator=g)

    def loss_fast(m, a, b, cls, reg):
        lc, lr = m(a, b)
        return A.softmax_cross_entropy(lc, cls) + A.mean_squared_error(lr, reg)

    def loss_ref(m, a, b, cls, reg):
        h = torch.relu(a @ m.inp_a.kernel + m.inp_a.bias) + torch.relu(b @ m.inp_b.kernel + m.inp_b.bias)
        h = h + torch.relu(h @ m.block.kernel + m.block.bias)
        lc = h @ m.head_cls.kernel + m.head_cls.bias
        lr = torch.sigmoid(h @ m.head_reg.kernel + m.head_reg.bias)
        return torch.nn.functional.cross_entropy(lc, cls) + ((lr - reg) ** 2).mean()

    l1 = loss_fast(net, a.to(device), b.to(device), cls.to(device), reg.to(device))
    l1.backward()
    l2 = loss_ref(ref, a.double(), b.double(), cls, reg.double())
    l2.backward()
    _close(l1, l2)
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), ref.named_parameters()):
        assert n1 == n2
        _close(p1.grad, p2.grad)
    # and it trains with a stock PyTorch optimizer
    opt = torch.optim.Adam(net.parameters(), lr=3e-3)
    first = float(l1.detach())
    for _ in range(60):
        opt.zero_grad()
        l = loss_fast(net, a.to(device), b.to(device), cls.to(device), reg.to(device))
        l.backward()
        opt.step()
    assert float(l.detach()) < 0.6 * first

"""The one-launch training step (csrc/kernels/mlp_fused.hip) against the layer-by-layer HIP path and the CPU
reference: same seeds -> same dropout masks -> same variables, slots, statistics and step counters."""
import pytest
import torch

from distributed_tensorflow_amd.data import get_data
from distributed_tensorflow_amd.layers import Dense, Dropout
from distributed_tensorflow_amd.models import Sequential, xor_mlp

pytestmark = pytest.mark.gpu


def _pair(device, optimizer="adam", metrics=("binary_rounded_accuracy",), devices=None):
    ms = []
    for dev in (devices or (device, device)):
        m = xor_mlp()
        m.compile(loss="mean_squared_error", optimizer=optimizer, metrics=list(metrics), device=dev, seed=5)
        ms.append(m)
    for m in ms:                       # fixed dropout streams: the comparison is reproducible run to run
        for i, l in enumerate(m.layers):
            if hasattr(l, "seed"):
                l.seed = 4242 + 17 * i
    return ms


def test_plan_detection(gpu_device):
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", device=gpu_device, seed=0)
    plan = m.fused_step_plan(50)
    assert plan is not None and plan["dims"] == (64, 128, 128, 32) and plan["l1"].fused_keep == pytest.approx(0.7)
    assert m.fused_step_plan(65) is None                       # batch beyond the LDS-resident tile
    other = Sequential([Dense(64, input_dim=64, activation="relu"), Dense(32, activation="sigmoid")])
    other.compile(loss="mean_squared_error", optimizer="adam", device=gpu_device, seed=0)
    assert other.fused_step_plan(50) is None
    with pytest.raises(RuntimeError):
        other.fit(torch.zeros(100, 64), torch.zeros(100, 32), batch_size=50, epochs=1, verbose=0, fused=True)


@pytest.mark.parametrize("optimizer", ["adam", "sgd"])
def test_fused_equals_layerwise(gpu_device, optimizer):
    """Both paths compute the same sums in different orders, so they agree to rounding.  Adam on a ReLU network is
    not a contraction: one pre-activation within rounding distance of zero flips a mask and moves a rarely-active
    weight by ~10 learning rates, so the long run is compared through its loss and the bulk of the weights, and the
    strict element-wise comparison is made over the first 20 steps."""
    xt, yt = get_data(2000, val_set_size=0, seed=2, device=gpu_device)[:2]
    a, b = _pair(gpu_device, optimizer)
    for i in range(20):
        a.train_step(xt[i * 50:(i + 1) * 50], yt[i * 50:(i + 1) * 50])
    b.train_steps_fused(xt, yt, 50, 20)
    assert a.global_step == b.global_step == 20
    torch.testing.assert_close(a.store.flat, b.store.flat, rtol=2e-4, atol=2e-6)
    for k in a.store.slots:
        torch.testing.assert_close(a.store.slots[k], b.store.slots[k], rtol=2e-4, atol=1e-9)
    torch.testing.assert_close(a.store.beta_powers, b.store.beta_powers, rtol=1e-5, atol=0)
    a.stats.zero_()
    b.stats.zero_()
    ha = a.fit(xt, yt, epochs=2, batch_size=50, shuffle=False, verbose=0, fused=False)
    hb = b.fit(xt, yt, epochs=2, batch_size=50, shuffle=False, verbose=0, fused=True)
    assert a.global_step == b.global_step == 100
    d = (a.store.flat - b.store.flat).abs()
    assert float((d > 1e-4).float().mean()) < 0.01 and float(d.median()) < 1e-6
    assert ha.history["loss"][-1] == pytest.approx(hb.history["loss"][-1], rel=1e-3)
    assert ha.history["acc"][-1] == pytest.approx(hb.history["acc"][-1], abs=2e-3)


def test_fused_ragged_last_batch_and_keras_accuracy(gpu_device):
    xt, yt = get_data(130, val_set_size=0, seed=4, device=gpu_device)[:2]     # 50 + 50 + 30
    a, b = _pair(gpu_device, metrics=("accuracy", "binary_rounded_accuracy"))
    ha = a.fit(xt, yt, epochs=3, batch_size=50, shuffle=False, verbose=0, fused=False)
    hb = b.fit(xt, yt, epochs=3, batch_size=50, shuffle=False, verbose=0)     # automatic selection
    assert a.global_step == b.global_step == 9
    torch.testing.assert_close(a.store.flat, b.store.flat, rtol=2e-4, atol=2e-6)
    for key in ("loss", "acc", "binary_rounded_accuracy"):
        assert ha.history[key][-1] == pytest.approx(hb.history[key][-1], rel=1e-4, abs=1e-6), key


def test_fused_matches_cpu_reference(gpu_device):
    xt, yt = get_data(500, val_set_size=0, seed=7)[:2]
    g, c = _pair(gpu_device, devices=(gpu_device, torch.device("cpu")))
    g.train_steps_fused(xt.to(gpu_device), yt.to(gpu_device), 50, 10)
    for i in range(10):
        c.train_step(xt[i * 50:(i + 1) * 50], yt[i * 50:(i + 1) * 50])
    torch.testing.assert_close(g.store.flat.cpu(), c.store.flat, rtol=2e-3, atol=2e-5)
    lg, lc = g.read_stats(), c.read_stats()
    assert lg["loss"] == pytest.approx(lc["loss"], abs=1e-4) and lg["acc"] == pytest.approx(lc["acc"], abs=2e-3)
    assert g.global_step == 10


def test_fused_no_dropout_and_tiny_batch(gpu_device):
    ms = []
    for _ in range(2):
        m = Sequential([Dense(128, input_dim=64, activation="relu"), Dense(128, activation="relu"),
                        Dense(32, activation="sigmoid")])
        m.compile(loss="mean_squared_error", optimizer="adam", device=gpu_device, seed=11)
        ms.append(m)
    a, b = ms
    xt, yt = get_data(64, val_set_size=0, seed=9, device=gpu_device)[:2]
    for i in range(8):
        a.train_step(xt[i * 8:(i + 1) * 8], yt[i * 8:(i + 1) * 8])
    b.train_steps_fused(xt, yt, 8, 8)
    torch.testing.assert_close(a.store.flat, b.store.flat, rtol=2e-4, atol=2e-6)
    assert a.global_step == b.global_step == 8


def test_async_workers_inside_the_gpu_share_the_variables(gpu_device):
    """Four asynchronous workers (one compute unit each) take the steps of a launch between them, lock-free on the
    shared variables - the reference's training mode.  Checked: ``global_step`` advances by exactly the number of
    steps, every mini-batch is consumed once, the control words are re-armed, training reaches the accuracy of the
    single-worker run (the races make runs differ in the last digits, not in quality)."""
    import distributed_tensorflow_amd.ops.functional as Fn
    from distributed_tensorflow_amd.data import get_data
    from distributed_tensorflow_amd.models import xor_mlp
    xt, yt, xv, yv = get_data(30000, seed=3, device=gpu_device)
    accs = {}
    for workers in (1, 4):
        m = xor_mlp()
        m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=gpu_device, seed=1)
        plan = m.fused_step_plan(50)
        assert plan is not None
        for _ in range(6):                               # 6 x 600 steps
            m.stats.zero_()
            m.train_steps_fused(xt, yt, 50, 600, plan, workers=workers)
            seen = m.stats[2].item()
            assert seen == 50 * 600                      # every mini-batch exactly once
        assert m.global_step == 3600
        accs[workers] = m.evaluate(xv, yv, batch_size=1000)[1]
        if workers > 1:
            assert Fn._fused_control_words(gpu_device).tolist() == [-1, 0, 0, 0]
    # step 3600 is on the steep part of this task's learning curve and the racing updates differ from run to run:
    # the asynchronous run has been seen between 0.84 and 0.91 where the single worker reaches 0.905
    assert accs[1] > 0.8 and accs[4] > 0.78 and accs[4] > accs[1] - 0.1, accs
    # a launch with fewer steps than workers, and one step: nobody takes a step twice
    m.stats.zero_()
    m.train_steps_fused(xt, yt, 50, 3, plan, workers=8)
    assert m.global_step == 3603 and m.stats[2].item() == 150


def test_fused_gradient_kernel_matches_the_layer_kernels(gpu_device):
    """``dtf_mlp3_fused_grad`` (forward, loss and backward of one batch in one launch, what a cluster replica runs before
    it hands its gradients to the strategy) against the layer-by-layer kernels: same gradients, same statistics, and
    variables / step counter untouched.  Dropout on: both paths derive the masks from (seed, step, element)."""
    import torch
    from distributed_tensorflow_amd.data import get_data
    from distributed_tensorflow_amd.models import xor_mlp
    from distributed_tensorflow_amd.ops import functional as F
    x, y = get_data(200, val_set_size=0, seed=3, device=gpu_device)[:2]
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=gpu_device, seed=5)
    st = m.store
    st.global_step.fill_(17)
    eff = m._effective
    offsets = [v.offset for l in eff for v in (l.kernel, l.bias)]
    dims = (64, 128, 128, 32)
    for B in (50, 64, 7):
        xb, yb = x[:B].contiguous(), y[:B].contiguous()
        m.stats.zero_()
        m.compute_gradients(xb, yb)                     # LocalStrategy: the layer kernels
        ref_grad, ref_stats = st.flat_grad.clone(), m.stats.clone()
        flat0, step0 = st.flat.clone(), int(st.global_step.item())
        grads = torch.full_like(st.flat_grad, float("nan"))
        stats = torch.zeros_like(m.stats)
        F.mlp3_fused_grad(xb, yb, st.flat, grads, offsets, dims, True, m._extra_categorical, eff[0].fused_keep,
                          eff[1].fused_keep, eff[0].seed, eff[1].seed, st.global_step, stats)
        for l in eff:
            for v in (l.kernel, l.bias):
                torch.testing.assert_close(grads[v.offset:v.offset + v.numel], ref_grad[v.offset:v.offset + v.numel],
                                           rtol=2e-4, atol=2e-6, msg="%s at batch %d" % (v.name, B))
        torch.testing.assert_close(stats[:3], ref_stats[:3], rtol=1e-4, atol=1e-5)
        assert torch.equal(st.flat, flat0) and int(st.global_step.item()) == step0

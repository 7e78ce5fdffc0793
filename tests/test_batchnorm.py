"""BatchNormalization (Keras semantics, last axis): forward in both learning phases, moving averages, all gradients
against PyTorch autograd / ``torch.nn.functional.batch_norm``; on the CPU reference path and - marked gpu - on the HIP
kernels (vector and scalar channel counts, row counts that do not fill the last block); checkpoints carry Keras' names."""
import pytest
import torch

from distributed_tensorflow_amd.layers import BatchNormalization, Conv2D, Dense, Flatten, MaxPooling2D
from distributed_tensorflow_amd.models import Sequential
from distributed_tensorflow_amd.ops import functional as F


def _run_functional(device, R, C, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(R, C, generator=g) * 2 + 0.5)
    dy = torch.randn(R, C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    mm, mv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    eps, mom = 1e-3, 0.9
    # reference in fp64
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = mm.double().clone(), mv.double().clone()
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    y_ref = (xd - mean) / torch.sqrt(var + eps) * gd + bd
    (y_ref * dy.double()).sum().backward()
    rm = mom * rm + (1 - mom) * mean.detach()
    rv = mom * rv + (1 - mom) * var.detach()

    t = lambda a: a.clone().to(device)
    xk, dyk, gk, bk, mmk, mvk = t(x), t(dy), t(gamma), t(beta), t(mm), t(mv)
    ws = torch.empty(F.batchnorm_workspace_floats(R, C), device=device) if torch.device(device).type == "cuda" else None
    y, sm, si = torch.empty_like(xk), torch.empty(C, device=device), torch.empty(C, device=device)
    F.batchnorm_forward(xk, gk, bk, mmk, mvk, mom, eps, True, y, sm, si, ws)
    torch.testing.assert_close(y.cpu().double(), y_ref.detach(), rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(mmk.cpu().double(), rm, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(mvk.cpu().double(), rv, rtol=1e-4, atol=1e-6)
    dg, db, dx = torch.empty(C, device=device), torch.empty(C, device=device), torch.empty_like(xk)
    F.batchnorm_backward(xk, dyk, gk, sm, si, dg, db, dx, ws)
    torch.testing.assert_close(db.cpu().double(), bd.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dg.cpu().double(), gd.grad, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(dx.cpu().double(), xd.grad, rtol=2e-3, atol=2e-5)
    # inference: the moving statistics
    yi = torch.empty_like(xk)
    F.batchnorm_forward(xk, gk, bk, mmk, mvk, mom, eps, False, yi, sm, si, ws)
    want = torch.nn.functional.batch_norm(x.double(), mmk.cpu().double(), mvk.cpu().double(), gamma.double(), beta.double(),
                                          False, 0.0, eps)
    torch.testing.assert_close(yi.cpu().double(), want, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("R,C", [(50, 128), (777, 10), (4096, 32)])
def test_functional_cpu(R, C):
    _run_functional("cpu", R, C, R + C)


@pytest.mark.gpu
@pytest.mark.parametrize("R,C", [(50, 128), (777, 10), (70001, 64), (300, 1300), (100000, 6), (5, 2048)])
def test_functional_gpu(gpu_device, R, C):
    _run_functional(gpu_device, R, C, R + C)


def _models(device):
    m = Sequential([Conv2D(8, (3, 3), padding="same", input_shape=(8, 8, 4)), BatchNormalization(), MaxPooling2D((2, 2)),
                    Flatten(), Dense(16, activation="relu"), BatchNormalization(momentum=0.9), Dense(3, activation="sigmoid")])
    m.compile(loss="mean_squared_error", optimizer="sgd", device=device, seed=4)
    return m


def _torch_grads(m, x, y, training=True):
    w = {v.name: v.tensor.detach().cpu().double().requires_grad_(v.trainable) for v in m.store.variables.values()
         if v.tensor.dtype == torch.float32}
    bn = lambda h, p, axis_last: torch.nn.functional.batch_norm(
        h, w[p + "/moving_mean"].clone(), w[p + "/moving_variance"].clone(), w[p + "/gamma"], w[p + "/beta"], training, 0.1, 1e-3)
    h = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w["conv2d_1/kernel"].permute(3, 2, 0, 1), w["conv2d_1/bias"],
                                   padding=1)
    h = bn(h, "batch_normalization_1", False)
    h = torch.nn.functional.max_pool2d(h, 2).permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    h = torch.relu(h @ w["dense_1/kernel"] + w["dense_1/bias"])
    h = bn(h, "batch_normalization_2", True)
    p = torch.sigmoid(h @ w["dense_2/kernel"] + w["dense_2/bias"])
    ((p - y.double()) ** 2).mean().backward()
    return p.detach(), {k: t.grad for k, t in w.items() if t.requires_grad}


def check_model(device):
    g = torch.Generator().manual_seed(9)
    x, y = torch.randn(12, 8, 8, 4, generator=g), torch.rand(12, 3, generator=g)
    m = _models(device)
    _, want = _torch_grads(m, x, y)
    m.compute_gradients(x.to(device), y.to(device))
    for v in m.store.trainable:
        torch.testing.assert_close(v.grad.cpu().double(), want[v.name], rtol=3e-3, atol=3e-6, msg=lambda s: v.name + ": " + s)
    # the step above moved the moving averages; prediction uses them
    mv = m.store.variables["batch_normalization_2/moving_variance"].tensor
    assert not torch.allclose(mv.cpu(), torch.ones(16))
    p_want, _ = _torch_grads(m, x, y, training=False)
    p = torch.as_tensor(m.predict(x.to(device), batch_size=12)).cpu().double()
    torch.testing.assert_close(p, p_want, rtol=2e-4, atol=2e-5)
    return m


def test_model_cpu(tmp_path):
    m = check_model("cpu")
    from distributed_tensorflow_amd.saver import Saver, list_variables
    prefix = Saver(m.store, m.optimizer).save(str(tmp_path / "bn.ckpt"), global_step=0)
    names = dict(list_variables(prefix))
    for k in ("gamma", "beta", "moving_mean", "moving_variance"):
        assert names["batch_normalization_1/" + k] == (8,)
    m2 = _models("cpu")
    Saver(m2.store, m2.optimizer).restore(prefix)
    a = m.store.variables["batch_normalization_1/moving_mean"].tensor
    assert torch.equal(m2.store.variables["batch_normalization_1/moving_mean"].tensor, a) and a.abs().sum() > 0
    cfg = m.get_config()
    assert Sequential.from_config(cfg).get_config()["layers"][1]["class"] == "BatchNormalization"


@pytest.mark.gpu
def test_model_gpu(gpu_device):
    check_model(gpu_device)


def test_training_improves_and_keras_namespace():
    from distributed_tensorflow_amd import compat as tf
    from distributed_tensorflow_amd.data import mnist_like
    assert issubclass(tf.keras.layers.BatchNormalization, BatchNormalization)
    x, y = mnist_like(2000, seed=1)
    m = Sequential([Dense(64, input_dim=784), BatchNormalization(), Dense(64, activation="relu"), Dense(10, activation="softmax")])
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device="cpu", seed=1)
    l0 = m.evaluate(x, y, batch_size=500)[0]
    m.fit(x, y, epochs=3, batch_size=100, verbose=0)
    l1, acc = m.evaluate(x, y, batch_size=500)[:2]
    assert l1 < 0.5 * l0 and acc > 0.8


ASYNC_BN_WORKER = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
torch.set_num_threads(1)
from distributed_tensorflow_amd.cluster import ClusterSpec, Server
from distributed_tensorflow_amd.data import mnist_like
from distributed_tensorflow_amd.engine import ModelEngine
from distributed_tensorflow_amd.layers import BatchNormalization, Dense
from distributed_tensorflow_amd.models import Sequential
from distributed_tensorflow_amd.parallel.async_ps import AsyncPSStrategy

hosts, ckpt = sys.argv[1].split(","), sys.argv[2]
srv = Server(ClusterSpec({"worker": hosts}), "worker", 0, config={"device": "cpu"})
strat = AsyncPSStrategy(srv, max_staleness=0)
m = Sequential([Dense(16, input_dim=784), BatchNormalization(momentum=0.5), Dense(10, activation="softmax")])
m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", strategy=strat, device="cpu", seed=2)
eng = ModelEngine(m)
eng.synchronize_state(True)
x, y = mnist_like(64 * 4, seed=0)
for i in range(4):
    m.train_step(x[i * 64:(i + 1) * 64], y[i * 64:(i + 1) * 64])
prefix = eng.save_checkpoint(eng.make_saver(), ckpt, 4)
mm = m.store.variables["batch_normalization_1/moving_mean"].tensor.tolist()
strat.worker_done()
strat.shutdown_owners()
print("RESULT " + json.dumps({"prefix": prefix, "moving_mean": mm}))
'''


def test_async_checkpoint_carries_the_chiefs_moving_statistics(tmp_path):
    """Asynchronous parameter servers: the owners write the trained variables and their slots; the moving statistics
    live on the workers, and the chief's are added to shard 0."""
    import json
    import os
    import subprocess
    import sys
    from distributed_tensorflow_amd.saver import BundleReader
    from .test_example_cluster import _env, _ports
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(ASYNC_BN_WORKER % {"root": root})
    ports = _ports(2)
    out = subprocess.run([sys.executable, str(script), "127.0.0.1:%d" % ports[0], str(tmp_path / "model.ckpt")],
                         env=_env({"DTF_RENDEZVOUS_PORT": str(ports[1])}), capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    with BundleReader(res["prefix"]) as r:
        assert r.has_tensor("batch_normalization_1/gamma") and r.has_tensor("batch_normalization_1/gamma/Adam")
        got = r.get_tensor("batch_normalization_1/moving_mean").tolist()
        assert r.get_tensor("batch_normalization_1/moving_variance").shape == (16,)
    assert got == pytest.approx(res["moving_mean"]) and any(abs(v) > 1e-4 for v in got)


def check_autograd_function(device):
    from distributed_tensorflow_amd.ops import autograd as A
    g = torch.Generator().manual_seed(11)
    x = torch.randn(6, 5, 5, 8, generator=g)
    gamma, beta = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g)
    w = torch.randn(6, 5, 5, 8, generator=g)
    for training in (True, False):
        mm, mv = torch.zeros(8) + 0.1, torch.ones(8) * 1.5
        xa, ga, ba = (t.clone().to(device).requires_grad_(True) for t in (x, gamma, beta))
        mma, mva = mm.clone().to(device), mv.clone().to(device)
        (A.batch_norm(xa, ga, ba, mma, mva, 0.9, 1e-3, training) * w.to(device)).sum().backward()
        xr, gr, br = (t.clone().double().requires_grad_(True) for t in (x, gamma, beta))
        mmr, mvr = mm.double().clone(), mv.double().clone()
        y = torch.nn.functional.batch_norm(xr.permute(0, 3, 1, 2), mmr, mvr, gr, br, training, 0.1, 1e-3).permute(0, 2, 3, 1)
        (y * w.double()).sum().backward()
        torch.testing.assert_close(xa.grad.cpu().double(), xr.grad, rtol=2e-3, atol=2e-5)
        torch.testing.assert_close(ga.grad.cpu().double(), gr.grad, rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(ba.grad.cpu().double(), br.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(mma.cpu().double(), mmr, rtol=1e-5, atol=1e-6)      # (PyTorch's running variance is the
        # unbiased one, Keras' the biased one: only the mean is compared)


def test_autograd_function_cpu():
    check_autograd_function("cpu")


@pytest.mark.gpu
def test_autograd_function_gpu(gpu_device):
    check_autograd_function(gpu_device)


def test_graph_model_with_a_skip_connection_uses_it():
    """A residual block with batch normalisation in the graph front end (a DAG, so it runs on the autograd functions)."""
    from distributed_tensorflow_amd import compat as tf
    tf.reset_default_graph()
    K = tf.keras.backend
    x = tf.placeholder(tf.float32, shape=(None, 16))
    y = tf.placeholder(tf.float32, shape=(None, 16))
    h = tf.keras.layers.Dense(16, activation="relu")(x)
    h = tf.keras.layers.BatchNormalization(momentum=0.8)(h)
    out = tf.keras.layers.Dense(16, activation="sigmoid")(h + x)
    loss = tf.reduce_mean(tf.keras.losses.mean_squared_error(y, out))
    train = tf.train.AdamOptimizer(0.01).minimize(loss)
    g = torch.Generator().manual_seed(2)
    bx = torch.rand(64, 16, generator=g).numpy()
    by = (torch.rand(64, 16, generator=g) > 0.5).float().numpy()
    with tf.train.MonitoredTrainingSession() as sess:
        eng = tf.get_default_graph().engine()
        assert eng.dag
        first = sess.run(loss, {x: bx, y: by, K.learning_phase(): 1})
        for _ in range(60):
            sess.run(train, {x: bx, y: by, K.learning_phase(): 1})
        last = sess.run(loss, {x: bx, y: by, K.learning_phase(): 0})       # inference mode: the moving statistics
        mv = eng.model.store.variables["batch_normalization_1/moving_variance"].tensor
    assert last < first and not torch.allclose(mv.cpu(), torch.ones(16))


@pytest.mark.gpu
def test_graph_replay_equals_eager_with_batchnorm(gpu_device):
    """``fit(steps_per_graph=N)``: the batch-normalisation kernels (statistics, moving averages, gradients) replay from
    a captured hipGraph exactly as they run eagerly."""
    from distributed_tensorflow_amd.data import mnist_like
    x, y = mnist_like(1024, seed=2, device=gpu_device)
    finals = []
    for spg in (0, 4):
        m = Sequential([Dense(32, input_dim=784), BatchNormalization(momentum=0.9), Dense(32, activation="relu"),
                        Dense(10, activation="softmax")])
        m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device=gpu_device, seed=5)
        m.fit(x, y, epochs=2, batch_size=128, shuffle=False, verbose=0, steps_per_graph=spg, fused=False)
        finals.append((m.store.flat.clone(), m.store.variables["batch_normalization_1/moving_variance"].tensor.clone()))
    torch.testing.assert_close(finals[0][0], finals[1][0], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(finals[0][1], finals[1][1], rtol=1e-6, atol=1e-7)
    assert not torch.allclose(finals[0][1].cpu(), torch.ones(32))

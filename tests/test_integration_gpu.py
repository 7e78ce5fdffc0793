"""End-to-end runs on one MI355X: convergence of the reference workload, hipGraph replay, RCCL path, checkpoints."""
import os
import subprocess
import sys

import pytest
import torch

from distributed_tensorflow_amd.data import get_data, mnist_like
from distributed_tensorflow_amd.models import mnist_mlp, xor_mlp
from distributed_tensorflow_amd.ops import functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_library_is_what_runs(gpu_device):
    from distributed_tensorflow_amd.ops import _native
    _native.kernels()
    maps = open("/proc/self/maps").read()
    assert "libdtf_kernels.so" in maps


def test_gpu_matches_cpu_reference_training(gpu_device):
    """Same seeds, same batches: the HIP path and the PyTorch CPU reference stay together for 20 steps."""
    xt, yt = get_data(1000, val_set_size=0, seed=3)[:2]
    models = []
    for dev in (gpu_device, torch.device("cpu")):
        m = xor_mlp()
        m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=dev, seed=5)
        models.append(m)
    g, c = models
    for lg, lc in zip(g.layers, c.layers):      # same dropout streams
        if hasattr(lg, "seed"):
            lc.seed = lg.seed
    for i in range(20):
        xb, yb = xt[i * 50:(i + 1) * 50], yt[i * 50:(i + 1) * 50]
        g.train_step(xb.to(gpu_device), yb.to(gpu_device))
        c.train_step(xb, yb)
    torch.testing.assert_close(g.store.flat.cpu(), c.store.flat, rtol=2e-3, atol=2e-5)
    lg, lc = g.read_stats(), c.read_stats()
    assert abs(lg["loss"] - lc["loss"]) < 1e-4 and abs(lg["acc"] - lc["acc"]) < 2e-3


def test_reference_workload_with_asynchronous_workers_inside_the_gpu(gpu_device):
    """The reference's own semantics - several workers updating shared variables without locks, one shared
    ``global_step`` as the budget (/root/reference/example.py:168-170,187) - inside one GPU: four workers, one compute
    unit each, take the 30000 steps between them.  Measured over these five seeds: mean validation accuracy 0.932
    (0.953 with one worker; `profiles/r3_xor_async_workers.txt`); the updates race, so runs differ slightly and the
    assertion leaves a margin."""
    finals = []
    for seed in range(5):
        xt, yt, xv, yv = get_data(30000, seed=1 + seed, device=gpu_device)
        m = xor_mlp()
        m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=gpu_device,
                  seed=seed)
        h = m.fit(xt, yt, epochs=50, batch_size=50, validation_data=(xv, yv), verbose=0, async_workers=4)
        assert m.global_step == 30000
        assert h.history["val_acc"][-1] > 0.86, h.history["val_acc"][-5:]
        finals.append(h.history["val_acc"][-1])
    assert sum(finals) / len(finals) >= 0.915, finals


@pytest.mark.parametrize("path", ["fused_step_kernel", "graph_replay"])
def test_reference_workload_converges(gpu_device, path):
    """The reference run: B=50, 600 steps/epoch, 50 epochs = 30000 steps, Adam 1e-3, dropout 0.3.
    Validation accuracy (rounded bits, dropout off) must reach the plateau of the CPU proxy (0.93), through the
    one-launch step kernel and through hipGraph replay of the layer-by-layer step.
    The plateau depends on the draw (data, initialisation, dropout masks, shuffling): single runs land between
    about 0.90 and 0.985, so the claim is checked on the mean of five seeds (runs are reproducible: data, initialisation,
    dropout and the epoch shuffles are all seeded)."""
    finals = []
    for seed in range(5):
        xt, yt, xv, yv = get_data(30000, seed=1 + seed, device=gpu_device)
        m = xor_mlp()
        m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=gpu_device,
                  seed=seed)
        assert m.fused_step_plan(50) is not None
        h = m.fit(xt, yt, epochs=50, batch_size=50, validation_data=(xv, yv), verbose=0, steps_per_graph=100,
                  fused=(path == "fused_step_kernel"))
        assert m.global_step == 30000
        assert h.history["val_acc"][4] > 0.80           # learning well under way after 5 of 50 epochs
        assert h.history["val_acc"][-1] > 0.88, h.history["val_acc"][-5:]
        finals.append(h.history["val_acc"][-1])
    assert sum(finals) / len(finals) >= 0.93, finals


def test_graph_replay_equals_eager(gpu_device):
    xt, yt = get_data(2000, val_set_size=0, seed=2, device=gpu_device)[:2]
    res = []
    for spg in (0, 10):
        m = xor_mlp()
        for l in m.layers:
            if hasattr(l, "seed"):
                l.seed = 77 + len(res) * 0
        m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=gpu_device, seed=9)
        seeds = [getattr(l, "seed", None) for l in m.layers]
        res.append((m, seeds))
    a, b = res[0][0], res[1][0]
    for la, lb in zip(a.layers, b.layers):
        if hasattr(la, "seed"):
            lb.seed = la.seed
    ha = a.fit(xt, yt, epochs=2, batch_size=50, shuffle=False, verbose=0, steps_per_graph=0, fused=False)
    hb = b.fit(xt, yt, epochs=2, batch_size=50, shuffle=False, verbose=0, steps_per_graph=10, fused=False)
    assert a.global_step == b.global_step == 80
    torch.testing.assert_close(a.store.flat, b.store.flat, rtol=1e-5, atol=1e-7)
    assert abs(ha.history["loss"][-1] - hb.history["loss"][-1]) < 1e-6


def test_graphed_train_step_equals_eager(gpu_device):
    """``train_step_graphed`` (persistent batch buffers + one hipGraph of forward, loss and backward, then the
    strategy's apply) against ``train_step``: the same variables after 40 steps on varying batches, two batch shapes."""
    xt, yt = get_data(2000, val_set_size=0, seed=4, device=gpu_device)[:2]
    models = []
    for _ in range(2):
        m = xor_mlp()
        m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=gpu_device, seed=9)
        models.append(m)
    a, b = models
    for la, lb in zip(a.layers, b.layers):
        if hasattr(la, "seed"):
            lb.seed = la.seed
    for i in range(40):
        n = 50 if i % 4 else 32                       # a second shape: its own buffers and graph
        lo = (i * 50) % 1900
        a.train_step(xt[lo:lo + n], yt[lo:lo + n])
        b.train_step_graphed(xt[lo:lo + n], yt[lo:lo + n])
    assert a.global_step == b.global_step == 40
    torch.testing.assert_close(a.store.flat, b.store.flat, rtol=1e-5, atol=1e-7)
    assert a.read_stats()["loss"] == pytest.approx(b.read_stats()["loss"], rel=1e-5)


def test_mnist_mlp_learns_and_checkpoints(gpu_device, tmp_path):
    from distributed_tensorflow_amd.saver import Saver, list_variables
    x, y = mnist_like(16384, seed=0, device=gpu_device)
    xt, yt = mnist_like(4096, seed=1, device=gpu_device)
    m = mnist_mlp()
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device=gpu_device, seed=0)
    h = m.fit(x, y, epochs=8, batch_size=256, validation_data=(xt, yt), verbose=0)
    assert h.history["loss"][-1] < h.history["loss"][0]
    assert h.history["val_acc"][-1] > 0.5
    s = Saver(m.store, m.optimizer)
    prefix = s.save(str(tmp_path / "model.ckpt"), global_step=m.global_step)
    names = dict(list_variables(prefix))
    assert names["dense_1/kernel"] == (784, 128) and names["dense_3/kernel"] == (128, 10)
    m2 = mnist_mlp()
    m2.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device=gpu_device, seed=1)
    Saver(m2.store, m2.optimizer).restore(prefix)
    assert torch.equal(m2.store.flat, m.store.flat)
    assert m2.evaluate(xt, yt, batch_size=1024)[1] == pytest.approx(h.history["val_acc"][-1], abs=1e-6)


def test_rccl_path_single_rank(gpu_device, tmp_path):
    """The sync-replica strategy with real RCCL calls (reduce-scatter / all-gather / broadcast) on a 1-rank group
    must equal the local strategy."""
    code = r'''
import os, sys, torch
sys.path.insert(0, %r)
os.environ["DTF_FORCE_COLLECTIVES"] = "1"
from distributed_tensorflow_amd.data import mnist_like
from distributed_tensorflow_amd.models import mnist_mlp
from distributed_tensorflow_amd.parallel import Communicator, SyncReplicasStrategy, init_process_group
dev = torch.device("cuda", 0)
init_process_group("nccl", rank=0, world_size=1, master_addr="127.0.0.1", master_port=%d, device=dev)
x, y = mnist_like(4096, seed=0, device=dev)
outs = []
for mode in ("local", "bucketed", "replicated", "replicated+graph"):
    strat = None
    if mode == "bucketed":
        strat = SyncReplicasStrategy(Communicator(), bucket_bytes=1 << 16, allreduce_below_bytes=0)
    elif mode.startswith("replicated"):
        strat = SyncReplicasStrategy(Communicator())
    m = mnist_mlp()
    for i, l in enumerate(m.layers):
        if hasattr(l, "seed"): l.seed = 100 + i
    m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], strategy=strat, device=dev, seed=0)
    for i in range(8):
        xb, yb = x[i*512:(i+1)*512], y[i*512:(i+1)*512]
        if mode == "replicated+graph":
            m.graphed_gradients(xb, yb).replay()      # forward+loss+backward from a hipGraph, RCCL call eager
            m.strategy.apply_gradients(m)
        else:
            m.train_step(xb, yb)
    torch.cuda.synchronize()
    outs.append([v.tensor.clone() for v in m.store.trainable] + [m.global_step])
    if mode == "bucketed":
        assert len(strat.buckets) > 1 and not strat.replicated
    if mode.startswith("replicated"):
        assert strat.replicated and not strat.buckets
for other in outs[1:]:
    for a, b in zip(outs[0][:-1], other[:-1]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)
    assert outs[0][-1] == other[-1] == 8
import torch.distributed as dist
dist.destroy_process_group()
print("RCCL_OK")
''' % (ROOT, 29500 + os.getpid() % 1000)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_graph_style_example_on_gpu(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "example.py"), "--log_dir", str(tmp_path),
                          "--last_step", "3000", "--seed", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "Running single-machine training" in out.stdout and "Training complete!" in out.stdout
    assert "Validation Accuracy:" in out.stdout


def test_learning_rate_change_recaptures_the_graph(gpu_device):
    """A captured hipGraph holds the learning rate as a kernel argument: after a scheduler changes it, replayed steps
    must use the new value (same result as eager launches)."""
    from distributed_tensorflow_amd import callbacks as cb
    from distributed_tensorflow_amd.models import mnist_mlp
    from distributed_tensorflow_amd.data import mnist_like
    x, y = mnist_like(2048, seed=5, device=gpu_device)
    flats = []
    for spg in (4, 0):
        m = mnist_mlp(dropout=0.0)
        m.compile(loss="sparse_categorical_crossentropy", optimizer="adam", metrics=["accuracy"], device=gpu_device, seed=2)
        m.fit(x, y, batch_size=256, epochs=3, shuffle=False, verbose=0, steps_per_graph=spg, fused=False,
              callbacks=[cb.LearningRateScheduler(lambda epoch, lr: 1e-3 * (0.3 ** epoch))])
        assert m.optimizer.lr == pytest.approx(1e-3 * 0.09)
        flats.append(m.store.flat.clone())
    torch.testing.assert_close(flats[0], flats[1], rtol=1e-4, atol=1e-6)


def test_host_feeds_through_pinned_ring_equal_device_feeds(gpu_device):
    """Session-style steps fed with host arrays (staged through the pinned ring, never read back, so copies and
    kernels of several steps are in flight) must train exactly as the same steps fed with device tensors."""
    from distributed_tensorflow_amd.engine import TRAIN_OP, FeedStager, ModelEngine

    def build():
        m = xor_mlp()
        m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device=gpu_device, seed=11)
        return m, ModelEngine(m)

    x, y, _, _ = get_data(50 * 12, val_set_size=0, seed=4)
    x, y = x.numpy() if hasattr(x, "numpy") else x, y.numpy() if hasattr(y, "numpy") else y
    m1, e1 = build()
    m2, e2 = build()
    m3, e3 = build()
    for i in range(12):      # more steps than the ring is deep
        xb, yb = x[i * 50:(i + 1) * 50], y[i * 50:(i + 1) * 50]
        e1.run([TRAIN_OP], {"x": xb.astype("float64"), "y": yb})                      # host arrays, dtype converted on the way
        e2.run([TRAIN_OP], {"x": torch.as_tensor(xb, dtype=torch.float32).to(gpu_device), "y": torch.as_tensor(yb).to(gpu_device)})
        e3.run([TRAIN_OP], {"x": torch.as_tensor(xb, dtype=torch.float32), "y": torch.as_tensor(yb)})   # host tensors
    assert isinstance(e1._stager, FeedStager) and not hasattr(e2, "_stager") and isinstance(e3._stager, FeedStager)
    assert torch.equal(m1.store.flat, m2.store.flat) and torch.equal(m3.store.flat, m2.store.flat)
    assert int(m1.store.global_step.item()) == 12

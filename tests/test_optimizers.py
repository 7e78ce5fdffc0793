"""RMSProp and Adagrad (TensorFlow and Keras forms): the flat-range update rules against independent fp64 formulas and
torch.optim, slot names / initial values in checkpoints, and the two rules under the synchronous and the asynchronous
strategies (what a script naming ``tf.train.RMSPropOptimizer`` / ``'rmsprop'`` needs beyond Adam,
/root/reference/example.py:168, /root/reference/example2.py:165)."""
import numpy as np
import pytest
import torch

from distributed_tensorflow_amd import optimizers
from distributed_tensorflow_amd.data import get_data
from distributed_tensorflow_amd.models import xor_mlp
from distributed_tensorflow_amd.ops import functional as F
from distributed_tensorflow_amd.saver import Saver, list_variables, BundleReader


def _rand(n, seed):
    return torch.randn(n, generator=torch.Generator().manual_seed(seed))


def test_rmsprop_tf_form_matches_the_formula_and_torch():
    n = 1003
    var, ms, mom = _rand(n, 1), torch.ones(n), torch.zeros(n)
    ref = var.double().clone()
    a, mo = torch.ones(n, dtype=torch.double), torch.zeros(n, dtype=torch.double)
    for t in range(4):
        g = _rand(n, 10 + t)
        F.rmsprop_apply(var, ms, g, 0.01, 0.9, mom, 0.5, 1e-10, True, grad_scale=0.5)
        gd = g.double() * 0.5
        a = 0.9 * a + 0.1 * gd * gd
        mo = 0.5 * mo + 0.01 * gd / (a + 1e-10).sqrt()
        ref = ref - mo
    torch.testing.assert_close(var.double(), ref, rtol=1e-5, atol=1e-6)
    # Keras form == torch.optim.RMSprop (accumulator from zero, epsilon outside the root)
    p = torch.nn.Parameter(_rand(n, 2))
    opt = torch.optim.RMSprop([p], lr=1e-3, alpha=0.9, eps=1e-8)
    var, ms = p.detach().clone(), torch.zeros(n)
    for t in range(4):
        g = _rand(n, 20 + t)
        p.grad = g.clone()
        opt.step()
        F.rmsprop_apply(var, ms, g, 1e-3, 0.9, None, 0.0, 1e-8, False)
    torch.testing.assert_close(var, p.detach(), rtol=1e-5, atol=1e-6)


def test_adagrad_forms():
    n = 517
    p = torch.nn.Parameter(_rand(n, 3))
    opt = torch.optim.Adagrad([p], lr=0.01, initial_accumulator_value=0.1, eps=0.0)     # TensorFlow's ApplyAdagrad
    var, acc = p.detach().clone(), torch.full((n,), 0.1)
    for t in range(4):
        g = _rand(n, 30 + t)
        p.grad = g.clone()
        opt.step()
        F.adagrad_apply(var, acc, g, 0.01, 0.0, True)
    torch.testing.assert_close(var, p.detach(), rtol=1e-5, atol=1e-6)
    var, acc, ref, a = _rand(n, 4), torch.zeros(n), _rand(n, 4).double(), torch.zeros(n, dtype=torch.double)
    for t in range(3):                                                                    # Keras: sqrt(acc) + epsilon
        g = _rand(n, 40 + t)
        F.adagrad_apply(var, acc, g, 0.01, 1e-8, False)
        a = a + g.double() ** 2
        ref = ref - 0.01 * g.double() / (a.sqrt() + 1e-8)
    torch.testing.assert_close(var.double(), ref, rtol=1e-5, atol=1e-6)


def test_adadelta_matches_torch():
    n = 733
    p = torch.nn.Parameter(_rand(n, 5))
    opt = torch.optim.Adadelta([p], lr=1.0, rho=0.95, eps=1e-8)
    var, acc, upd = p.detach().clone(), torch.zeros(n), torch.zeros(n)
    for t in range(5):
        g = _rand(n, 50 + t)
        p.grad = g.clone()
        opt.step()
        F.adadelta_apply(var, acc, upd, g, 1.0, 0.95, 1e-8)
    torch.testing.assert_close(var, p.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("spec,slots", [
    ("adadelta", {"Adadelta": 0.0, "Adadelta_1": 0.0}),
    ("rmsprop", {"RMSProp": 0.0}),
    (optimizers.RMSProp(lr=1e-3, momentum=0.9), {"RMSProp": 1.0, "RMSProp_1": 0.0}),
    ("adagrad", {"Adagrad": 0.0}),
    (optimizers.Adagrad(lr=0.05), {"Adagrad": 0.1}),
])
def test_training_and_checkpoint_names(tmp_path, spec, slots):
    x, y, xv, yv = get_data(4000, val_set_size=500, seed=2)
    m = xor_mlp(dropout=0.0)
    m.compile(loss="mean_squared_error", optimizer=spec, metrics=["binary_rounded_accuracy"], device="cpu", seed=3)
    first = {suffix: m.store.slot_view(name, m.store.trainable[0]).flatten()[0].item()
             for name, suffix in m.optimizer.slot_checkpoint_names.items() if name in m.store.slots}
    assert first == pytest.approx(slots)                    # TensorFlow starts `rms` at one and `accumulator` at 0.1
    l0 = m.evaluate(xv, yv, batch_size=500)[0]
    m.fit(x, y, epochs=4, batch_size=50, verbose=0)
    assert m.evaluate(xv, yv, batch_size=500)[0] < l0
    prefix = Saver(m.store, m.optimizer).save(str(tmp_path / "model.ckpt"), global_step=m.global_step)
    names = dict(list_variables(prefix))
    for suffix in slots:
        assert names["dense_1/kernel/" + suffix] == (64, 128)
    assert "beta1_power" not in names
    # a fresh model resumes with the slots of the file
    m2 = xor_mlp(dropout=0.0)
    m2.compile(loss="mean_squared_error", optimizer=spec if isinstance(spec, str) else optimizers.from_description(spec.describe()),
               metrics=["binary_rounded_accuracy"], device="cpu", seed=9)
    Saver(m2.store, m2.optimizer).restore(prefix)
    assert torch.equal(m2.store.flat, m.store.flat)
    for name in m.store.slots:
        for v, v2 in zip(m.store.trainable, m2.store.trainable):      # (the padding behind the last variable is not saved)
            assert torch.equal(m2.store.slot_view(name, v2), m.store.slot_view(name, v))
    m.fit(x[:500], y[:500], epochs=1, batch_size=50, verbose=0, shuffle=False)
    m2._host_step = m2.global_step
    m2.fit(x[:500], y[:500], epochs=1, batch_size=50, verbose=0, shuffle=False)
    torch.testing.assert_close(m2.store.flat, m.store.flat, rtol=1e-6, atol=1e-7)


def test_describe_round_trip_and_pump_codes():
    for opt in (optimizers.Adam(lr=2e-3), optimizers.SGD(lr=0.1, momentum=0.8), optimizers.RMSProp(momentum=0.5),
                optimizers.RMSProp(keras=True), optimizers.Adagrad(), optimizers.Adagrad(keras=True), optimizers.Adadelta(lr=0.5)):
        back = optimizers.from_description(opt.describe())
        assert type(back) is type(opt) and back.describe() == opt.describe()
        assert back.pump_parameters() == opt.pump_parameters() and back.slot_order == opt.slot_order
    assert [o.pump_parameters()[0] for o in (optimizers.Adam(), optimizers.SGD(), optimizers.RMSProp(), optimizers.Adagrad(),
                                             optimizers.Adadelta())] == [0, 1, 2, 3, 4]


def test_tf_namespace_builds_them():
    from distributed_tensorflow_amd import compat as tf
    a = tf.train.RMSPropOptimizer(0.01, momentum=0.9).optimizer
    assert isinstance(a, optimizers.RMSProp) and a.slot_order == ["rms", "momentum"] and a.epsilon == 1e-10 and not a.keras
    b = tf.train.AdagradOptimizer(0.1).optimizer
    assert isinstance(b, optimizers.Adagrad) and b.slot_initial("accumulator") == pytest.approx(0.1)
    c = tf.train.AdadeltaOptimizer().optimizer
    assert isinstance(c, optimizers.Adadelta) and c.lr == 0.001 and optimizers.get("adadelta").lr == 1.0
    assert tf.keras.optimizers.Adadelta().lr == 1.0
    with pytest.raises(NotImplementedError):
        tf.train.RMSPropOptimizer(0.01, centered=True)


ASYNC_WORKER = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
torch.set_num_threads(1)
from distributed_tensorflow_amd import optimizers
from distributed_tensorflow_amd.cluster import ClusterSpec, Server
from distributed_tensorflow_amd.data import get_data
from distributed_tensorflow_amd.models import xor_mlp
from distributed_tensorflow_amd.parallel.async_ps import AsyncPSStrategy

index, hosts, steps, desc = int(sys.argv[1]), sys.argv[2].split(","), int(sys.argv[3]), json.loads(sys.argv[4])
device = sys.argv[5] if len(sys.argv) > 5 else "cpu"
srv = Server(ClusterSpec({"worker": hosts}), "worker", index, config={"device": device})
strat = AsyncPSStrategy(srv, max_staleness=0)
m = xor_mlp(dropout=0.0)
m.compile(loss="mean_squared_error", optimizer=optimizers.from_description(desc), metrics=["binary_rounded_accuracy"],
          strategy=strat, device=device, seed=4)
x, y = get_data(50 * steps, val_set_size=0, seed=index, device=device)[:2]
for i in range(steps):
    m.train_step(x[i * 50:(i + 1) * 50], y[i * 50:(i + 1) * 50])
strat.worker_done()
strat.cp0.barrier("all_done", len(hosts), timeout_s=120)
strat.pull(m)
gs = strat.global_step()
strat.cp0.barrier("pulled", len(hosts), timeout_s=120)
if srv.is_chief:
    strat.shutdown_owners()
print("RESULT " + json.dumps({"index": index, "global_step": gs, "flat": m.store.flat.tolist()}))
'''


@pytest.mark.parametrize("opt", [optimizers.RMSProp(lr=2e-3, momentum=0.9), optimizers.Adagrad(lr=0.05),
                                 optimizers.RMSProp(keras=True), optimizers.Adadelta(lr=1.0)],
                         ids=["rmsprop-momentum", "adagrad", "rmsprop-keras", "adadelta"])
def test_parameter_server_owner_applies_the_rule(tmp_path, opt):
    _owner_applies_the_rule(tmp_path, opt, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("opt", [optimizers.RMSProp(lr=2e-3, momentum=0.9), optimizers.Adagrad(lr=0.05),
                                 optimizers.Adadelta(lr=1.0)], ids=["rmsprop-momentum", "adagrad", "adadelta"])
def test_parameter_server_owner_applies_the_rule_on_the_gpu(tmp_path, opt, gpu_device):
    _owner_applies_the_rule(tmp_path, opt, str(gpu_device))


def _owner_applies_the_rule(tmp_path, opt, device):
    """One worker that is its own variable owner (the native owner service applies every gradient, its slots in the
    owner's region in the optimizer's order): the variables after 20 steps equal those of the same steps taken locally.
    Then two workers: every gradient is applied once and both end with the owners' variables."""
    import json
    import os
    import subprocess
    import sys
    from .test_example_cluster import _env, _ports
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(ASYNC_WORKER % {"root": root})
    steps = 20

    def run(n):
        ports = _ports(n + 1)
        hosts = ",".join("127.0.0.1:%d" % p for p in ports[:n])
        env = _env({"DTF_RENDEZVOUS_PORT": str(ports[n]), "DTF_COMM_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        if device != "cpu":
            env.pop("CUDA_VISIBLE_DEVICES"), env.pop("HIP_VISIBLE_DEVICES")       # the workers time-share the test's GPU
        procs = [subprocess.Popen([sys.executable, str(script), str(i), hosts, str(steps), json.dumps(opt.describe()), device],
                                  env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for i in range(n)]
        out = []
        try:
            for p in procs:
                text, _ = p.communicate(timeout=240)
                assert p.returncode == 0, text[-3000:]
                out.append(json.loads([l for l in text.splitlines() if l.startswith("RESULT ")][-1][7:]))
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        return sorted(out, key=lambda r: r["index"])

    one = run(1)[0]
    assert one["global_step"] == steps
    local = xor_mlp(dropout=0.0)
    local.compile(loss="mean_squared_error", optimizer=optimizers.from_description(opt.describe()),
                  metrics=["binary_rounded_accuracy"], device=device, seed=4)
    x, y = get_data(50 * steps, val_set_size=0, seed=0, device=device)[:2]
    for i in range(steps):
        local.train_step(x[i * 50:(i + 1) * 50], y[i * 50:(i + 1) * 50])
    # (on the GPU the worker's gradient comes from the one-launch kernel, the local model's from the layer kernels:
    # the same sums in another order)
    tol = dict(rtol=2e-5, atol=2e-6) if device == "cpu" else dict(rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(torch.tensor(one["flat"]), local.store.flat.cpu(), **tol)
    two = run(2)
    assert all(r["global_step"] == 2 * steps for r in two) and two[0]["flat"] == two[1]["flat"]


def test_sharded_synchronous_slots_start_from_the_rules_value():
    """A single-rank run of the sharded synchronous strategy (slots exist only as the rank's shards) equals local
    training with TensorFlow's RMSProp, whose mean-square slot starts at one."""
    import os
    import torch.distributed as dist
    from distributed_tensorflow_amd.parallel.comm import Communicator
    from distributed_tensorflow_amd.parallel.sync_replicas import SyncReplicasStrategy
    from .test_example_cluster import _ports
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_ports(1)[0]), DTF_FORCE_COLLECTIVES="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        x, y = get_data(50 * 10, val_set_size=0, seed=1)[:2]
        models = []
        for strategy in (None, SyncReplicasStrategy(Communicator(), allreduce_below_bytes=0)):
            m = xor_mlp(dropout=0.0)
            m.compile(loss="mean_squared_error", optimizer=optimizers.RMSProp(lr=1e-3, momentum=0.5), strategy=strategy,
                      metrics=["binary_rounded_accuracy"], device="cpu", seed=6)
            for i in range(10):
                m.train_step(x[i * 50:(i + 1) * 50], y[i * 50:(i + 1) * 50])
            models.append(m)
        st = models[1].strategy
        assert not st.replicated and set(st.slot_shards) == {"rms", "momentum"}
        torch.testing.assert_close(models[1].store.flat, models[0].store.flat, rtol=1e-6, atol=1e-7)
    finally:
        os.environ.pop("DTF_FORCE_COLLECTIVES", None)
        dist.destroy_process_group()

"""MonitoredTrainingSession + hooks on the Sequential engine: stop, checkpoints, summaries, step counter, recovery."""
import glob
import os

import pytest
import torch

from distributed_tensorflow_amd import session as ss
from distributed_tensorflow_amd import summary as sm
from distributed_tensorflow_amd.data import get_data
from distributed_tensorflow_amd.engine import ACCURACY, GLOBAL_STEP, LOSS, TRAIN_OP, ModelEngine
from distributed_tensorflow_amd.models import xor_mlp
from distributed_tensorflow_amd.saver import latest_checkpoint


def make(seed=0):
    m = xor_mlp()
    m.compile(loss="mean_squared_error", optimizer="adam", metrics=["binary_rounded_accuracy"], device="cpu", seed=seed)
    return m, ModelEngine(m)


DATA = get_data(500, val_set_size=100, seed=0)


def loop(sess, n=10 ** 9):
    xt, yt = DATA[0], DATA[1]
    i = 0
    while not sess.should_stop() and i < n:
        lo = (i * 50) % 450
        sess.run([LOSS, ACCURACY, TRAIN_OP], {"x": xt[lo:lo + 50], "y": yt[lo:lo + 50]})
        i += 1
    return i


def test_stop_at_step_absolute_and_relative(tmp_path):
    m, eng = make()
    with ss.MonitoredTrainingSession(engine=eng, hooks=[ss.StopAtStepHook(last_step=25)]) as sess:
        assert loop(sess) == 25 and sess.global_step_value() == 25
    with ss.MonitoredTrainingSession(engine=eng, hooks=[ss.StopAtStepHook(num_steps=5)]) as sess:
        assert loop(sess) == 5 and sess.global_step_value() == 30
    with pytest.raises(ValueError):
        ss.StopAtStepHook()


def test_chief_default_hooks_write_files(tmp_path):
    sm.reset_collection()
    sm.FileWriterCache.clear()
    m, eng = make()
    sm.scalar("loss", lambda: eng._last_logs.get("loss", 0.0))
    d = str(tmp_path)
    with ss.MonitoredTrainingSession(engine=eng, is_chief=True, checkpoint_dir=d, hooks=[ss.StopAtStepHook(last_step=250)],
                                     save_checkpoint_secs=600, save_summaries_steps=100, log_step_count_steps=100) as sess:
        loop(sess)
    sm.FileWriterCache.clear()
    assert latest_checkpoint(d).endswith("model.ckpt-250")
    assert os.path.exists(os.path.join(d, "model.ckpt-1.index"))        # saved right after the first step, too
    (ev_path,) = glob.glob(os.path.join(d, "events.out.tfevents.*"))
    ev = [e for e in sm.read_events(ev_path) if "scalars" in e]
    assert sorted(e["step"] for e in ev if "loss" in e["scalars"]) == [1, 101, 201]
    assert sorted(e["step"] for e in ev if "global_step/sec" in e["scalars"]) == [100, 200]
    sm.reset_collection()


def test_non_chief_writes_nothing(tmp_path):
    m, eng = make()
    with ss.MonitoredTrainingSession(engine=eng, is_chief=False, checkpoint_dir=str(tmp_path),
                                     hooks=[ss.StopAtStepHook(last_step=5)]) as sess:
        loop(sess)
    assert os.listdir(str(tmp_path)) == []


def test_resume_restores_variables_slots_and_step(tmp_path):
    d = str(tmp_path)
    m, eng = make(seed=0)
    with ss.MonitoredTrainingSession(engine=eng, checkpoint_dir=d, hooks=[ss.StopAtStepHook(last_step=20)]) as sess:
        loop(sess)
    flat, mslot = m.store.flat.clone(), m.store.slot("m").clone()
    m2, eng2 = make(seed=123)
    with ss.MonitoredTrainingSession(engine=eng2, checkpoint_dir=d, hooks=[ss.StopAtStepHook(last_step=20)]) as sess:
        assert sess.global_step_value() == 20 and sess.should_stop()     # budget already spent: nothing runs
        assert torch.equal(m2.store.flat, flat) and torch.equal(m2.store.slot("m"), mslot)
        assert abs(m2.store.beta_powers[0].item() - 0.9 ** 21) < 1e-6     # TF convention: beta^(t+1) after t updates


def test_recoverable_error_restores_last_checkpoint_and_retries(tmp_path):
    d = str(tmp_path)
    m, eng = make()
    fault = ss.FaultInjectionHook(at_step=12, error=ss.AbortedError)
    hooks = [ss.CheckpointSaverHook(d, save_secs=None, save_steps=5), fault, ss.StopAtStepHook(last_step=20)]
    with ss.MonitoredSession(eng, checkpoint_dir=d, hooks=hooks, recovery_wait_secs=0.0) as sess:
        n = loop(sess)
        assert fault.fired and sess._recoveries == 1
        assert sess.global_step_value() == 20
    # steps 11 and 12 were lost with the fault (last checkpoint at 11: saves at 1, 6, 11) and redone
    assert n == 20 + 1
    assert latest_checkpoint(d).endswith("model.ckpt-20")


def test_fatal_error_propagates_and_skips_end_hooks(tmp_path):
    m, eng = make()
    ck = ss.CheckpointSaverHook(str(tmp_path), save_secs=None, save_steps=1000)
    with pytest.raises(ZeroDivisionError):
        with ss.MonitoredSession(eng, hooks=[ck, ss.FaultInjectionHook(at_step=3, error=ZeroDivisionError)]) as sess:
            loop(sess)
    assert ck.saved_steps == [1]           # no final save after a crash


def test_logging_and_nan_hooks():
    m, eng = make()
    lines = []
    hooks = [ss.LoggingTensorHook({"step": GLOBAL_STEP}, every_n_iter=2, log=lines.append), ss.StopAtStepHook(last_step=4)]
    with ss.MonitoredSession(eng, hooks=hooks) as sess:
        loop(sess)
    assert lines == ["INFO: step = 1", "INFO: step = 3"]    # fetched with the train op: the value after the update
    m.store.flat.fill_(float("nan"))
    with pytest.raises(FloatingPointError):
        with ss.MonitoredSession(eng, hooks=[ss.NanLossHook(LOSS)]) as sess:
            loop(sess, 1)


def test_control_plane_timeout_is_recovered_like_unavailable(tmp_path):
    """A real transient fault - a control-plane request that timed out inside the engine - takes the same path as TF's
    UnavailableError: restore the last checkpoint, retry the run."""
    from distributed_tensorflow_amd.control import ControlTimeout
    d = str(tmp_path)
    m, eng = make()
    real_run, fired = eng.run, []

    def flaky(fetches, feed_dict=None):
        if m._host_step == 7 and not fired:
            fired.append(1)
            raise ControlTimeout("take_many co/reply/0/0 timed out")
        return real_run(fetches, feed_dict)

    eng.run = flaky
    hooks = [ss.CheckpointSaverHook(d, save_secs=None, save_steps=5), ss.StopAtStepHook(last_step=12)]
    with ss.MonitoredSession(eng, checkpoint_dir=d, hooks=hooks, recovery_wait_secs=0.0) as sess:
        loop(sess)
        assert fired and sess._recoveries == 1
        assert sess.global_step_value() == 12


def test_hooks_share_one_step_read_per_run_and_no_device_sync(tmp_path, monkeypatch):
    """A ``run`` that fetches nothing from the device costs no device->host read: every hook shares ONE read of the
    engine's host-side step counter (StopAtStep, StepCounter, SummarySaver, CheckpointSaver all ask).  Reads made by
    the CPU reference kernels (``ops/functional.py`` takes the step from the tensor; the HIP kernels read it on the
    device) are not the session's."""
    import traceback
    m, eng = make()
    reads = []
    real_item, real_tolist = torch.Tensor.item, torch.Tensor.tolist

    def note(real):
        def wrapped(self):
            caller = traceback.extract_stack()[-2].filename
            if not caller.endswith(os.path.join("ops", "functional.py")):
                reads.append(caller)
            return real(self)
        return wrapped

    monkeypatch.setattr(torch.Tensor, "item", note(real_item))
    monkeypatch.setattr(torch.Tensor, "tolist", note(real_tolist))
    with ss.MonitoredTrainingSession(engine=eng, checkpoint_dir=str(tmp_path), save_summaries_steps=None,
                                     hooks=[ss.StopAtStepHook(last_step=40)]) as sess:
        x, y = DATA[0][:50], DATA[1][:50]
        sess.run(TRAIN_OP, {"x": x, "y": y})               # first step: the checkpoint hook saves (reads allowed)
        del reads[:]
        s0 = sess.step_reads
        for _ in range(20):
            sess.run(TRAIN_OP, {"x": x, "y": y})
        assert sess.step_reads - s0 == 20                   # one shared read per run ...
        assert reads == []                                  # ... served from the host mirror: no tensor was read back


def test_step_latency_percentiles_and_comm_time(tmp_path):
    sm.FileWriterCache.clear()
    m, eng = make()
    m.strategy.comm_seconds = 0.0                    # a strategy that keeps the counter
    hook = ss.StepLatencyHook(every_n_steps=10, output_dir=str(tmp_path))
    orig = eng.run

    def run(f, fd=None):
        m.strategy.comm_seconds += 0.002
        return orig(f, fd)
    eng.run = run
    with ss.MonitoredSession(eng, hooks=[hook, ss.StopAtStepHook(last_step=20)]) as sess:
        loop(sess)
    assert set(hook.last) == {"step_latency_ms/p50", "step_latency_ms/p90", "step_latency_ms/p99", "step_latency_ms/max",
                              "comm_ms/step"}
    assert hook.last["step_latency_ms/p50"] <= hook.last["step_latency_ms/p99"] <= hook.last["step_latency_ms/max"]
    assert abs(hook.last["comm_ms/step"] - 2.0) < 1e-6
    assert ss.StepLatencyHook.percentiles(list(range(1, 101))) == {"p50": 51, "p90": 91, "p99": 100, "max": 100}
    sm.FileWriterCache.clear()
    (path,) = glob.glob(os.path.join(str(tmp_path), "events.out.tfevents.*"))
    tags = set()
    for e in sm.read_events(path):
        tags |= set(e.get("scalars", {}))
    assert {"step_latency_ms/p99", "comm_ms/step"} <= tags


def test_fit_inside_session_names_final_checkpoint_by_trained_step(tmp_path):
    """`model.fit` inside a session never passes through `run` (/root/reference/example2.py:189-200): the end hooks
    must still see the real step, so the final checkpoint is `model.ckpt-<steps trained>`."""
    m, eng = make()
    d = str(tmp_path)
    xt, yt = DATA[0], DATA[1]
    with ss.MonitoredTrainingSession(engine=eng, checkpoint_dir=d, save_summaries_steps=None, log_step_count_steps=None) as sess:
        assert sess.global_step_value() == 0
        m.fit(xt, yt, epochs=2, batch_size=50, verbose=0, shuffle=False)
        sess.check_health()
        assert sess.global_step_value() == 20
    assert latest_checkpoint(d).endswith("model.ckpt-20")

"""Fault test: the training process is killed mid-run; a fresh process started on the same log directory restores
the latest checkpoint and continues from its global_step (what MonitoredTrainingSession does for the reference at
/root/reference/example.py:189-192 when a task restarts)."""
import os
import subprocess
import sys
import time

from distributed_tensorflow_amd.saver import BundleReader, latest_checkpoint

from .test_example_cluster import SCRIPT, _env


def test_kill_then_resume_from_latest_checkpoint(tmp_path):
    log_dir = str(tmp_path)
    args = [sys.executable, SCRIPT, "--log_dir", log_dir, "--seed", "3", "--save_checkpoint_secs", "0.3"]
    first = subprocess.Popen(args + ["--last_step", "1000000"], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             text=True)
    try:
        deadline = time.time() + 180
        step = 0
        while time.time() < deadline:
            ck = latest_checkpoint(log_dir)
            if ck is not None:
                step = int(ck.rsplit("-", 1)[1])
                if step >= 200:
                    break
            assert first.poll() is None, first.stdout.read()[-2000:]
            time.sleep(0.1)
        assert step >= 200, "no checkpoint appeared"
    finally:
        first.kill()            # SIGKILL of exactly this child: no clean shutdown, no final checkpoint
        first.wait(timeout=30)
    ck = latest_checkpoint(log_dir)
    saved = int(ck.rsplit("-", 1)[1])
    with BundleReader(ck) as r:
        assert int(r.get_tensor("global_step").item()) == saved
        w_saved = r.get_tensor("dense_1/kernel").clone()
    target = saved + 150
    out = subprocess.run(args + ["--last_step", str(target)], env=_env(), capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "Training complete!" in out.stdout
    assert "Restoring parameters from %s" % ck in out.stdout
    final = latest_checkpoint(log_dir)
    assert final.endswith("model.ckpt-%d" % target)           # it ran exactly to the absolute step, not 'target' more steps
    with BundleReader(final) as r:
        assert int(r.get_tensor("global_step").item()) == target
        assert not (r.get_tensor("dense_1/kernel") == w_saved).all()

"""The filled-in outlines and the Keras variant of the example run as scripts (single machine, CPU): the reference
ships outline_tensorflow.py / outline_keras.py empty (/root/reference/outline_tensorflow.py, outline_keras.py) and
example2.py as its Keras path (/root/reference/example2.py:148-200)."""
import glob
import os
import subprocess
import sys

from .test_example_cluster import ROOT, _env


def _run(script, args, tmp_path, extra_env=None):
    env = _env(dict({"DTF_LOGS_ROOT": str(tmp_path)}, **(extra_env or {})))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)] + args, env=env, capture_output=True,
                         text=True, timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    return out.stdout


def test_outline_tensorflow_trains_and_checkpoints(tmp_path):
    text = _run("outline_tensorflow.py", ["--log_dir", str(tmp_path / "logs"), "--last_step", "120"], tmp_path)
    assert "Training complete!" in text
    assert glob.glob(str(tmp_path / "logs" / "model.ckpt-120.index"))
    assert glob.glob(str(tmp_path / "logs" / "events.out.tfevents.*"))


def test_outline_keras_fits_with_tensorboard_callback(tmp_path):
    text = _run("outline_keras.py", [], tmp_path, {"EPOCHS": "2"})
    assert "Epoch 2/2" in text and "val_acc" in text
    assert glob.glob(str(tmp_path / "**" / "events.out.tfevents.*"), recursive=True)


def test_example2_keras_variant(tmp_path):
    text = _run("example2.py", ["--log_dir", str(tmp_path / "logs"), "--epochs", "2", "--seed", "1"], tmp_path)
    assert "Running single-machine training" in text
    assert "Epoch 2/2" in text and "val_acc" in text
    assert glob.glob(str(tmp_path / "logs" / "**" / "events.out.tfevents.*"), recursive=True)

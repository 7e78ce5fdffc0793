"""Host-side sanitizer runs of the C++ runtime: the control plane's thread-per-connection server and the event
writer's queue thread under ThreadSanitizer and AddressSanitizer (CPU only; no GPU code is instrumented)."""
import os
import subprocess

import pytest

from distributed_tensorflow_amd import _build


@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_runtime_stress_under_sanitizer(tmp_path, sanitizer):
    try:
        exe = _build.build_runtime_stress(sanitizer, verbose=False)
    except RuntimeError as e:          # toolchain without the sanitizer runtime
        pytest.skip("cannot build with -fsanitize=%s: %s" % (sanitizer, str(e)[-200:]))
    env = dict(os.environ)
    env["TSAN_OPTIONS"] = "halt_on_error=0 report_signal_unsafe=0"
    env["ASAN_OPTIONS"] = "detect_leaks=1 verify_asan_link_order=0"
    out = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=300, env=env)
    report = out.stdout[-2000:] + out.stderr[-4000:]
    assert "STRESS_OK" in out.stdout, report
    assert out.returncode == 0, report
    assert "ThreadSanitizer" not in out.stderr and "AddressSanitizer" not in out.stderr, report
    assert any(f.startswith("events.out.tfevents") for f in os.listdir(tmp_path))

"""In-tree build of the native libraries.

Two shared objects are produced under ``distributed_tensorflow_amd/_lib``:

``libdtf_kernels.so``
    every HIP kernel in ``csrc/kernels`` compiled by ``hipcc --offload-arch=gfx950``.
``libdtf_runtime.so``
    the host-side C++ runtime in ``csrc/runtime`` (TensorBundle checkpoints, event
    files, crc32c, the TCP control plane); it has no GPU dependency, so it also loads
    and is tested on a CPU-only machine.

hipcc cross-compiles gfx950 code objects without a GPU, so ``build_all`` works on a CPU
box; the resulting ``.so`` files travel with the source tree to the GPU machine.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "csrc"
LIB_DIR = Path(__file__).resolve().parent / "_lib"
OBJ_DIR = ROOT / "build" / "obj"

GPU_ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", shutil.which("hipcc") or "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", shutil.which("g++") or "g++")

KERNEL_FLAGS = ["--offload-arch=" + GPU_ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                "-I", str(CSRC / "kernels")]
RUNTIME_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-pthread", "-Wall",
                 "-I", str(CSRC / "runtime")]


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(str(p).encode())
        h.update(Path(p).read_bytes())
    return h.hexdigest()


def _run(cmd):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(map(str, cmd)), proc.stdout))
    return proc.stdout


def _compile_objects(sources, compiler, flags, headers, verbose, tag):
    obj_dir = OBJ_DIR / tag
    obj_dir.mkdir(parents=True, exist_ok=True)
    hdr_digest = _digest(headers) if headers else ""
    jobs = []
    objs = []
    for src in sources:
        obj = obj_dir / (src.stem + ".o")
        stamp = obj_dir / (src.stem + ".stamp")
        want = _digest([src]) + hdr_digest + " ".join(flags)
        objs.append(obj)
        if obj.exists() and stamp.exists() and stamp.read_text() == want:
            continue
        jobs.append((src, obj, stamp, want))

    def one(job):
        src, obj, stamp, want = job
        if verbose:
            print("[build] %s %s" % (Path(compiler).name, src.name), flush=True)
        _run([compiler, *flags, "-c", str(src), "-o", str(obj)])
        stamp.write_text(want)

    if jobs:
        workers = max(1, min(len(jobs), int(os.environ.get("MAX_JOBS", "6"))))
        with cf.ThreadPoolExecutor(workers) as ex:
            list(ex.map(one, jobs))
    return objs, bool(jobs)


def build_kernels(verbose: bool = True) -> Path:
    sources = sorted((CSRC / "kernels").glob("*.hip"))
    headers = sorted((CSRC / "kernels").glob("*.h"))
    out = LIB_DIR / "libdtf_kernels.so"
    objs, rebuilt = _compile_objects(sources, HIPCC, KERNEL_FLAGS, headers, verbose, "kernels")
    if rebuilt or not out.exists():
        LIB_DIR.mkdir(parents=True, exist_ok=True)
        _run([HIPCC, "--offload-arch=" + GPU_ARCH, "-shared", "-fPIC", "-o", str(out), *map(str, objs)])
        if verbose:
            print("[build] linked %s" % out, flush=True)
    return out


def build_runtime(verbose: bool = True) -> Path:
    sources = sorted((CSRC / "runtime").glob("*.cc"))
    headers = sorted((CSRC / "runtime").glob("*.h"))
    out = LIB_DIR / "libdtf_runtime.so"
    if not sources:
        return out
    objs, rebuilt = _compile_objects(sources, CXX, RUNTIME_FLAGS, headers, verbose, "runtime")
    if rebuilt or not out.exists():
        LIB_DIR.mkdir(parents=True, exist_ok=True)
        _run([CXX, "-shared", "-fPIC", "-pthread", "-o", str(out), *map(str, objs)])
        if verbose:
            print("[build] linked %s" % out, flush=True)
    return out


def build_runtime_stress(sanitizer: str = "thread", verbose: bool = True) -> Path:
    """Host-only stress executable of the C++ runtime (control plane + event writer) instrumented with a sanitizer
    (``thread`` or ``address``); returns its path.  Sanitizers apply to host code only - GPU code is not involved."""
    if sanitizer not in ("thread", "address", "undefined"):
        raise ValueError("sanitizer must be thread, address or undefined")
    out_dir = ROOT / "build" / "bin"
    out_dir.mkdir(parents=True, exist_ok=True)
    out = out_dir / ("stress_runtime_%s" % sanitizer)
    sources = [CSRC / "runtime" / "tests" / "stress_runtime.cc", CSRC / "runtime" / "control_plane.cc",
               CSRC / "runtime" / "events_writer.cc", CSRC / "runtime" / "crc32c.cc",
               CSRC / "runtime" / "input_pipeline.cc"]
    stamp = out_dir / (out.name + ".stamp")
    want = _digest(sources + sorted((CSRC / "runtime").glob("*.h")))
    if out.exists() and stamp.exists() and stamp.read_text() == want:
        return out
    if verbose:
        print("[build] %s -fsanitize=%s stress_runtime" % (Path(CXX).name, sanitizer), flush=True)
    _run([CXX, "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer",
          "-I", str(CSRC / "runtime"), *map(str, sources), "-o", str(out)])
    stamp.write_text(want)
    return out


def build_all(verbose: bool = True):
    return build_kernels(verbose), build_runtime(verbose)


if __name__ == "__main__":
    build_all(verbose=True)
    sys.exit(0)

"""bench.py started plainly (no torchrun): ``--gpus N > 1`` must start N ranks by itself, print ``n_gpus == N`` with N
distinct ranks, and fail loudly - never run a smaller job silently - when the ranks cannot be had.  CPU rehearsal over
gloo (``--device cpu``); the GPU twin lives in test_shared_gpu_ranks.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(extra=None):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "JOB_NAME", "TASK_INDEX", "PS_HOSTS",
              "WORKER_HOSTS"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    env.update(extra or {})
    return env


def _run(args, env=None, timeout=240):
    return subprocess.run([sys.executable, BENCH] + args, env=_env(env), capture_output=True, text=True, timeout=timeout,
                          cwd=ROOT)


@pytest.mark.parametrize("world", [2, 4])
def test_self_launch_runs_every_rank(world):
    out = _run(["--gpus", str(world), "--device", "cpu", "--batch-per-gpu", "128", "--steps", "3", "--warmup", "1",
                "--steady", "0.05", "--eval", "3"])
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # ONE JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1
    c = d["config"]
    assert c["parallelism"] == "dp%d" % world and c["global_batch"] == 128 * world
    # the collectives went over gloo here: the RCCL field must stay empty, the group size is reported separately
    assert c["rccl_ranks"] is None and c["backend"] == "gloo" and c["group_ranks"] == world
    assert c["self_launched"] is True and c["launch_variant"] == "gloo-cpu"
    assert c["preflight"]["allreduce_473KB_max_abs_err"] == 0.0
    assert sorted(r["rank"] for r in c["ranks"]) == list(range(world))
    assert len({r["pid"] for r in c["ranks"]}) == world          # really N processes
    assert c["test_acc"] is not None and 0.0 <= c["test_acc"] <= 1.0
    assert d["value"] > 0 and d["scaling"] == "weak"


def test_missing_gpus_is_an_error_not_a_smaller_run():
    """No GPU here: ``--gpus 2`` on the GPU path must exit non-zero and print no result line."""
    out = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], env={"CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "GPU" in out.stderr


def test_world_size_must_match_gpus():
    out = _run(["--gpus", "2", "--device", "cpu"], env={"RANK": "0", "WORLD_SIZE": "1"})
    assert out.returncode != 0 and "must agree" in (out.stderr + out.stdout)


def test_async_mode_self_launched():
    out = _run(["--gpus", "2", "--device", "cpu", "--mode", "async", "--batch-per-gpu", "64", "--steps", "4", "--warmup", "1",
                "--steady", "0", "--max-staleness", "2"])
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "async-ps2"
    assert sum(d["config"]["staleness_histogram_rank0"].values()) > 0


def test_driver_style_torchrun_launch_on_cpu():
    """The multi-GPU driver's command line (``python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...``), rehearsed with CPU ranks: the ranks join the
    launcher's own store, rank 0 prints the one JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--device", "cpu", "--batch-per-gpu", "128", "--steps", "3",
           "--warmup", "1", "--steady", "0.05", "--eval", "2"]
    out = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["group_ranks"] == 2 and d["config"]["self_launched"] is False
    assert d["config"]["rccl_ranks"] is None and d["config"]["backend"] == "gloo"
    assert sorted(r["rank"] for r in d["config"]["ranks"]) == [0, 1]


def test_failed_bring_up_is_tried_again_with_fresh_ranks():
    """Rank 1 of the first launch variant dies while the group comes up: the supervisor ends that set of children and
    a second, fresh set (the next variant) produces the result."""
    out = _run(["--gpus", "2", "--device", "cpu", "--batch-per-gpu", "128", "--steps", "2", "--warmup", "1", "--steady", "0",
                "--eval", "0"], env={"DTF_BENCH_INJECT_INIT_FAIL": "0:1"})
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["launch_variant"] == "gloo-cpu-second-try" and d["n_gpus"] == 2
    assert "injected bring-up failure" in out.stderr and "trying the next one" in out.stderr


def test_failed_bring_up_under_torchrun_is_tried_again():
    """The same with one supervisor per rank (the driver's torchrun line): they agree on the retry through the
    launcher's store."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--device", "cpu", "--batch-per-gpu", "128", "--steps", "2",
           "--warmup", "1", "--steady", "0", "--eval", "0"]
    out = subprocess.run(cmd, env=_env({"DTF_BENCH_INJECT_INIT_FAIL": "0:1"}), capture_output=True, text=True, timeout=300,
                         cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0])["config"]["launch_variant"] == "gloo-cpu-second-try"


def test_every_variant_failing_is_an_error():
    out = _run(["--gpus", "2", "--device", "cpu", "--batch-per-gpu", "128", "--steps", "2", "--warmup", "1", "--steady", "0",
                "--eval", "0"], env={"DTF_BENCH_INJECT_INIT_FAIL": "0:1,1:0"})
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "no launch variant" in out.stderr

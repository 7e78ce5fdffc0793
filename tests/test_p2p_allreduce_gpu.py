"""One-shot all-reduce over peer pointers (csrc/kernels/p2p_allreduce.hip): three processes that share the test box's
one GPU map each other's exchange buffers through hipIpc and reduce into bit-identical sums.  (On one device the
peer loads are local; the protocol - publish, flag round, rank-ordered sum, double buffering - is what is exercised.)"""
import os
import signal
import subprocess
import sys

import pytest

from .test_example_cluster import _ports

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_p2p_worker.py")


def test_a_late_rank_poisons_the_result_and_is_reported(tmp_path):
    """One rank arrives 1.5 s late with a 0.3 s timeout: the others get NaN (not their local gradient) and
    ``check()`` raises."""
    test_ranks_on_one_gpu_reduce_through_mapped_buffers(2, tmp_path, extra_env={"P2P_DELAY_TEST": "1"})


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_on_one_gpu_reduce_through_mapped_buffers(world, tmp_path, extra_env=None):
    port = _ports(1)[0]
    procs, logs = [], []
    try:
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_PORT=str(port), PYTHONFAULTHANDLER="1",
                       HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **(extra_env or {}))
            logs.append(open(str(tmp_path / ("rank%d.log" % r)), "w+"))
            procs.append(subprocess.Popen([sys.executable, WORKER], env=env, stdout=logs[-1], stderr=subprocess.STDOUT))
        failed = None
        for r, p in enumerate(procs):
            try:
                rc = p.wait(timeout=120)
            except subprocess.TimeoutExpired:
                rc = None
            if rc != 0 and failed is None:
                failed = (r, rc)
        if failed is not None:
            report = []
            for r, p in enumerate(procs):
                if p.poll() is None:
                    p.send_signal(signal.SIGABRT)        # exactly the processes started here; stacks go to the logs
                    try:
                        p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                logs[r].seek(0)
                report.append("==== rank %d rc=%s\n%s" % (r, p.poll(), logs[r].read()[-3000:]))
            raise AssertionError("rank %d rc=%s\n%s" % (failed[0], failed[1], "\n".join(report)))
        for r in range(world):
            logs[r].seek(0)
            assert "P2P_OK rank %d" % r in logs[r].read()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in logs:
            f.close()

"""Task -> device mapping of a one-node cluster: no two tasks of an RCCL cluster may share a GPU."""
import itertools

import pytest

from distributed_tensorflow_amd.cluster import plan_devices


@pytest.mark.parametrize("num_ps,num_workers,n_gpus", list(itertools.product([0, 1, 2], [1, 2, 4, 7, 8], [1, 2, 4, 8])))
def test_no_two_tasks_share_a_gpu(num_ps, num_workers, n_gpus):
    plan = plan_devices(num_ps, num_workers, n_gpus)
    if num_workers > n_gpus:
        # more workers than GPUs (the reference's default local run on a one-GPU machine): workers wrap around,
        # owners are colocated (hipIpc peer regions work on one device), no ps task takes a GPU
        assert plan["oversubscribed"] and plan["colocated"] and plan["ps"] == [None] * num_ps
        assert plan["worker"] == ["cuda:%d" % (i % n_gpus) for i in range(num_workers)]
        return
    assert not plan["oversubscribed"]
    used = [d for d in plan["worker"] + plan["ps"] if d is not None]
    assert len(used) == len(set(used))                       # distinct devices
    assert plan["worker"] == ["cuda:%d" % i for i in range(num_workers)]
    if num_ps and num_workers + num_ps > n_gpus:
        # the ps tasks do not fit: control plane only, owners live inside the workers
        assert plan["ps"] == [None] * num_ps and plan["colocated"]
    elif num_ps:
        assert plan["ps"] == ["cuda:%d" % (num_workers + j) for j in range(num_ps)] and not plan["colocated"]
    else:
        assert plan["colocated"]


def test_one_ps_eight_workers_on_eight_gpus():
    plan = plan_devices(1, 8, 8)
    assert plan == {"worker": ["cuda:%d" % i for i in range(8)], "ps": [None], "colocated": True,
                    "oversubscribed": False}


def test_cpu_and_shared_gpu_plans():
    assert plan_devices(1, 2, 0) == {"worker": ["cpu", "cpu"], "ps": ["cpu"], "colocated": False,
                                     "oversubscribed": False}
    assert plan_devices(0, 3, 0)["colocated"]
    shared = plan_devices(1, 2, 1, shared=True)
    assert shared["worker"] == ["cuda:0", "cuda:0"] and shared["ps"] == ["cuda:0"] and not shared["colocated"]

"""ClusterSpec / placement / device setter / Server basics (surface of /root/reference/example.py:108-141)."""
import pytest

from distributed_tensorflow_amd import cluster as cl


def spec():
    return cl.ClusterSpec({"ps": ["h0:2222", "h1:2222"], "worker": ["w0:2223", "w1:2223", "w2:2223"]})


def test_cluster_spec():
    s = spec()
    assert s.jobs == ["ps", "worker"]
    assert s.num_tasks("worker") == 3 and s.task_indices("ps") == [0, 1]
    assert s.task_address("worker", 2) == "w2:2223" and s.job_tasks("ps") == ["h0:2222", "h1:2222"]
    assert cl.ClusterSpec(s) == s and cl.ClusterSpec(s.as_dict()) == s
    with pytest.raises(ValueError):
        s.num_tasks("evaluator")
    with pytest.raises(ValueError):
        s.task_address("ps", 5)
    assert cl.ClusterSpec({"worker": {0: "a:1", 1: "b:2"}}).job_tasks("worker") == ["a:1", "b:2"]


XOR_VARS = [("dense_1/kernel", 8192 * 4), ("dense_1/bias", 128 * 4), ("dense_2/kernel", 16384 * 4),
            ("dense_2/bias", 128 * 4), ("dense_3/kernel", 4096 * 4), ("dense_3/bias", 32 * 4)]


def test_round_robin_matches_tf_default():
    setter = cl.replica_device_setter(worker_device="/job:worker/task:1", cluster=spec())
    # ops stay on the worker, variables alternate over the ps tasks in creation order
    assert setter("MatMul") == "/job:worker/task:1"
    assert [setter("Variable", name=n) for n, _ in XOR_VARS] == ["/job:ps/task:%d" % (i % 2) for i in range(6)]
    assert setter("Variable", name="dense_1/bias") == "/job:ps/task:1"       # slots colocate with their variable
    assert setter.place_variables(XOR_VARS) == {n: i % 2 for i, (n, _) in enumerate(XOR_VARS)}
    one = cl.replica_device_setter(cluster={"ps": ["a:1"], "worker": ["b:2"]})
    assert set(one.place_variables(XOR_VARS).values()) == {0}


def test_byte_balanced_and_flat():
    p = cl.ByteBalancedPlacement().assign(XOR_VARS, 2)
    loads = [sum(b for n, b in XOR_VARS if p[n] == k) for k in range(2)]
    assert abs(loads[0] - loads[1]) <= max(b for _, b in XOR_VARS)
    rr = cl.RoundRobinPlacement().assign(XOR_VARS, 2)
    rr_loads = [sum(b for n, b in XOR_VARS if rr[n] == k) for k in range(2)]
    assert max(loads) <= max(rr_loads)
    assert set(cl.FlatShardPlacement().assign(XOR_VARS, 4).values()) == {-1}
    setter = cl.replica_device_setter(cluster=spec(), ps_strategy="byte_balanced")
    assert setter.place_variables(XOR_VARS) == p


def test_server_layout_and_errors():
    s = cl.Server(spec(), job_name="worker", task_index=1, start=False)
    assert (s.num_ps, s.num_workers, s.world_size, s.rank, s.is_chief) == (2, 3, 5, 3, False)
    assert s.target == "dtf://w1:2223" and s.control_address(1) == ("h1", 2222)
    assert cl.Server(spec(), job_name="ps", task_index=1, start=False).rank == 1
    assert cl.Server(spec(), job_name="worker", task_index=0, start=False).is_chief
    assert cl.Server.current() is not None
    with pytest.raises(ValueError):
        cl.Server(spec(), job_name="worker", task_index=7, start=False)
    with pytest.raises(ValueError):
        cl.split_host_port("nohost")


def test_device_and_target_messages(capsys):
    assert cl.device_and_target(None, 0, None, None) == (None, "", None)
    assert "Running single-machine training" in capsys.readouterr().out
    for args, msg in (((None, "a:1", "b:2"), "task_index"), ((0, "", "b:2"), "ps_hosts"), ((0, "a:1", None), "worker_hosts")):
        with pytest.raises(ValueError, match="Must specify an explicit `%s`" % msg):
            cl.device_and_target("worker", *args)
    assert "Running distributed training" in capsys.readouterr().out

"""Native TCP control plane: KV rendezvous, counters, queues, staleness clocks, barriers, heartbeats."""
import threading
import time

import pytest

from distributed_tensorflow_amd.control import ControlClient, ControlServer, ControlTimeout


@pytest.fixture()
def server():
    s = ControlServer("127.0.0.1", 0)
    yield s
    s.stop()


def test_kv_counter_queue(server):
    a, b = ControlClient("127.0.0.1", server.port), ControlClient("127.0.0.1", server.port)
    a.set("k", b"v1")
    assert b.get("k") == b"v1"
    a.set_json("cfg", {"x": [1, 2]})
    assert b.get_json("cfg") == {"x": [1, 2]}
    with pytest.raises(ControlTimeout):
        b.get("missing", timeout_s=0.05)
    # a blocked get is released by a later set
    got = []
    t = threading.Thread(target=lambda: got.append(b.get("late", timeout_s=5)))
    t.start()
    time.sleep(0.05)
    a.set("late", b"x" * 1000)
    t.join(5)
    assert got == [b"x" * 1000]
    assert [a.add("gs", 1), b.add("gs", 1), a.add("gs", 0), a.add("gs", 5)] == [1, 2, 2, 7]
    for i in range(3):
        a.push_json("q", {"i": i})
    assert b.qlen("q") == 3 and [b.pop_json("q")["i"] for _ in range(3)] == [0, 1, 2]
    with pytest.raises(ControlTimeout):
        b.pop("q", timeout_s=0.05)
    a.delete("k")
    with pytest.raises(ControlTimeout):
        b.get("k", timeout_s=0.05)


def test_concurrent_counter_is_atomic(server):
    def work():
        c = ControlClient("127.0.0.1", server.port)
        for _ in range(200):
            c.add("n", 1)
        c.close()
    ts = [threading.Thread(target=work) for _ in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert ControlClient("127.0.0.1", server.port).add("n", 0) == 1600


def test_staleness_clock_gate(server):
    fast, slow = ControlClient("127.0.0.1", server.port), ControlClient("127.0.0.1", server.port)
    fast.clock_set("w", 0, 0)
    slow.clock_set("w", 1, 0)
    assert fast.clock_wait("w", 2, 2, timeout_s=1) == 0           # 2 - 0 <= 2: may proceed
    with pytest.raises(ControlTimeout):
        fast.clock_wait("w", 3, 2, timeout_s=0.1)                  # 3 - 0 > 2: gated
    released = []
    fast.clock_set("w", 0, 3)                                      # the fast worker has done three pushes
    waiter = ControlClient("127.0.0.1", server.port)
    t = threading.Thread(target=lambda: released.append(waiter.clock_wait("w", 3, 2, timeout_s=5)))
    t.start()
    time.sleep(0.05)
    assert not released
    slow.clock_set("w", 1, 1)                                      # the slow worker catches up by one
    t.join(5)
    assert released == [1] and fast.clock_min("w") == 1


def test_barrier_and_heartbeat(server):
    n, passed = 4, []

    def work(i):
        c = ControlClient("127.0.0.1", server.port)
        time.sleep(0.01 * i)
        c.barrier("b", n, timeout_s=5)
        passed.append(i)
        c.barrier("b", n, timeout_s=5)       # reusable
    ts = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    [t.start() for t in ts]
    [t.join(10) for t in ts]
    assert sorted(passed) == list(range(n))
    c = ControlClient("127.0.0.1", server.port)
    with pytest.raises(ControlTimeout):
        c.barrier("lonely", 2, timeout_s=0.05)
    c.heartbeat("worker:0")
    c.heartbeat("worker:1")
    time.sleep(0.12)
    c.heartbeat("worker:1")
    assert c.dead_members(0.1) == ["worker:0"] and c.dead_members(10) == []


def test_folded_operations(server):
    a, b = ControlClient("127.0.0.1", server.port), ControlClient("127.0.0.1", server.port)
    # push_many / pop
    a.push_many_json([("q/0", {"i": 0}), ("q/1", {"i": 1}), ("q/0", {"i": 2})])
    assert b.pop_json("q/0")["i"] == 0 and b.pop_json("q/0")["i"] == 2 and b.pop_json("q/1")["i"] == 1
    # take_many blocks until every key is there, returns in order, erases
    got = []
    t = threading.Thread(target=lambda: got.append(b.take_many_json(["r/0", "r/1"], timeout_s=5)))
    t.start()
    a.set_json("r/1", {"v": 1})
    time.sleep(0.05)
    assert not got
    a.set_json("r/0", {"v": 0})
    t.join(5)
    assert got == [[{"v": 0}, {"v": 1}]]
    with pytest.raises(ControlTimeout):
        b.take_many(["r/0"], timeout_s=0.05)
    # add_set: the reply carries the counter value it produced
    assert a.add_set("gs", 1, "reply", b'{"global_step": %GS%}') == 1
    assert a.add_set("gs", 1, "reply", b'{"global_step": %GS%}') == 2
    assert b.take_json("reply") == {"global_step": 2}
    assert a.add_set(None, 0, "plain", b"x") == 0 and b.get("plain") == b"x"
    # wait_counter blocks server-side
    done = []
    t = threading.Thread(target=lambda: done.append(b.wait_counter("ready", 2, timeout_s=5)))
    t.start()
    a.add("ready", 1)
    time.sleep(0.05)
    assert not done
    a.add("ready", 1)
    t.join(5)
    assert done == [2]
    with pytest.raises(ControlTimeout):
        b.wait_counter("ready", 3, timeout_s=0.05)
    # clock_set_wait: publish and gate in one call
    assert a.clock_set_wait("w", 0, 0, 1) == 0
    assert b.clock_set_wait("w", 1, 1, 1, timeout_s=1) == 0
    with pytest.raises(ControlTimeout):
        b.clock_set_wait("w", 1, 2, 1, timeout_s=0.05)          # 2 - 0 > 1
    assert a.clock_set_wait("w", 0, 1, None) == 1                # publish only
    assert b.clock_set_wait("w", 1, 2, 1, timeout_s=1) == 1
    n0 = a.calls
    a.add("x", 1)
    a.push_many([("q", b"1"), ("q", b"2")])
    assert a.calls == n0 + 2


def test_oversized_frame_drops_the_connection_not_the_server(server):
    import socket
    import struct
    s = socket.create_connection(("127.0.0.1", server.port))
    s.sendall(struct.pack("<IB", 0xFFFFFFF0, 1))          # a 4 GB "frame"
    s.settimeout(5)
    assert s.recv(16) == b""                               # closed by the server
    s.close()
    c = ControlClient("127.0.0.1", server.port)
    assert c.add("alive", 1) == 1                          # the server still serves


def test_many_short_connections_are_reaped(server):
    for i in range(64):
        c = ControlClient("127.0.0.1", server.port)
        c.add("n", 1)
        c.close()
    c = ControlClient("127.0.0.1", server.port)
    assert c.add("n", 0) == 64

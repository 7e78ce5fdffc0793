"""Python face of the native TCP control plane (``csrc/runtime/control_plane.cc``).

``ControlServer`` runs inside a task (every "ps" task; the chief in ps-less clusters) and
``ControlClient`` talks to it: key/value rendezvous, atomic counters (the shared
``global_step``), blocking request queues, staleness clocks, barriers and heartbeats.
"""
from __future__ import annotations

import ctypes as C
import json
from typing import List, Optional

from .ops import _native as nv

ST_OK, ST_TIMEOUT, ST_ERROR, ST_CLOSED = 0, 1, -1, -2
_lib = None


def _rt():
    global _lib
    if _lib is not None:
        return _lib
    lib = nv.runtime()
    i64, p, s = C.c_int64, C.c_void_p, C.c_char_p
    pi64 = C.POINTER(C.c_int64)
    nv.declare(lib, "dtf_cp_server_start", [s, C.c_int], p)
    nv.declare(lib, "dtf_cp_server_port", [p])
    nv.declare(lib, "dtf_cp_server_stop", [p], None)
    nv.declare(lib, "dtf_cp_connect", [s, C.c_int, C.c_double], p)
    nv.declare(lib, "dtf_cp_close", [p], None)
    nv.declare(lib, "dtf_cp_set", [p, s, s, i64])
    nv.declare(lib, "dtf_cp_get", [p, s, p, i64, pi64, i64])
    nv.declare(lib, "dtf_cp_delete", [p, s])
    nv.declare(lib, "dtf_cp_add", [p, s, i64, pi64])
    nv.declare(lib, "dtf_cp_push", [p, s, s, i64])
    nv.declare(lib, "dtf_cp_pop", [p, s, p, i64, pi64, i64])
    nv.declare(lib, "dtf_cp_qlen", [p, s, pi64])
    nv.declare(lib, "dtf_cp_clock_set", [p, s, i64, i64, pi64])
    nv.declare(lib, "dtf_cp_clock_min", [p, s, pi64])
    nv.declare(lib, "dtf_cp_clock_wait", [p, s, i64, i64, i64, pi64])
    nv.declare(lib, "dtf_cp_barrier", [p, s, i64, i64])
    nv.declare(lib, "dtf_cp_heartbeat", [p, s])
    nv.declare(lib, "dtf_cp_dead", [p, i64, p, i64, pi64])
    nv.declare(lib, "dtf_cp_take_many", [p, p, i64, p, i64, pi64, i64])
    nv.declare(lib, "dtf_cp_push_many", [p, p, i64, i64])
    nv.declare(lib, "dtf_cp_clock_set_wait", [p, s, i64, i64, i64, i64, pi64])
    nv.declare(lib, "dtf_cp_wait_counter", [p, s, i64, i64, pi64])
    nv.declare(lib, "dtf_cp_add_set", [p, s, i64, s, s, i64, pi64])
    _lib = lib
    return lib


class ControlPlaneError(RuntimeError):
    pass


class ControlTimeout(ControlPlaneError):
    pass


class ControlServer:
    def __init__(self, host: str = "127.0.0.1", port: int = 0):
        self._h = _rt().dtf_cp_server_start(host.encode(), int(port))
        if not self._h:
            raise ControlPlaneError("cannot listen on %s:%d" % (host, port))
        self.host = host
        self.port = _rt().dtf_cp_server_port(self._h)

    def stop(self):
        h, self._h = self._h, None
        if h:
            _rt().dtf_cp_server_stop(h)

    def __del__(self):
        try:
            self.stop()
        except Exception:
            pass


class ControlClient:
    def __init__(self, host: str, port: int, timeout_s: float = 60.0):
        self._h = _rt().dtf_cp_connect(host.encode(), int(port), float(timeout_s))
        if not self._h:
            raise ControlPlaneError("cannot reach control plane at %s:%d" % (host, port))
        self.host, self.port = host, port
        self._buf = C.create_string_buffer(1 << 16)
        self.calls = 0          # round trips made through this client (every method below is exactly one)

    def _check(self, st: int, what: str):
        self.calls += 1
        if st == ST_OK:
            return
        if st == ST_TIMEOUT:
            raise ControlTimeout(what + " timed out")
        raise ControlPlaneError("%s failed (status %d)" % (what, st))

    def _ms(self, timeout_s: Optional[float]) -> int:
        return -1 if timeout_s is None else int(timeout_s * 1000)

    def _read(self, fn, key: bytes, timeout_s) -> bytes:
        n = C.c_int64()
        st = fn(self._h, key, self._buf, len(self._buf), C.byref(n), self._ms(timeout_s))
        if st == ST_ERROR and n.value > len(self._buf):
            # value larger than the scratch buffer: it is lost for queues; size the buffer generously
            raise ControlPlaneError("control-plane value of %d bytes exceeds the client buffer" % n.value)
        self._check(st, key.decode())
        return self._buf.raw[:n.value]

    # -- key/value
    def set(self, key: str, value: bytes):
        self._check(_rt().dtf_cp_set(self._h, key.encode(), value, len(value)), "set " + key)

    def get(self, key: str, timeout_s: Optional[float] = None) -> bytes:
        """Blocks until ``key`` exists."""
        return self._read(_rt().dtf_cp_get, key.encode(), timeout_s)

    def set_json(self, key: str, obj):
        self.set(key, json.dumps(obj).encode())

    def get_json(self, key: str, timeout_s: Optional[float] = None):
        return json.loads(self.get(key, timeout_s).decode())

    def delete(self, key: str):
        self.calls += 1
        _rt().dtf_cp_delete(self._h, key.encode())

    # -- counters
    def add(self, key: str, delta: int = 1) -> int:
        out = C.c_int64()
        self._check(_rt().dtf_cp_add(self._h, key.encode(), int(delta), C.byref(out)), "add " + key)
        return out.value

    # -- queues
    def push(self, queue: str, value: bytes):
        self._check(_rt().dtf_cp_push(self._h, queue.encode(), value, len(value)), "push " + queue)

    def pop(self, queue: str, timeout_s: Optional[float] = None) -> bytes:
        return self._read(_rt().dtf_cp_pop, queue.encode(), timeout_s)

    def push_json(self, queue: str, obj):
        self.push(queue, json.dumps(obj).encode())

    def pop_json(self, queue: str, timeout_s: Optional[float] = None):
        return json.loads(self.pop(queue, timeout_s).decode())

    def qlen(self, queue: str) -> int:
        out = C.c_int64()
        self._check(_rt().dtf_cp_qlen(self._h, queue.encode(), C.byref(out)), "qlen " + queue)
        return out.value

    # -- staleness clocks
    def clock_set(self, group: str, member: int, value: int) -> int:
        out = C.c_int64()
        self._check(_rt().dtf_cp_clock_set(self._h, group.encode(), int(member), int(value), C.byref(out)), "clock_set")
        return out.value

    def clock_min(self, group: str) -> int:
        out = C.c_int64()
        self._check(_rt().dtf_cp_clock_min(self._h, group.encode(), C.byref(out)), "clock_min")
        return out.value

    def clock_wait(self, group: str, mine: int, bound: int, timeout_s: Optional[float] = None) -> int:
        """Block until ``mine - min(group clocks) <= bound``; returns the minimum clock."""
        out = C.c_int64()
        st = _rt().dtf_cp_clock_wait(self._h, group.encode(), int(mine), int(bound), self._ms(timeout_s), C.byref(out))
        self._check(st, "clock_wait")
        return out.value

    # -- folded operations: one round trip each
    def take_many(self, keys, timeout_s: Optional[float] = None) -> List[bytes]:
        """Block until every key exists; return their values in order and erase them."""
        import struct
        keys = list(keys)
        packed = b"".join(k.encode() + b"\0" for k in keys)
        n = C.c_int64()
        st = _rt().dtf_cp_take_many(self._h, packed, len(keys), self._buf, len(self._buf), C.byref(n), self._ms(timeout_s))
        if st == ST_ERROR and n.value > len(self._buf):
            raise ControlPlaneError("control-plane values of %d bytes exceed the client buffer" % n.value)
        self._check(st, "take_many " + ",".join(keys[:3]))
        raw, out, pos = self._buf.raw[:n.value], [], 0
        for _ in keys:
            (ln,) = struct.unpack_from("<I", raw, pos)
            out.append(raw[pos + 4:pos + 4 + ln])
            pos += 4 + ln
        return out

    def take_many_json(self, keys, timeout_s: Optional[float] = None):
        return [json.loads(v.decode()) for v in self.take_many(keys, timeout_s)]

    def take_json(self, key: str, timeout_s: Optional[float] = None):
        return self.take_many_json([key], timeout_s)[0]

    def push_many(self, items):
        """``items``: (queue, bytes) pairs, appended in order."""
        import struct
        items = list(items)
        packed = b"".join(q.encode() + b"\0" + struct.pack("<I", len(v)) + v for q, v in items)
        self._check(_rt().dtf_cp_push_many(self._h, packed, len(packed), len(items)), "push_many")

    def push_many_json(self, items):
        self.push_many([(q, json.dumps(o).encode()) for q, o in items])

    def clock_set_wait(self, group: str, member: int, value: int, bound: Optional[int],
                       timeout_s: Optional[float] = None) -> int:
        """Publish ``value`` as ``member``'s clock, then block until ``value - min(group clocks) <= bound``
        (``bound=None``: publish only).  Returns the minimum clock."""
        out = C.c_int64()
        st = _rt().dtf_cp_clock_set_wait(self._h, group.encode(), int(member), int(value),
                                         -1 if bound is None else int(bound), self._ms(timeout_s), C.byref(out))
        self._check(st, "clock_set_wait")
        return out.value

    def wait_counter(self, key: str, want: int, timeout_s: Optional[float] = None) -> int:
        """Block (on the server, no polling) until counter ``key`` >= ``want``; returns its value."""
        out = C.c_int64()
        self._check(_rt().dtf_cp_wait_counter(self._h, key.encode(), int(want), self._ms(timeout_s), C.byref(out)),
                    "wait_counter " + key)
        return out.value

    def add_set(self, counter: Optional[str], delta: int, key: str, value: bytes) -> int:
        """Add ``delta`` to ``counter`` (skipped when None) and set ``key`` to ``value``, where the text ``%GS%``
        inside ``value`` becomes the counter's new value.  Returns that value."""
        out = C.c_int64()
        self._check(_rt().dtf_cp_add_set(self._h, (counter or "").encode(), int(delta), key.encode(), value, len(value),
                                         C.byref(out)), "add_set " + key)
        return out.value

    # -- barrier / liveness
    def barrier(self, name: str, n: int, timeout_s: Optional[float] = None):
        self._check(_rt().dtf_cp_barrier(self._h, name.encode(), int(n), self._ms(timeout_s)), "barrier " + name)

    def heartbeat(self, who: str):
        self._check(_rt().dtf_cp_heartbeat(self._h, who.encode()), "heartbeat")

    def dead_members(self, max_age_s: float) -> List[str]:
        n = C.c_int64()
        st = _rt().dtf_cp_dead(self._h, int(max_age_s * 1000), self._buf, len(self._buf), C.byref(n))
        self._check(st, "dead_members")
        return [x for x in self._buf.raw[:n.value].decode().split("\n") if x]

    def close(self):
        h, self._h = self._h, None
        if h:
            _rt().dtf_cp_close(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

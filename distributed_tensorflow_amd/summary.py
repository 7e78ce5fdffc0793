"""Summaries and TensorBoard event files.

API of /root/reference/example.py:160,164,172-174,195,219: ``summary.scalar(name, value)``,
``summary.merge_all()``, ``FileWriter(logdir)``, ``writer.add_graph(...)``,
``writer.add_summary(s, step)``.  Event files are written by the native asynchronous writer
(``csrc/runtime/events_writer.cc``) in the TFRecord/Event format TensorBoard reads.

A "summary op" here is a named callable returning a Python/torch scalar; ``merge_all()``
returns an op evaluating every registered scalar into one :class:`Summary`.  The graph-style
session (``distributed_tensorflow_amd.compat``) evaluates these as fetches.
"""
from __future__ import annotations

import ctypes as C
import json
import struct
import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

from .ops import _native as nv

_lib = None


def _rt():
    global _lib
    if _lib is not None:
        return _lib
    lib = nv.runtime()
    nv.declare(lib, "dtf_events_open", [C.c_char_p, C.c_char_p, C.c_int, C.c_double], C.c_void_p)
    nv.declare(lib, "dtf_events_path", [C.c_void_p, C.c_char_p, C.c_int])
    nv.declare(lib, "dtf_events_add_scalars", [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float),
                                               C.c_int64, C.c_double])
    nv.declare(lib, "dtf_events_add_summary", [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_double])
    nv.declare(lib, "dtf_events_add_graph", [C.c_void_p, C.c_char_p, C.c_int64, C.c_double])
    nv.declare(lib, "dtf_events_add_histogram", [C.c_void_p, C.c_char_p, C.c_int64, C.c_double, C.c_double,
                                                 C.c_double, C.c_double, C.c_double, C.c_double,
                                                 C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int])
    nv.declare(lib, "dtf_events_flush", [C.c_void_p])
    nv.declare(lib, "dtf_events_close", [C.c_void_p])
    _lib = lib
    return lib


# --------------------------------------------------------------------------- Summary value container
def _varint(n: int) -> bytes:
    out = bytearray()
    while n >= 128:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


class Summary:
    """A set of (tag, simple_value) pairs; serialises to the ``tensorflow.Summary`` wire format."""

    def __init__(self, values: Optional[Sequence[Tuple[str, float]]] = None):
        self.value: List[Tuple[str, float]] = list(values or [])

    def add(self, tag: str, simple_value: float):
        self.value.append((tag, float(simple_value)))

    def SerializeToString(self) -> bytes:
        out = bytearray()
        for tag, v in self.value:
            t = tag.encode()
            body = b"\x0a" + _varint(len(t)) + t + b"\x15" + struct.pack("<f", v)
            out += b"\x0a" + _varint(len(body)) + body
        return bytes(out)

    def __repr__(self):
        return "Summary(%s)" % ", ".join("%s=%g" % tv for tv in self.value)


# --------------------------------------------------------------------------- summary ops (graph-style collection)
class ScalarSummaryOp:
    def __init__(self, name: str, source):
        self.name = name
        self.source = source   # callable -> scalar, a tensor-like with .item(), or a compat graph node

    def evaluate(self) -> float:
        v = self.source() if callable(self.source) else self.source
        if hasattr(v, "item"):
            v = v.item()
        return float(v)


class MergedSummaryOp:
    def __init__(self, ops: Sequence[ScalarSummaryOp]):
        self.ops = list(ops)

    def evaluate(self) -> Summary:
        return Summary([(op.name, op.evaluate()) for op in self.ops])


_SUMMARY_COLLECTION: List[ScalarSummaryOp] = []


def scalar(name: str, tensor) -> ScalarSummaryOp:
    """Register a scalar summary (``tf.summary.scalar``)."""
    op = ScalarSummaryOp(name, tensor)
    _SUMMARY_COLLECTION.append(op)
    return op


def merge_all() -> Optional[MergedSummaryOp]:
    """One op evaluating every summary registered so far (``tf.summary.merge_all``)."""
    return MergedSummaryOp(list(_SUMMARY_COLLECTION)) if _SUMMARY_COLLECTION else None


def merge(ops: Sequence[ScalarSummaryOp]) -> MergedSummaryOp:
    return MergedSummaryOp(ops)


def reset_collection():
    del _SUMMARY_COLLECTION[:]


# --------------------------------------------------------------------------- FileWriter
class FileWriter:
    """Asynchronous event-file writer (``tf.summary.FileWriter``)."""

    def __init__(self, logdir: str, graph=None, max_queue: int = 10, flush_secs: float = 120.0,
                 filename_suffix: str = ""):
        self.logdir = logdir
        self._h = _rt().dtf_events_open(logdir.encode(), filename_suffix.encode(), max_queue, float(flush_secs))
        if not self._h:
            raise IOError("cannot create an event file under %r" % logdir)
        buf = C.create_string_buffer(4096)
        _rt().dtf_events_path(self._h, buf, 4096)
        self.path = buf.value.decode()
        if graph is not None:
            self.add_graph(graph)

    def get_logdir(self) -> str:
        return self.logdir

    def _check(self):
        if not self._h:
            raise RuntimeError("FileWriter is closed")

    def add_summary(self, summary: Union[Summary, bytes, MergedSummaryOp], global_step=None):
        """``global_step`` is truncated to an integer, as TF does with the fractional epoch of
        /root/reference/example.py:219."""
        self._check()
        if isinstance(summary, MergedSummaryOp):
            summary = summary.evaluate()
        step = 0 if global_step is None else int(global_step)
        if isinstance(summary, Summary):
            n = len(summary.value)
            if n == 0:
                return
            tags = (C.c_char_p * n)(*[t.encode() for t, _ in summary.value])
            vals = (C.c_float * n)(*[v for _, v in summary.value])
            _rt().dtf_events_add_scalars(self._h, n, tags, vals, step, 0.0)
        else:
            data = bytes(summary)
            _rt().dtf_events_add_summary(self._h, data, len(data), step, 0.0)

    def add_scalar(self, tag: str, value: float, global_step: int = 0):
        self.add_summary(Summary([(tag, value)]), global_step)

    def add_graph(self, graph, global_step=None):
        """``writer.add_graph(sess.graph)`` (/root/reference/example.py:195): a serialized ``tensorflow.GraphDef`` in
        the event's ``graph_def`` field, which TensorBoard's graph tab parses.  ``graph`` may be a compat graph, an
        engine, a ``Sequential``, a list of ``graph_def.Node`` or serialized GraphDef bytes (see ``graph_def.nodes_for``)."""
        self._check()
        from . import graph_def as gd
        data = bytes(graph) if isinstance(graph, (bytes, bytearray)) else gd.encode_graph_def(gd.nodes_for(graph))
        _rt().dtf_events_add_graph(self._h, data, len(data), 0.0)

    def add_histogram(self, tag: str, bucket_limits: Sequence[float], bucket_counts: Sequence[float],
                      global_step: int = 0, minimum: float = 0.0, maximum: float = 0.0, total: float = 0.0,
                      sum_squares: float = 0.0):
        self._check()
        n = len(bucket_limits)
        lim = (C.c_double * n)(*bucket_limits)
        cnt = (C.c_double * n)(*bucket_counts)
        _rt().dtf_events_add_histogram(self._h, tag.encode(), int(global_step), 0.0, float(minimum), float(maximum),
                                       float(sum(bucket_counts)), float(total), float(sum_squares), lim, cnt, n)

    def flush(self):
        if self._h:
            _rt().dtf_events_flush(self._h)

    def close(self):
        h, self._h = self._h, None
        if h:
            _rt().dtf_events_close(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FileWriterCache:
    """One shared writer per log directory (the session hooks all write through it)."""

    _cache: Dict[str, FileWriter] = {}

    @classmethod
    def get(cls, logdir: str) -> FileWriter:
        w = cls._cache.get(logdir)
        if w is None or not w._h:
            w = FileWriter(logdir)
            cls._cache[logdir] = w
        return w

    @classmethod
    def clear(cls):
        for w in cls._cache.values():
            w.close()
        cls._cache.clear()


# --------------------------------------------------------------------------- reader (tools / tests)
def _masked_crc(data: bytes) -> int:
    lib = nv.runtime()
    lib.dtf_crc32c_masked.restype = C.c_uint32
    lib.dtf_crc32c_masked.argtypes = [C.c_char_p, C.c_size_t]
    return lib.dtf_crc32c_masked(data, len(data))


def read_events(path: str, verify: bool = True) -> List[Dict]:
    """Parse an event file into dicts: ``{'wall_time', 'step', 'file_version'|'graph_def'|'scalars'}``."""
    out = []
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos + 12 <= len(data):
        (length,) = struct.unpack_from("<Q", data, pos)
        (len_crc,) = struct.unpack_from("<I", data, pos + 8)
        payload = data[pos + 12:pos + 12 + length]
        if len(payload) < length or pos + 16 + length > len(data):
            break  # truncated tail (writer still running)
        (data_crc,) = struct.unpack_from("<I", data, pos + 12 + length)
        if verify:
            if _masked_crc(data[pos:pos + 8]) != len_crc or _masked_crc(payload) != data_crc:
                raise IOError("corrupt record at offset %d of %s" % (pos, path))
        out.append(_parse_event(payload))
        pos += 16 + length
    return out


def _read_varint(b: bytes, i: int) -> Tuple[int, int]:
    r, shift = 0, 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << shift
        if not c & 0x80:
            return r, i
        shift += 7


def _fields(b: bytes):
    i = 0
    while i < len(b):
        key, i = _read_varint(b, i)
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, i = _read_varint(b, i)
        elif wire == 1:
            v = b[i:i + 8]
            i += 8
        elif wire == 2:
            n, i = _read_varint(b, i)
            v = b[i:i + n]
            i += n
        elif wire == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError("unsupported wire type %d" % wire)
        yield field, wire, v


def _parse_event(payload: bytes) -> Dict:
    ev: Dict = {"step": 0}
    for field, wire, v in _fields(payload):
        if field == 1:
            ev["wall_time"] = struct.unpack("<d", v)[0]
        elif field == 2:
            ev["step"] = v
        elif field == 3:
            ev["file_version"] = v.decode()
        elif field == 4:
            ev["graph_def"] = v
        elif field == 5:
            scalars, histos = {}, {}
            for f2, _, val in _fields(v):
                if f2 != 1:
                    continue
                tag, simple, histo = None, None, None
                for f3, _, x in _fields(val):
                    if f3 == 1:
                        tag = x.decode()
                    elif f3 == 2:
                        simple = struct.unpack("<f", x)[0]
                    elif f3 == 5:
                        histo = x
                if simple is not None:
                    scalars[tag] = simple
                if histo is not None:
                    h = {}
                    for f4, _, x in _fields(histo):
                        if f4 in (1, 2, 3, 4, 5):
                            h[{1: "min", 2: "max", 3: "num", 4: "sum", 5: "sum_squares"}[f4]] = struct.unpack("<d", x)[0]
                        elif f4 in (6, 7):
                            h["bucket_limit" if f4 == 6 else "bucket"] = list(struct.unpack("<%dd" % (len(x) // 8), x))
                    histos[tag] = h
            ev["scalars"] = scalars
            if histos:
                ev["histograms"] = histos
    return ev

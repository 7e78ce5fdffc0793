"""Host-resident data sets streamed to the GPU: Python face of ``csrc/runtime/input_pipeline.cc``.

The reference feeds every step from Python lists through ``feed_dict`` (/root/reference/example.py:209-213).  When a
data set fits in HBM, ``Sequential.fit`` keeps it on the device and shuffles there; when it does not (or when the
caller wants it to stay on the host), :class:`HostPipeline` gathers shuffled mini-batches with C++ worker threads
into a ring of pinned staging buffers and copies them to the device on a separate HIP stream, ``depth`` batches
ahead of the training step that consumes them.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterator, Optional, Tuple

import torch

from .ops import _native as nv

_lib = None


def _rt():
    global _lib
    if _lib is None:
        lib = nv.runtime()
        i64, p = C.c_int64, C.c_void_p
        nv.declare(lib, "dtf_pipe_create", [p, p, i64, i64, i64, i64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(p), C.POINTER(p)], p)
        nv.declare(lib, "dtf_pipe_start_epoch", [p, i64], i64)
        nv.declare(lib, "dtf_pipe_next", [p, C.POINTER(i64), C.POINTER(i64)])
        nv.declare(lib, "dtf_pipe_release", [p, i64], None)
        nv.declare(lib, "dtf_pipe_perm_at", [p, i64], i64)
        nv.declare(lib, "dtf_pipe_destroy", [p], None)
        _lib = lib
    return _lib


def _host_tensor(a) -> torch.Tensor:
    t = a if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    if t.device.type != "cpu":
        raise ValueError("HostPipeline streams host-resident data; this tensor lives on %s" % t.device)
    if t.dtype == torch.float64:
        t = t.float()
    return t.contiguous()


class HostPipeline:
    """Iterate ``(x_batch, y_batch)`` device tensors of a host data set, epoch by epoch.

    ``depth`` staging slots (>= 2) are filled ahead by ``threads`` C++ workers; each batch is copied to ``device``
    on a side stream and handed out together with the event the consumer's stream must wait for, which
    :meth:`batches` does before yielding.  The order of epoch ``e`` depends only on ``(seed, e)``.
    """

    def __init__(self, x, y=None, batch_size: int = 32, shuffle: bool = True, seed: int = 0, drop_last: bool = False,
                 device=None, depth: int = 4, threads: int = 2):
        self.x = _host_tensor(x)
        self.y = _host_tensor(y) if y is not None else None
        if self.y is not None and self.y.shape[0] != self.x.shape[0]:
            raise ValueError("x and y disagree on the number of rows")
        self.rows = int(self.x.shape[0])
        self.batch_size = int(batch_size)
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.depth = max(2, int(depth))
        pin = self.device.type == "cuda"
        self._xs = [torch.empty((self.batch_size,) + tuple(self.x.shape[1:]), dtype=self.x.dtype, pin_memory=pin)
                    for _ in range(self.depth)]
        self._ys = [torch.empty((self.batch_size,) + tuple(self.y.shape[1:]), dtype=self.y.dtype, pin_memory=pin)
                    for _ in range(self.depth)] if self.y is not None else []
        xrow = self.x[0].numel() * self.x.element_size() if self.rows else 0
        yrow = self.y[0].numel() * self.y.element_size() if self.y is not None else 0
        xp = (C.c_void_p * self.depth)(*[t.data_ptr() for t in self._xs])
        yp = (C.c_void_p * self.depth)(*[t.data_ptr() for t in self._ys]) if self._ys else None
        self._h = _rt().dtf_pipe_create(self.x.data_ptr(), self.y.data_ptr() if self.y is not None else None,
                                        self.rows, xrow, yrow, self.batch_size, int(shuffle), int(seed) & (2**64 - 1),
                                        int(drop_last), self.depth, max(1, int(threads)), xp, yp)
        if not self._h:
            raise ValueError("invalid input pipeline configuration")
        self._copy_stream = torch.cuda.Stream(device=self.device) if pin else None
        self._pending = []          # (slot, event) whose device copies may still be running
        self.num_batches = 0

    def start_epoch(self, epoch: int) -> int:
        self._drain()
        self.num_batches = int(_rt().dtf_pipe_start_epoch(self._h, int(epoch)))
        return self.num_batches

    def order(self, i: int) -> int:
        """Row of the data set at position ``i`` of the running epoch's order."""
        return int(_rt().dtf_pipe_perm_at(self._h, int(i)))

    def _drain(self):
        for slot, ev in self._pending:
            if ev is not None:
                ev.synchronize()
            _rt().dtf_pipe_release(self._h, slot)
        self._pending = []

    def _retire(self, keep: int):
        """Hand back the slots whose copies have completed, oldest first, keeping at most ``keep`` outstanding."""
        while self._pending and (len(self._pending) > keep or (self._pending[0][1] is None or self._pending[0][1].query())):
            slot, ev = self._pending.pop(0)
            if ev is not None:
                ev.synchronize()
            _rt().dtf_pipe_release(self._h, slot)

    def next_batch(self) -> Optional[Tuple[torch.Tensor, Optional[torch.Tensor]]]:
        slot, rows = C.c_int64(), C.c_int64()
        st = _rt().dtf_pipe_next(self._h, C.byref(slot), C.byref(rows))
        if st != 0:
            return None
        s, n = slot.value, rows.value
        if self._copy_stream is None:
            xb = self._xs[s][:n].clone()
            yb = self._ys[s][:n].clone() if self._ys else None
            _rt().dtf_pipe_release(self._h, s)
            return xb, yb
        with torch.cuda.stream(self._copy_stream):
            xb = self._xs[s][:n].to(self.device, non_blocking=True)
            yb = self._ys[s][:n].to(self.device, non_blocking=True) if self._ys else None
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        torch.cuda.current_stream(self.device).wait_event(ev)
        xb.record_stream(torch.cuda.current_stream(self.device))
        if yb is not None:
            yb.record_stream(torch.cuda.current_stream(self.device))
        self._pending.append((s, ev))
        self._retire(self.depth - 2)     # the producers always have at least two slots to work in
        return xb, yb

    def batches(self, epoch: int) -> Iterator[Tuple[torch.Tensor, Optional[torch.Tensor]]]:
        self.start_epoch(epoch)
        while True:
            b = self.next_batch()
            if b is None:
                return
            yield b

    def close(self):
        h, self._h = self._h, None
        if h:
            for _, ev in self._pending:
                if ev is not None:
                    ev.synchronize()
            self._pending = []
            _rt().dtf_pipe_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

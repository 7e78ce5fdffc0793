"""One-shot all-reduce over peer pointers for the ranks of one node (``csrc/kernels/p2p_allreduce.hip``).

The gradient of the reference network is 116 KB (/root/reference/example.py:149-154 -> 28 960 parameters) and that of
the MNIST-shaped MLP 470 KB: a ring all-reduce of such a message is 2(N-1) dependent latencies.  Here every rank
publishes its contribution in a buffer that the other processes have mapped (hipIpc), raises a flag in every peer's
memory, and then adds up all N buffers itself, reading the peers over xGMI - one hop, and the same summation order on
every rank, so replicated parameters stay bit-identical.

The IPC handles travel once through ``torch.distributed.all_gather_object`` (any backend).  Select it for the
replicated synchronous mode with ``DTF_ALLREDUCE=p2p``; the default remains the RCCL all-reduce.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

from ..ops import _native as nv

_lib = None


def _k():
    global _lib
    if _lib is None:
        lib = nv.kernels()
        p, i64 = C.c_void_p, C.c_int64
        nv.declare(lib, "dtf_p2p_create", [C.c_int, C.c_int, i64], p)
        nv.declare(lib, "dtf_p2p_handle_bytes", [])
        nv.declare(lib, "dtf_p2p_is_fine_grained", [p])
        nv.declare(lib, "dtf_p2p_local_handle", [p, p])
        nv.declare(lib, "dtf_p2p_open_peers", [p, p])
        nv.declare(lib, "dtf_p2p_allreduce", [p, p, i64, p])
        nv.declare(lib, "dtf_p2p_error", [p])
        nv.declare(lib, "dtf_p2p_set_timeout_ms", [p, i64], None)
        nv.declare(lib, "dtf_p2p_destroy", [p], None)
        _lib = lib
    return _lib


class P2PTimeoutError(RuntimeError):
    """A rank of the node never took part in a peer-pointer all-reduce."""


class P2PAllReduce:
    """``capacity`` floats can be reduced per call.  Construction is collective over ``group``."""

    def __init__(self, capacity: int, device, group=None, timeout_s: Optional[float] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group if group is not None else dist.group.WORLD
        self.rank = dist.get_rank(self.group)
        self.world = dist.get_world_size(self.group)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("the peer-pointer all-reduce lives on the GPU")
        lib = _k()
        with torch.cuda.device(self.device):
            self._h = lib.dtf_p2p_create(self.rank, self.world, int(capacity))
            if not self._h:
                raise RuntimeError("cannot allocate the exchange buffer (%d ranks, %d floats)" % (self.world, capacity))
            nb = lib.dtf_p2p_handle_bytes()
            mine = C.create_string_buffer(nb)
            rc = lib.dtf_p2p_local_handle(self._h, mine)
            if rc:
                raise RuntimeError("hipIpcGetMemHandle failed (%d)" % rc)
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(mine.raw), group=self.group)
            rc = lib.dtf_p2p_open_peers(self._h, C.create_string_buffer(b"".join(handles), nb * self.world))
            if rc:
                raise RuntimeError("hipIpcOpenMemHandle failed (%d)" % rc)
        self.capacity = int(capacity)
        import os
        timeout_s = float(os.environ.get("DTF_P2P_TIMEOUT_SECS", "30")) if timeout_s is None else timeout_s
        lib.dtf_p2p_set_timeout_ms(self._h, int(timeout_s * 1000))
        self.timeout_s = timeout_s
        self.fine_grained = bool(lib.dtf_p2p_is_fine_grained(self._h))
        dist.barrier(group=self.group) if dist.get_backend(self.group) != "nccl" else \
            dist.barrier(group=self.group, device_ids=[self.device.index])

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        """In place, on the current stream; every rank calls with the same length in the same order."""
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("expects a contiguous float32 GPU tensor")
        rc = _k().dtf_p2p_allreduce(self._h, t.data_ptr(), t.numel(), torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            raise RuntimeError("peer-pointer all-reduce refused the call (%d)" % rc)

    def check(self) -> None:
        """Raise if a reduce gave up waiting for a peer (synchronises the device: for occasional use).  The output of
        such a reduce is all NaN; the error is permanent - the rank must be restarted with its gang."""
        e = _k().dtf_p2p_error(self._h)
        if e:
            raise P2PTimeoutError("peer-pointer all-reduce: a peer did not publish call %d within %.1f s; its result "
                                  "is NaN and this rank cannot continue" % (e, self.timeout_s))

    def close(self):
        h, self._h = self._h, None
        if h:
            _k().dtf_p2p_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""Peer-mapped memory regions: one-sided ``put`` / ``get`` between the ranks of one node.

The data plane of the asynchronous parameter-server mode when owners live inside the worker processes (8 workers on
8 GPUs, variables sharded over all eight HBMs).  Each rank allocates one region, publishes a handle through the
control plane and maps everybody else's region:

* on GPUs the region is device memory shared through hipIpc (``csrc/kernels/peer_mailbox.hip``); a ``put`` is one
  device-to-device copy over xGMI on the caller's current stream and ``sync()`` waits for that stream;
* on CPUs (tests, the reference's "localhost plumbing" configuration) it is a POSIX shared-memory segment and a
  ``put`` is a memcpy.

Either way a message needs no rendezvous with the receiving process - the sender writes, synchronises, and then tells
the receiver over the control plane - which is what lets owners serve workers in arrival order without any of the
ordering hazards of two threads sharing a send/recv communicator.  The reference moves the same tensors through
gRPC ``RecvTensor`` (/root/reference/example.py:128-129).
"""
from __future__ import annotations

import base64
import ctypes as C
import os
import uuid
from typing import List, Optional

import torch

from ..ops import _native as nv

_lib = None
_CREATED_HERE = set()       # shared-memory segments this process created (several ranks as threads: tests)


def _k():
    global _lib
    if _lib is None:
        lib = nv.kernels()
        p, ll = C.c_void_p, C.c_longlong
        nv.declare(lib, "dtf_mbox_create", [C.c_int, C.c_int, ll], p)
        nv.declare(lib, "dtf_mbox_handle_bytes", [])
        nv.declare(lib, "dtf_mbox_bytes", [p], ll)
        nv.declare(lib, "dtf_mbox_local_ptr", [p], p)
        nv.declare(lib, "dtf_mbox_peer_ptr", [p, C.c_int], p)
        nv.declare(lib, "dtf_mbox_local_handle", [p, p])
        nv.declare(lib, "dtf_mbox_open_peer", [p, C.c_int, p])
        nv.declare(lib, "dtf_mbox_reopen_peer", [p, C.c_int, p])
        nv.declare(lib, "dtf_mbox_set_peer_bytes", [p, C.c_int, ll])
        nv.declare(lib, "dtf_mbox_put", [p, C.c_int, ll, p, ll, p])
        nv.declare(lib, "dtf_mbox_get", [p, C.c_int, ll, p, ll, p])
        nv.declare(lib, "dtf_mbox_gather", [p, ll, p, ll, p])
        nv.declare(lib, "dtf_mbox_scatter", [p, ll, p, ll, p])
        nv.declare(lib, "dtf_mbox_ranges_create", [p, p, C.c_int], p)
        nv.declare(lib, "dtf_mbox_ranges_destroy", [p], None)
        nv.declare(lib, "dtf_mbox_ranges_run", [p, p, C.c_int, ll, p, ll, p])
        nv.declare(lib, "dtf_mbox_sync", [p])
        nv.declare(lib, "dtf_mbox_destroy", [p], None)
        _lib = lib
    return _lib


class _RangeDesc(C.Structure):
    _fields_ = [("peer", C.c_int), ("pad", C.c_int), ("remote_off", C.c_longlong), ("local_off", C.c_longlong),
                ("n16", C.c_longlong)]


class RangeTable:
    """A fixed list of ``(peer, remote_off, local_off, n)`` ranges (floats) between a local buffer and the peers'
    regions, moved by one kernel launch (``PeerRegion.move_ranges``)."""

    def __init__(self, ranges, handle=None):
        self.ranges = [tuple(int(v) for v in r) for r in ranges]
        self.handle = handle


class _DeviceMemory:
    """Lets ``torch.as_tensor`` alias raw device memory (the CUDA array interface)."""

    def __init__(self, ptr: int, nfloats: int):
        self.__cuda_array_interface__ = {"shape": (int(nfloats),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class PeerRegion:
    """``nfloats`` float32 of memory per rank, every rank's region addressable from every rank.

    Construction is collective over the ``world`` ranks that share ``name`` on the control plane ``cp`` (handles are
    exchanged through its key/value store; nobody returns before all regions are mapped).
    """

    def __init__(self, rank: int, world: int, nfloats: int, device, cp, name: str = "mbox", timeout_s: float = 300.0,
                 sizes: Optional[List[int]] = None, rejoin: bool = False):
        # `sizes`: floats per rank when the regions differ (ps tasks hold variables, slots and inboxes; their workers
        # hold nothing and allocate a token region); every rank passes the same list.  Default: `nfloats` everywhere.
        self.sizes = [int(nfloats)] * int(world) if sizes is None else [max(int(s), 1) for s in sizes]
        if len(self.sizes) != int(world):
            raise ValueError("one region size per rank")
        self.rank, self.world, self.nfloats = int(rank), int(world), self.sizes[int(rank)]
        self.device = torch.device(device)
        self.name = name
        self._h = None
        self._shm = None
        self._peers_shm: List = []
        self._peer_views: List[Optional[torch.Tensor]] = [None] * self.world
        self._tables: List[RangeTable] = []
        # DTF_PEER_KERNELS=0: gather / scatter fall back to one runtime copy per peer (hipMemcpyAsync) instead of the
        # kernels that load from / store to peer-mapped memory directly
        self._kernels = os.environ.get("DTF_PEER_KERNELS", "1") != "0"
        if self.device.type == "cuda":
            self._init_gpu(cp, timeout_s)
        else:
            self._init_cpu(cp, timeout_s)
        # `rejoin`: a restarted process takes the place of a rank whose peers mapped everything long ago - it publishes
        # its new handle, maps the others from theirs and waits for nobody
        if not rejoin:
            cp.barrier("%s/mapped" % name, self.world, timeout_s=timeout_s)

    # ------------------------------------------------------------------ GPU: hipIpc
    def _init_gpu(self, cp, timeout_s):
        lib = _k()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with torch.cuda.device(self.device):
            self._h = lib.dtf_mbox_create(self.rank, self.world, self.nfloats * 4)
            if not self._h:
                raise RuntimeError("cannot allocate a %d-float peer region on %s" % (self.nfloats, self.device))
            nb = lib.dtf_mbox_handle_bytes()
            mine = C.create_string_buffer(nb)
            rc = lib.dtf_mbox_local_handle(self._h, mine)
            if rc:
                raise RuntimeError("hipIpcGetMemHandle failed (%d)" % rc)
            cp.set("%s/handle/%d" % (self.name, self.rank), base64.b64encode(bytes(mine.raw)))
            for r in range(self.world):
                if r == self.rank:
                    continue
                raw = base64.b64decode(cp.get("%s/handle/%d" % (self.name, r), timeout_s=timeout_s))
                rc = lib.dtf_mbox_open_peer(self._h, r, C.create_string_buffer(raw, nb))
                if rc:
                    raise RuntimeError("hipIpcOpenMemHandle of rank %d's region failed (%d)" % (r, rc))
                lib.dtf_mbox_set_peer_bytes(self._h, r, self.sizes[r] * 4)
            self.local = torch.as_tensor(_DeviceMemory(lib.dtf_mbox_local_ptr(self._h), self.nfloats), device=self.device)

    # ------------------------------------------------------------------ CPU: POSIX shared memory
    def _init_cpu(self, cp, timeout_s):
        from multiprocessing import resource_tracker, shared_memory
        seg = "dtf_%s_%d" % (uuid.uuid4().hex[:12], self.rank)
        _CREATED_HERE.add(seg)
        self._shm = shared_memory.SharedMemory(name=seg, create=True, size=max(self.nfloats * 4, 4))
        self.local = torch.frombuffer(self._shm.buf, dtype=torch.float32, count=self.nfloats)
        self.local.zero_()
        cp.set("%s/handle/%d" % (self.name, self.rank), seg.encode())
        self._peer_views[self.rank] = self.local
        for r in range(self.world):
            if r == self.rank:
                continue
            peer_name = cp.get("%s/handle/%d" % (self.name, r), timeout_s=timeout_s).decode()
            shm = shared_memory.SharedMemory(name=peer_name)
            if shm._name.lstrip("/") not in _CREATED_HERE:
                try:    # the attaching process must not unlink the creator's segment when it exits
                    resource_tracker.unregister(shm._name, "shared_memory")
                except Exception:
                    pass
            self._peers_shm.append(shm)
            self._peer_views[r] = torch.frombuffer(shm.buf, dtype=torch.float32, count=self.sizes[r])

    # ------------------------------------------------------------------ data movement
    def _check(self, t: torch.Tensor, off: int, peer: Optional[int] = None):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("peer regions move contiguous float32 tensors")
        limit = min(self.sizes) if peer is None else self.sizes[peer]
        if off < 0 or off + t.numel() > limit:
            raise ValueError("range [%d, %d) outside the %d-float region" % (off, off + t.numel(), limit))

    def put(self, peer: int, dst_off: int, src: torch.Tensor):
        """``region[peer][dst_off : dst_off + src.numel()] = src`` (GPU: asynchronously, on the current stream)."""
        self._check(src, dst_off, peer)
        if src.numel() == 0:
            return
        if self._h is not None:
            rc = _k().dtf_mbox_put(self._h, int(peer), int(dst_off) * 4, src.data_ptr(), src.numel() * 4,
                                   torch.cuda.current_stream(self.device).cuda_stream)
            if rc:
                raise RuntimeError("peer put to rank %d failed (%d)" % (peer, rc))
        else:
            self._peer_views[peer][dst_off:dst_off + src.numel()].copy_(src.reshape(-1))

    def get(self, peer: int, src_off: int, dst: torch.Tensor):
        """``dst = region[peer][src_off : src_off + dst.numel()]``."""
        self._check(dst, src_off, peer)
        if dst.numel() == 0:
            return
        if self._h is not None:
            rc = _k().dtf_mbox_get(self._h, int(peer), int(src_off) * 4, dst.data_ptr(), dst.numel() * 4,
                                   torch.cuda.current_stream(self.device).cuda_stream)
            if rc:
                raise RuntimeError("peer get from rank %d failed (%d)" % (peer, rc))
        else:
            dst.reshape(-1).copy_(self._peer_views[peer][src_off:src_off + dst.numel()])

    def gather(self, src_off: int, dst: torch.Tensor, stream_ptr: Optional[int] = None):
        """``dst[j*n:(j+1)*n] = region[j][src_off : src_off + n]`` for every rank ``j`` (``n = dst.numel() / world``).
        On GPUs one kernel launch reads from all peers at once, so every xGMI link of this GPU carries its shard
        concurrently; a chain of per-peer copies on one stream would use the point-to-point links one at a time."""
        n = dst.numel() // self.world
        if dst.numel() != n * self.world:
            raise ValueError("gather needs a destination of world * n elements")
        self._check(dst[:n], src_off)
        if n == 0:
            return
        if self._h is not None and n % 4 == 0 and src_off % 4 == 0 and self._kernels:
            rc = _k().dtf_mbox_gather(self._h, int(src_off) * 4, dst.data_ptr(), n * 4,
                                      stream_ptr if stream_ptr is not None
                                      else torch.cuda.current_stream(self.device).cuda_stream)
            if rc:
                raise RuntimeError("peer gather failed (%d)" % rc)
            return
        for j in range(self.world):
            self.get(j, src_off, dst[j * n:(j + 1) * n])

    def scatter(self, dst_off: int, src: torch.Tensor):
        """``region[j][dst_off : dst_off + n] = src[j*n:(j+1)*n]`` for every rank ``j``: the one-launch counterpart of
        ``gather`` (stores to all peers in flight at once)."""
        n = src.numel() // self.world
        if src.numel() != n * self.world:
            raise ValueError("scatter needs a source of world * n elements")
        self._check(src[:n], dst_off)
        if n == 0:
            return
        if self._h is not None and n % 4 == 0 and dst_off % 4 == 0 and self._kernels:
            rc = _k().dtf_mbox_scatter(self._h, int(dst_off) * 4, src.data_ptr(), n * 4,
                                       torch.cuda.current_stream(self.device).cuda_stream)
            if rc:
                raise RuntimeError("peer scatter failed (%d)" % rc)
            return
        for j in range(self.world):
            self.put(j, dst_off, src[j * n:(j + 1) * n])

    def peer_address(self, peer: int, off: int = 0) -> Optional[int]:
        """Device address of float ``off`` of rank ``peer``'s region as this process's kernels see it (GPU only)."""
        if self._h is None:
            return None
        if not 0 <= off <= self.sizes[peer]:
            raise ValueError("offset %d outside rank %d's region" % (off, peer))
        base = _k().dtf_mbox_peer_ptr(self._h, int(peer))
        return None if not base else int(base) + int(off) * 4

    def remap_peer(self, peer: int, cp, timeout_s: float = 60.0):
        """Rank ``peer`` was restarted and has published the handle of a new region: map that one instead of the dead
        process's (GPU regions; a CPU segment is mapped anew the same way)."""
        if self._h is not None:
            lib = _k()
            nb = lib.dtf_mbox_handle_bytes()
            raw = base64.b64decode(cp.get("%s/handle/%d" % (self.name, peer), timeout_s=timeout_s))
            with torch.cuda.device(self.device):
                rc = lib.dtf_mbox_reopen_peer(self._h, int(peer), C.create_string_buffer(raw, nb))
            if rc:
                raise RuntimeError("hipIpcOpenMemHandle of rank %d's new region failed (%d)" % (peer, rc))
            lib.dtf_mbox_set_peer_bytes(self._h, int(peer), self.sizes[peer] * 4)
            return
        from multiprocessing import resource_tracker, shared_memory
        shm = shared_memory.SharedMemory(name=cp.get("%s/handle/%d" % (self.name, peer), timeout_s=timeout_s).decode())
        if shm._name.lstrip("/") not in _CREATED_HERE:
            try:
                resource_tracker.unregister(shm._name, "shared_memory")
            except Exception:
                pass
        self._peers_shm.append(shm)
        self._peer_views[peer] = torch.frombuffer(shm.buf, dtype=torch.float32, count=self.sizes[peer])

    def make_ranges(self, ranges) -> RangeTable:
        """Prepare ``ranges`` = ``[(peer, remote_off, local_off, n), ...]`` (floats) for ``move_ranges``."""
        t = RangeTable(ranges)
        for peer, roff, loff, n in t.ranges:
            if not 0 <= peer < self.world or roff < 0 or loff < 0 or n < 0 or roff + n > self.sizes[peer]:
                raise ValueError("range (%d, %d, %d, %d) outside rank %d's region" % (peer, roff, loff, n, peer))
        vec = all(roff % 4 == 0 and loff % 4 == 0 and n % 4 == 0 for _, roff, loff, n in t.ranges)
        if self._h is not None and self._kernels and vec and t.ranges:
            arr = (_RangeDesc * len(t.ranges))()
            for d, (peer, roff, loff, n) in zip(arr, t.ranges):
                d.peer, d.remote_off, d.local_off, d.n16 = peer, roff * 4, loff * 4, n // 4
            with torch.cuda.device(self.device):
                t.handle = _k().dtf_mbox_ranges_create(self._h, arr, len(t.ranges))
            if not t.handle:
                raise RuntimeError("cannot build the range table")
            self._tables.append(t)
        return t

    def move_ranges(self, table: RangeTable, gather: bool, remote_base: int, local: torch.Tensor,
                    stream_ptr: Optional[int] = None):
        """Every range of ``table`` at once: ``local[local_off:+n] = region[peer][remote_base + remote_off:+n]`` when
        ``gather``, the other direction otherwise (GPU: one launch on the current stream, all peers in flight)."""
        if local.dtype != torch.float32 or not local.is_contiguous():
            raise ValueError("peer regions move contiguous float32 tensors")
        if table.handle and remote_base % 4 == 0:
            rc = _k().dtf_mbox_ranges_run(self._h, table.handle, 1 if gather else 0, int(remote_base) * 4, local.data_ptr(),
                                          local.numel() * 4, stream_ptr if stream_ptr is not None
                                          else torch.cuda.current_stream(self.device).cuda_stream)
            if rc:
                raise RuntimeError("peer range %s failed (%d)" % ("gather" if gather else "scatter", rc))
            return
        flat = local.reshape(-1)
        for peer, roff, loff, n in table.ranges:
            if gather:
                self.get(peer, remote_base + roff, flat[loff:loff + n])
            else:
                self.put(peer, remote_base + roff, flat[loff:loff + n])

    def sync(self):
        """Wait until every put / get issued on the current stream has completed (and is visible to its target)."""
        if self._h is not None:
            rc = _k().dtf_mbox_sync(torch.cuda.current_stream(self.device).cuda_stream)
            if rc:
                raise RuntimeError("stream synchronisation failed (%d)" % rc)

    def close(self):
        h, self._h = self._h, None
        if h:
            self.local = None
            with torch.cuda.device(self.device):
                for t in self._tables:
                    if t.handle:
                        _k().dtf_mbox_ranges_destroy(t.handle)
                        t.handle = None
                self._tables = []
                _k().dtf_mbox_destroy(h)
        if self._shm is not None or self._peers_shm:
            # tensors made by frombuffer pin the mappings: drop them before closing
            self.local = None
            self._peer_views = [None] * self.world
            for shm in self._peers_shm:
                try:
                    shm.close()
                except Exception:
                    pass
            self._peers_shm = []
            if self._shm is not None:
                try:
                    self._shm.close()
                except Exception:
                    pass
                try:
                    self._shm.unlink()
                except Exception:
                    pass
                self._shm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""Doorbell words and the native owner pump of the asynchronous parameter-server mode.

Every owner has one *control block*: a header (``version``, ``global_step``) and one 128-byte *lane* per worker with the
worker's doorbell / clock and the owner's acknowledgement.  Workers and owners exchange gradients and parameters through
peer-mapped device memory (:mod:`.peer_region`); these words are all that tells the other side that data is complete, so
a training step makes no control-plane round trip and never synchronises a stream on the host.  The layout is the one of
``csrc/kernels/owner_pump.hip`` (``Ctl`` / ``Lane``), which also holds the owner's service thread.

On GPUs the blocks live as replicas in every rank's own device memory and are written by kernels in stream order
(:class:`ControlBlocks`, replica mode); a GPU worker's whole push is one native call (:class:`PushLink`).  On CPUs - and
where a GPU's memory is not addressable by its host - they are POSIX shared memory written by host code.

The reference's counterpart is TF's gRPC worker service inside ``server.join()``
(/root/reference/example.py:128-131): it executes ``ApplyAdam`` for every gradient that arrives.
"""
from __future__ import annotations

import ctypes as C
import time
import uuid
from typing import List, Optional

import numpy as np
import torch

from ..ops import _native as nv

MAX_DEPTH = 4
HIST = 64
# word indices inside a lane (uint64 units) and inside the header
_DOOR, _CLOCK, _PULLV, _ACK, _ACKV = 0, 1, 2, 8, 9
_LANE_WORDS = 16
_HDR_WORDS = 16
_VERSION, _GSTEP, _GSBASE = 0, 1, 2

_lib = None
_CREATED_HERE = set()       # shared-memory segments this process created (several ranks as threads: tests)


def _k():
    global _lib
    if _lib is None:
        lib = nv.kernels()
        p, ll, ull = C.c_void_p, C.c_longlong, C.c_ulonglong
        nv.declare(lib, "dtf_pump_config_bytes", [])
        nv.declare(lib, "dtf_pump_lane_bytes", [])
        nv.declare(lib, "dtf_pump_ctl_bytes", [C.c_int], ll)
        nv.declare(lib, "dtf_pump_start", [p], p)
        nv.declare(lib, "dtf_pump_pause", [p])
        nv.declare(lib, "dtf_pump_resume", [p, C.c_int])
        nv.declare(lib, "dtf_pump_error", [p])
        nv.declare(lib, "dtf_pump_set_replica", [p, C.c_int, p])
        nv.declare(lib, "dtf_pump_stream", [p], p)
        nv.declare(lib, "dtf_pump_histogram", [p, p], None)
        nv.declare(lib, "dtf_pump_stop", [p])
        nv.declare(lib, "dtf_doorbell_ring", [p, C.c_int, p, p, C.c_int, p, p, C.c_int, C.c_int, ull, ull])
        nv.declare(lib, "dtf_words_broadcast", [p, C.c_int, ll, ull, p])
        nv.declare(lib, "dtf_device_large_bar", [])
        nv.declare(lib, "dtf_link_create", [], p)
        nv.declare(lib, "dtf_link_destroy", [p], None)
        nv.declare(lib, "dtf_link_stream", [p], p)
        nv.declare(lib, "dtf_link_push", [p, p, p, p, p, ll, ll, ll, p, p, C.c_int, p, p, C.c_int, C.c_int, ull, ull])
        assert lib.dtf_pump_lane_bytes() == _LANE_WORDS * 8
        _lib = lib
    return _lib


class _PumpConfig(C.Structure):
    _fields_ = [("on_gpu", C.c_int), ("owner", C.c_int), ("workers", C.c_int), ("depth", C.c_int),
                ("optimizer", C.c_int), ("pad", C.c_int),
                ("n", C.c_longlong), ("slot_stride", C.c_longlong),
                ("var", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("inbox", C.c_void_p),
                ("step", C.c_void_p), ("beta_powers", C.c_void_p), ("ticket", C.c_void_p),
                ("lr", C.c_float), ("b1", C.c_float), ("b2", C.c_float), ("eps", C.c_float), ("momentum", C.c_float),
                ("pad2", C.c_int), ("ctl", C.c_void_p), ("rep_base", C.c_void_p * 64), ("rep_world", C.c_int),
                ("pad3", C.c_int), ("block_word", C.c_longlong)]


class ControlBlocks:
    """The control blocks of all owners, mapped into this process.  Construction is collective over the ``world``
    ranks sharing ``name`` on the control plane ``cp``: rank ``r`` creates one block with ``lanes`` lanes (default
    ``world``: every rank is a worker and an owner).  ``owner_rank(j)`` is the rank whose block owner ``j`` uses
    (default ``j``); with ps tasks the owners are the first ranks and the lanes count the workers only.  All methods
    take owner and worker *indices*.

    Two homes for the words:

    * POSIX shared memory, one segment per block, written by host code (CPUs; GPUs fall back to it, with the writes in
      stream host functions);
    * *replica mode* (``region`` given, a GPU whose memory the CPU can address, ``DTF_DOORBELL_KERNELS`` not 0): every
      rank holds a replica of ALL blocks in the last ``area_floats(world, lanes)`` floats of its peer region - its own
      device memory, which its CPU reads through the large BAR - and every update is stored into all replicas by a
      kernel (``ring`` / the pump's acknowledgement / ``_broadcast``), in stream order behind the data it announces.
      A worker writes only into the owners' replicas and its own (``ring_array``); the owners write into everybody's,
      and they are the ones that map a restarted worker's new region (``replace_replica`` / ``copy_block_to``), which
      is why the workers' clocks are published by owner 0 with its acknowledgements rather than by the workers."""

    @staticmethod
    def area_floats(world: int, lanes: Optional[int] = None) -> int:
        """Floats a rank's region must reserve at its end for replica mode."""
        return 2 * int(world) * (_HDR_WORDS + _LANE_WORDS * int(world if lanes is None else lanes))

    @staticmethod
    def replica_mode_available(device) -> bool:
        import os
        if torch.device(device).type != "cuda" or os.environ.get("DTF_DOORBELL_KERNELS", "1") == "0":
            return False
        with torch.cuda.device(device):
            return bool(_k().dtf_device_large_bar())

    def __init__(self, rank: int, world: int, cp, name: str = "co/ctl", timeout_s: float = 300.0,
                 lanes: Optional[int] = None, owner_rank=None, rejoin: bool = False, region=None):
        from multiprocessing import resource_tracker, shared_memory
        self.rank, self.world = int(rank), int(world)
        self.lanes = self.world if lanes is None else int(lanes)
        self._blk = owner_rank if owner_rank is not None else (lambda j: j)
        self._block_words = _HDR_WORDS + _LANE_WORDS * self.lanes
        nbytes = self._block_words * 8
        self._nbytes = nbytes
        self._shms: List = []
        self._mine = None
        self.replicas: Optional[List[int]] = None       # replica mode: every rank's control area as this GPU addresses it
        self._rep_array = None
        if region is not None:
            # replica mode: the areas were zeroed with the regions, and every region is mapped when its constructor returns
            area = self.area_floats(self.world, self.lanes)
            self.replicas = [region.peer_address(r, region.sizes[r] - area) for r in range(self.world)]
            if not all(self.replicas):
                raise RuntimeError("a peer region is not mapped")
            self._rep_array = (C.c_void_p * self.world)(*self.replicas)
            self._device = region.device
            self._region = region
            # what a worker's doorbell is written into: the replicas of the owners (who poll it) and its own (which
            # keeps the version the gradient was computed from for the staleness histogram); ranks in table order
            self._ring_ranks = None          # set by use_ring_ranks(); default: every replica
            local = self.replicas[self.rank]
            self._base = [local + k * nbytes for k in range(self.world)]
            self._views = [(C.c_uint64 * self._block_words).from_address(b) for b in self._base]
            self.words = [np.ctypeslib.as_array(v) for v in self._views]
            if not rejoin:
                cp.barrier("%s/mapped" % name, self.world, timeout_s=timeout_s)
            return
        seg = "dtfc_%s_%d" % (uuid.uuid4().hex[:12], self.rank)
        _CREATED_HERE.add(seg)
        self._mine = shared_memory.SharedMemory(name=seg, create=True, size=nbytes)
        self._mine.buf[:nbytes] = bytes(nbytes)
        cp.set("%s/handle/%d" % (name, self.rank), seg.encode())
        self._shms = [None] * self.world
        self._shms[self.rank] = self._mine
        for r in range(self.world):
            if r == self.rank:
                continue
            peer = cp.get("%s/handle/%d" % (name, r), timeout_s=timeout_s).decode()
            shm = shared_memory.SharedMemory(name=peer)
            if shm._name.lstrip("/") not in _CREATED_HERE:
                try:    # the attaching process must not unlink the creator's segment when it exits
                    resource_tracker.unregister(shm._name, "shared_memory")
                except Exception:
                    pass
            self._shms[r] = shm
        self.words = [np.frombuffer(s.buf, dtype=np.uint64, count=nbytes // 8) for s in self._shms]
        self._base = [C.addressof(C.c_char.from_buffer(s.buf)) for s in self._shms]
        if not rejoin:      # a restarted process maps the blocks of ranks that have been running for a while
            cp.barrier("%s/mapped" % name, self.world, timeout_s=timeout_s)

    # ---- replica mode: which replicas a worker's doorbell goes to, and re-admission of a restarted worker
    def use_ring_ranks(self, ranks):
        """Restrict this worker's doorbell writes to the replicas of ``ranks`` (the owners' and its own)."""
        self._ring_ranks = sorted(set(int(r) for r in ranks))
        self._ring_array = (C.c_void_p * len(self._ring_ranks))(*[self.replicas[r] for r in self._ring_ranks])

    def ring_array(self):
        """(array of replica bases, count, index of this rank's own replica in it) for doorbell kernels."""
        if getattr(self, "_ring_ranks", None):
            return self._ring_array, len(self._ring_ranks), self._ring_ranks.index(self.rank)
        return self._rep_array, self.world, self.rank

    def replace_replica(self, r: int):
        """Rank ``r`` was restarted (its region has been re-mapped): its control area is at a new address."""
        area = self.area_floats(self.world, self.lanes)
        self.replicas[r] = self._region.peer_address(r, self._region.sizes[r] - area)
        self._rep_array[r] = self.replicas[r]
        return self.replicas[r]

    def copy_block_to(self, r: int, owner: int):
        """Owner side: bring rank ``r``'s (new, zeroed) replica of this owner's block up to date from the local one."""
        area = self.area_floats(self.world, self.lanes)
        k = self._blk(owner)
        bf = 2 * self._block_words
        lo = self._region.sizes[self.rank] - area + k * bf
        self._region.put(r, self._region.sizes[r] - area + k * bf, self._region.local[lo:lo + bf])
        self._region.sync()

    # ---- replica mode: positions inside a control area, in 64-bit words
    def block_word(self, owner: int) -> int:
        return self._blk(owner) * self._block_words

    def lane_word(self, owner: int, worker: int) -> int:
        return self.block_word(owner) + _HDR_WORDS + _LANE_WORDS * worker

    def global_step_address(self) -> Optional[int]:
        """Replica mode: the shared global step (owner 0's word) in this rank's own device memory."""
        return None if self.replicas is None else self._base[self._blk(0)] + _GSTEP * 8

    def _broadcast(self, word: int, value: int):
        """Replica mode: one word, the same value into every replica (a kernel; returns when it has completed)."""
        with torch.cuda.device(self._device):
            rc = _k().dtf_words_broadcast(self._rep_array, self.world, int(word), int(value) % (1 << 64), None)
        if rc:
            raise RuntimeError("control word broadcast failed (%d)" % rc)

    # ---- addresses for the native side
    def block_address(self, owner: int) -> int:
        return self._base[self._blk(owner)]

    def lane_address(self, owner: int, worker: int) -> int:
        return self._base[self._blk(owner)] + (_HDR_WORDS + _LANE_WORDS * worker) * 8

    def _w(self, owner: int):
        return self.words[self._blk(owner)]

    # ---- words
    def _lane(self, owner: int, worker: int) -> int:
        return _HDR_WORDS + _LANE_WORDS * worker

    def version(self, owner: int) -> int:
        return int(self._w(owner)[_VERSION])

    def global_step(self) -> int:
        return int(self._w(0)[_GSTEP])

    def ack(self, owner: int, worker: int) -> int:
        return int(self._w(owner)[self._lane(owner, worker) + _ACK])

    def pull_version(self, owner: int, worker: int, slot: int) -> int:
        return int(self._w(owner)[self._lane(owner, worker) + _PULLV + slot])

    def ack_version(self, owner: int, worker: int, slot: int) -> int:
        return int(self._w(owner)[self._lane(owner, worker) + _ACKV + slot])

    def door(self, owner: int, worker: int) -> int:
        return int(self._w(owner)[self._lane(owner, worker) + _DOOR])

    def clock(self, worker: int) -> int:
        return int(self._w(0)[self._lane(0, worker) + _CLOCK])

    def set_clock(self, worker: int, value: int):
        if self.replicas is not None:
            self._broadcast(self.lane_word(0, worker) + _CLOCK, value)
            return
        self._w(0)[self._lane(0, worker) + _CLOCK] = np.uint64(value)

    def min_clock(self, skip=()) -> int:
        """The slowest worker's clock; ``skip``: worker indices to leave out (tasks known to be absent)."""
        w = self._w(0)
        clocks = [int(w[self._lane(0, i) + _CLOCK]) for i in range(self.lanes) if i not in skip]
        return min(clocks) if clocks else 1 << 60

    def set_lane(self, owner: int, worker: int, seq: int):
        """Owner side, pump paused: a restarted worker continues its lane at gradient number ``seq`` - whatever its
        predecessor rang but never completed is dropped."""
        if self.replicas is not None:
            self._broadcast(self.lane_word(owner, worker) + _DOOR, seq)
            self._broadcast(self.lane_word(owner, worker) + _ACK, seq)
            return
        w = self._w(owner)
        w[self._lane(owner, worker) + _DOOR] = np.uint64(seq)
        w[self._lane(owner, worker) + _ACK] = np.uint64(seq)

    def rebase_step(self, owner: int, global_step: int):
        """Owner (re-)initialised or restored with the cluster-wide step at ``global_step``: doors, acknowledgements
        and ``version`` stay monotonic (workers keep counting), only the step's base moves.  Pump paused."""
        w = self._w(owner)
        version = int(w[_VERSION])
        w[_GSBASE] = np.uint64((int(global_step) - version) % (1 << 64))       # read by this owner's pump only
        if self.replicas is not None:
            self._broadcast(self.block_word(owner) + _GSTEP, int(global_step))
            return
        w[_GSTEP] = np.uint64(int(global_step))

    def close(self):
        self.words = []
        self._base = []
        self._views = []
        if self._mine is None:
            return              # replica mode: the memory belongs to the peer region
        for i, s in enumerate(self._shms):
            if s is None:
                continue
            try:
                s.close()
            except Exception:
                pass
        try:
            self._mine.unlink()
        except Exception:
            pass
        self._shms = []


def ring(stream_ptr: int, on_gpu: bool, blocks: "ControlBlocks", worker: int, owners: int, pulled: List[int], slot: int,
         seq: int, clock: int):
    """After everything queued on the stream so far: publish gradient number ``seq`` of worker ``worker`` (slot ``slot``
    of the inboxes) to owners ``0 .. owners-1``.  Replica mode: a kernel on the stream stores the words into every
    replica; otherwise host code writes the shared-memory lanes (from a stream host function on a GPU)."""
    n = int(owners)
    pv = (C.c_ulonglong * n)(*pulled)
    if blocks.replicas is not None:
        lw = (C.c_longlong * n)(*[blocks.lane_word(j, worker) for j in range(n)])
        reps, nrep, _ = blocks.ring_array()
        rc = _k().dtf_doorbell_ring(C.c_void_p(stream_ptr or 0), 1, None, reps, nrep, lw, pv, n, int(slot), int(seq),
                                    int(clock))
    else:
        la = (C.c_void_p * n)(*[blocks.lane_address(j, worker) for j in range(n)])
        rc = _k().dtf_doorbell_ring(C.c_void_p(stream_ptr or 0), 1 if on_gpu else 0, la, None, 0, None, pv, n, int(slot),
                                    int(seq), int(clock))
    if rc:
        raise RuntimeError("doorbell failed (%d)" % rc)


class PushLink:
    """A GPU worker's native link to its owners (``dtf_link_*`` in ``csrc/kernels/owner_pump.hip``): the communication
    stream and events of the push, and one call per step that queues everything - wait for the backward pass, move the
    gradient into the owners' inboxes, ring the doorbells, release the gradient buffer."""

    def __init__(self, device, region, table, blocks: "ControlBlocks", worker: int, owners: int):
        self.device = torch.device(device)
        lib = _k()
        with torch.cuda.device(self.device):
            self._h = lib.dtf_link_create()
        if not self._h:
            raise RuntimeError("cannot create the push link")
        self._region, self._table, self._blocks = region, table, blocks
        n = self.n = int(owners)
        if blocks.replicas is not None:
            self._lanes = None
            self._lane_words = (C.c_longlong * n)(*[blocks.lane_word(j, worker) for j in range(n)])
        else:
            self._lanes = (C.c_void_p * n)(*[blocks.lane_address(j, worker) for j in range(n)])
            self._lane_words = None
        self._pulled = (C.c_ulonglong * n)()

    def stream_ptr(self) -> int:
        return _k().dtf_link_stream(self._h)

    def push(self, cur_stream_ptr: int, grad: torch.Tensor, remote_base: int, shard_floats: int, pulled: List[int], slot: int,
             seq: int, clock: int):
        for i, v in enumerate(pulled):
            self._pulled[i] = v
        b = self._blocks
        reps, nrep, _ = b.ring_array() if b.replicas is not None else (None, 0, 0)
        rc = _k().dtf_link_push(self._h, self._region._h, self._table.handle if self._table is not None else None,
                                C.c_void_p(cur_stream_ptr or 0), grad.data_ptr(), grad.numel() * 4, int(remote_base) * 4,
                                int(shard_floats) * 4, self._lanes, reps, nrep, self._lane_words, self._pulled, self.n,
                                int(slot), int(seq), int(clock))
        if rc:
            raise RuntimeError("push to the owners failed (%d)" % rc)

    def close(self):
        h, self._h = self._h, None
        if h:
            with torch.cuda.device(self.device):
                _k().dtf_link_destroy(h)


class OwnerPump:
    """The owner's native service thread: applies arriving gradient shards to ``var`` / slots on its own stream."""

    def __init__(self, *, device, owner: int, workers: int, depth: int, n: int, slot_stride: int, var, m, v, inbox,
                 step, beta_powers, optimizer, ctl_address: int, blocks: Optional["ControlBlocks"] = None):
        lib = _k()
        assert lib.dtf_pump_config_bytes() == C.sizeof(_PumpConfig), "owner_pump.hip and doorbell.py disagree"
        self.device = torch.device(device)
        on_gpu = self.device.type == "cuda"
        self._keep = (var, m, v, inbox, step, beta_powers)
        self._ticket = torch.zeros(1, dtype=torch.int32, device=self.device) if on_gpu else None
        cfg = _PumpConfig()
        cfg.on_gpu, cfg.owner, cfg.workers, cfg.depth = int(on_gpu), int(owner), int(workers), int(depth)
        # the rule and its constants: 0 Adam, 1 SGD, 2 RMSProp (b1 = rho, b2 = epsilon inside the root), 3 Adagrad
        code, lr, b1, b2, eps, momentum = optimizer.pump_parameters()
        is_adam = code == 0
        cfg.optimizer = code
        cfg.n, cfg.slot_stride = int(n), int(slot_stride)
        cfg.var, cfg.m, cfg.v, cfg.inbox = nv.ptr(var), nv.ptr(m), nv.ptr(v), nv.ptr(inbox)
        cfg.step, cfg.beta_powers = nv.ptr(step), nv.ptr(beta_powers if is_adam else None)
        cfg.ticket = nv.ptr(self._ticket)
        cfg.lr, cfg.b1, cfg.b2, cfg.eps, cfg.momentum = lr, b1, b2, eps, momentum
        cfg.ctl = ctl_address
        if on_gpu and blocks is not None and blocks.replicas is not None:
            # replica mode: acknowledgements stored into every rank's replica by a kernel behind the apply
            for r, a in enumerate(blocks.replicas):
                cfg.rep_base[r] = a
            cfg.rep_world = blocks.world
            cfg.block_word = blocks.block_word(owner)
        if on_gpu:
            with torch.cuda.device(self.device):
                self._h = lib.dtf_pump_start(C.byref(cfg))
        else:
            self._h = lib.dtf_pump_start(C.byref(cfg))
        if not self._h:
            raise RuntimeError("cannot start the owner pump")

    def pause(self):
        rc = _k().dtf_pump_pause(self._h)
        if rc:
            raise RuntimeError("owner pump failed (%d)" % rc)

    def resume(self, resync: bool = False):
        _k().dtf_pump_resume(self._h, 1 if resync else 0)

    def set_replica(self, r: int, address: int):
        """Paused pump, replica mode: rank ``r``'s control area moved (a restarted worker)."""
        rc = _k().dtf_pump_set_replica(self._h, int(r), C.c_void_p(address))
        if rc:
            raise RuntimeError("cannot update the pump's replica table (%d)" % rc)

    def error(self) -> int:
        return int(_k().dtf_pump_error(self._h)) if self._h else 0

    def stream(self) -> Optional[int]:
        return _k().dtf_pump_stream(self._h)

    def histogram(self):
        out = (C.c_longlong * HIST)()
        _k().dtf_pump_histogram(self._h, out)
        return {i: int(out[i]) for i in range(HIST) if out[i]}

    def stop(self) -> int:
        h, self._h = self._h, None
        return int(_k().dtf_pump_stop(h)) if h else 0


def spin_until(pred, timeout_s: float, what: str, check=None):
    """Poll ``pred`` (a cheap read of shared words) until true; the first polls spin, later ones sleep briefly."""
    if pred():
        return 0.0
    t0 = time.perf_counter()
    n = 0
    while not pred():
        n += 1
        if n > 150:         # ~0.2 ms of polling first: a sleep, however short it is asked to be, costs ~0.1 ms
            time.sleep(0.00002 if n < 2000 else 0.0005)
        if n % 256 == 0:
            if check is not None:
                check()
            if time.perf_counter() - t0 > timeout_s:
                raise TimeoutError("timed out after %.0f s waiting for %s" % (timeout_s, what))
    return time.perf_counter() - t0

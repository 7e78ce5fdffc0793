"""Asynchronous parameter-server training with bounded staleness.

This is the reference's actual mode (/root/reference/example.py:136-139,168-170): every
worker trains on its own data, reads the variables from the parameter servers, and sends
gradients that the servers apply one by one, lock-free, with the plain (non-synchronising)
optimizer; ``global_step`` is shared, so ``StopAtStepHook(last_step=N)`` is a cluster-wide
budget.

Re-designed for one MI355X node:

* an *owner* holds ``[begin, end)`` ranges of the flat variable buffer together with the
  optimizer slots for them, resident in its HBM, and applies the fused apply-gradients kernel
  to its shard on arrival, on a stream of its own - updates from different workers interleave
  freely across owners (Hogwild-style, ``use_locking=False`` semantics), each shard update
  itself is atomic;
* **colocated owners** (no ps tasks, or ps tasks without a GPU of their own): a service thread
  inside every worker process owns 1/N of the flat buffer, so 8 workers on 8 GPUs shard the
  variables across all eight HBMs and nobody else holds optimizer state.  The data plane is one-sided and stays
  on the devices: a worker *puts* its gradient shard into the owner's inbox slot through peer-mapped memory over
  xGMI (``peer_region.PeerRegion``: hipIpc on GPUs, POSIX shared memory on CPUs) and rings a doorbell word from a
  kernel queued behind the copies; the owner's native service thread (``csrc/kernels/owner_pump.hip``)
  launches the fused apply kernel on its own stream and an acknowledgement kernel behind it (``parallel/doorbell.py``:
  the words are replicas in every rank's device memory, polled by the local CPU; host-written shared memory on
  CPUs); a worker *gets* the owners' variable shards straight into its flat buffer at the
  start of a step.  A step makes no control-plane round trip and never waits for a stream on the host; with
  two inbox slots per worker the next forward pass starts while the owners are still applying (that lag is
  part of the staleness bound).  Version, staleness, ``global_step`` and the workers' clocks ride in the words;
* **ps tasks as owners** (separate processes, as in the reference; whole variables placed
  round-robin or byte-balanced).  On one node, with ps tasks and workers on the same kind of device, they use the
  same one-sided plane: every ps task allocates a region (its packed variables, two slot planes, inbox slots),
  workers put the flat ranges of an owner's variables into their inbox slot and get them back from its variable
  plane, doorbells and the native pump do the rest - no process group, no socket on the step path; for the
  reference network at a batch of at most 64 a worker's whole step is one kernel that reads the variables from and
  writes the gradients into the ps tasks' memory and rings the doorbells itself (``train_step_direct``);
* ps tasks on other nodes, or on another kind of device than their workers (``DTF_PS_DATA_PLANE=p2p`` forces it):
  one packed gradient message to and one packed parameter message from every owner per step over
  ``torch.distributed`` send/recv - RCCL between GPUs, gloo between CPU processes - instead of TF's one RecvTensor
  RPC per variable per direction, each process with exactly one thread on the data plane.  Control traffic is
  folded: per step a worker makes one round trip per control server to queue its requests, one to collect the
  replies (which carry the owner's version, the observed staleness and the new ``global_step``), and - with a
  staleness bound - one to publish its clock and wait.  The control plane otherwise serves rendezvous, barriers,
  heartbeats, checkpoints and shutdown;
* *bounded staleness* (stale-synchronous parallel): a worker that has completed ``c`` pushes may
  start its next step only while ``c - min(all clocks) <= max_staleness``.  ``max_staleness=None``
  is the reference's unbounded behaviour.  The observed staleness of every push (owner version at
  apply minus owner version at the worker's last pull) is histogrammed;
* checkpoints are written by the owners: every owner writes its own data shard (variables *and*
  optimizer slots), the chief merges the index; on restore the owners reload their shards from the
  files themselves, slots and ``beta*_power`` included.
"""
from __future__ import annotations

import json
import threading
import time
import uuid
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from ..control import ControlClient, ControlTimeout
from ..variables import VariableStore



def _ranges_numel(ranges: Sequence[Tuple[int, int]]) -> int:
    return sum(e - b for b, e in ranges)


def merge_ranges(ranges: Sequence[Tuple[int, int]]) -> List[Tuple[int, int]]:
    out: List[Tuple[int, int]] = []
    for b, e in sorted(ranges):
        if out and out[-1][1] == b:
            out[-1] = (out[-1][0], e)
        else:
            out.append((b, e))
    return out


def owner_ranges(store: VariableStore, placement: Dict[str, int], num_owners: int) -> List[List[Tuple[int, int]]]:
    """Flat-buffer ranges owned by each owner.  Placement value -1 (``flat``) splits the whole buffer evenly."""
    if num_owners <= 0:
        return []
    if placement and all(v == -1 for v in placement.values()):
        return [[store.shard_range(j, num_owners)] for j in range(num_owners)]
    per: List[List[Tuple[int, int]]] = [[] for _ in range(num_owners)]
    tr = store.trainable
    for i, v in enumerate(tr):
        end = tr[i + 1].offset if i + 1 < len(tr) else store.flat_len
        per[placement.get(v.name, i % num_owners) % num_owners].append((v.offset, end))
    return [merge_ranges(r) for r in per]


def owned_pieces(variables, ranges):
    """Intersections of the variables with an owner's flat ranges.

    ``variables``: ``(name, offset, numel, shape)`` in layout order.  Returns dicts with the variable (``name``,
    ``shape``, ``offset``, ``numel``), the piece inside it (``lo``: first element of the variable covered, ``n``
    elements) and ``shard_off``: where the piece starts in the owner's packed shard."""
    out = []
    off = 0
    for b, e in ranges:
        for name, voff, numel, shape in variables:
            lo, hi = max(b, voff), min(e, voff + numel)
            if lo < hi:
                out.append({"name": name, "shape": tuple(shape), "offset": voff, "numel": numel, "lo": lo - voff,
                            "n": hi - lo, "shard_off": off + lo - b})
        off += e - b
    return out


def owner_of_element(all_ranges, index: int) -> int:
    for j, rs in enumerate(all_ranges):
        for b, e in rs:
            if b <= index < e:
                return j
    raise ValueError("flat index %d belongs to no owner" % index)


def _pack(flat: torch.Tensor, ranges, out: torch.Tensor) -> torch.Tensor:
    if len(ranges) == 1:
        b, e = ranges[0]
        return flat[b:e]
    pos = 0
    for b, e in ranges:
        out[pos:pos + e - b].copy_(flat[b:e])
        pos += e - b
    return out


def _unpack(buf: torch.Tensor, flat: torch.Tensor, ranges):
    pos = 0
    for b, e in ranges:
        if buf.data_ptr() != flat[b:e].data_ptr():
            flat[b:e].copy_(buf[pos:pos + e - b])
        pos += e - b


def region_sizes(stride: int, depth: int, num_ps: int, num_workers: int, area: int = 0) -> List[int]:
    """Floats per rank of the peer regions of a cluster whose ps tasks are the owners (ranks: ps tasks, then workers):
    variables + two slot planes + ``depth`` inbox slots per worker on a ps task, a token allocation on a worker; every
    rank ends with ``area`` floats for its replica of the control blocks (``ControlBlocks.area_floats``; 0: the
    blocks live in shared memory)."""
    return [3 * stride + num_workers * depth * stride + area] * num_ps + [64 + area] * num_workers


_LOOPBACK = {"localhost", "127.0.0.1", "0.0.0.0", "::1"}


def split_host(address: str) -> str:
    return address.rsplit(":", 1)[0]


def _numel(shape) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n


# --------------------------------------------------------------------------- data plane: torch.distributed send/recv
_TAG_PUSH, _TAG_PULL, _TAG_INIT = 11, 12, 13


def _staged(t: torch.Tensor) -> bool:
    """GPU tensor on the gloo backend (ranks time-sharing one GPU): moved through host memory explicitly, see
    ``Communicator._staged``."""
    return t.is_cuda and dist.get_backend() == "gloo"


class _P2P:
    """send/recv between the global ranks of a cluster whose owners are ps tasks.

    Every process has ONE thread on the data plane (a worker's training thread, a ps task's service loop), and each
    direction has its own process group: on ``push`` only workers send and only owners receive, on ``pull`` the
    reverse, so every (source, destination) pair of a communicator carries one strictly ordered stream.  RCCL matches
    the operations of a pair purely by issue order on one communicator, and PyTorch routes an operation issued inside
    a batch through the group-wide communicator but a lone one through a two-rank communicator of its own - so over
    RCCL *every* operation here is issued as a batch (of one, if need be): both ends then always meet on the same
    communicator.
    """

    def __init__(self, rank: int):
        self.rank = rank
        self.push_group = None      # worker -> owner (gradients, initial values)
        self.pull_group = None      # owner -> worker (parameters)

    def make_direction_groups(self):
        """Collective over the whole process group: every member calls this once, right after joining."""
        if self.push_group is None and dist.is_initialized() and dist.get_world_size() > 1:
            self.push_group = dist.new_group()
            self.pull_group = dist.new_group()

    def _group(self, tag: int):
        return self.pull_group if tag == _TAG_PULL else self.push_group

    def _issue(self, ops):
        """Start the operations (all of one group); returns the work handles."""
        if not ops:
            return []
        if dist.get_backend() == "nccl":
            return dist.batch_isend_irecv(ops)
        return [op.op(op.tensor, op.peer, group=op.group, tag=op.tag) for op in ops]

    def send(self, t: torch.Tensor, dst: int, tag: int):
        self.exchange([(t, dst, tag)], [])

    def recv(self, t: torch.Tensor, src: int, tag: int):
        self.exchange([], [(t, src, tag)])

    def exchange(self, sends, recvs):
        """Post every send ``(tensor, dst, tag)`` and receive ``(tensor, src, tag)`` at once, then wait for all.

        A worker must never block on one owner before it has posted its traffic to the others: owners serve
        workers in arrival order, so blocking sends issued one owner at a time can form a cycle.  Over RCCL the
        sends form one group and the receives another (a batch may not mix communicators); the operations of a
        group progress independently of each other.
        """
        by_group: Dict[int, list] = {}
        staged = []
        keep = []
        for t, dst, tag in sends:
            h = t.cpu() if _staged(t) else t
            keep.append(h)
            by_group.setdefault(id(self._group(tag)), []).append(dist.P2POp(dist.isend, h, dst, group=self._group(tag), tag=tag))
        for t, src, tag in recvs:
            h = t
            if _staged(t):
                h = torch.empty(t.shape, dtype=t.dtype)
                staged.append((t, h))
            by_group.setdefault(id(self._group(tag)), []).append(dist.P2POp(dist.irecv, h, src, group=self._group(tag), tag=tag))
        try:
            works = []
            for ops in by_group.values():
                works += self._issue(ops)
            for w in works:
                w.wait()
        except RuntimeError as e:
            # the transport reports a dead peer as a plain RuntimeError ("Connection closed by peer", an RCCL abort):
            # for the training loop that is the loss of a task, whichever detector notices it first
            from ..failure import ClusterLostError
            raise ClusterLostError("data plane failed (%s)" % str(e).splitlines()[0][:200]) from e
        for t, h in staged:
            t.copy_(h)


# --------------------------------------------------------------------------- owner / parameter-server side
class ParameterServerService:
    """Owner of a set of flat ranges: serves init / pushpull / pull / save / restore / stats / shutdown.

    Two data planes: ``region`` (a :class:`~.peer_region.PeerRegion` shared with the workers: colocated owners, and
    ps tasks that share a node and a device type with their workers) or ``p2p`` (``torch.distributed`` send/recv: ps
    tasks on other nodes or other kinds of device).  Layout of a region, in floats, with ``S`` the layout's
    ``stride`` (the shard length ``flat_len / owners`` of colocated owners; the largest owner's element count, rounded
    up, for ps tasks), ``W`` workers and ``D`` inbox slots per worker:

    * ``[0, S)``                       this owner's variables (workers read them with one-sided gets);
    * ``[S, 2S)`` and ``[2S, 3S)``     its optimizer slots (Adam m and v / momentum): nobody else holds any;
    * ``[3S + (w*D + d)*S, ...)``      inbox slot ``d`` of worker ``w`` (its gradient shard number ``t`` with ``t % D == d``).

    With a region the hot path (apply on arrival, acknowledge) is the native pump of ``doorbell.OwnerPump``; this
    class then only serves the rare requests that arrive over the control plane (init, save, restore, stats, shutdown).
    """

    def __init__(self, server=None, owner_index: Optional[int] = None, control: Optional[ControlClient] = None,
                 control0: Optional[ControlClient] = None, rank: Optional[int] = None, device=None,
                 p2p: Optional[_P2P] = None, region=None, num_workers: Optional[int] = None, worker_rank_of=None,
                 num_owners: Optional[int] = None, ns: str = "ps"):
        self.server = server
        self.ns = ns                        # key prefix: "ps" for ps tasks, "co" for owners colocated with workers
        self.owner_index = owner_index if owner_index is not None else server.task_index
        if control is None:
            host, port = server.control_address(self.owner_index)
            control = ControlClient(host if host not in ("0.0.0.0",) else "127.0.0.1", port)
        self.cp = control                   # this owner's request queue and replies live here
        if control0 is None:
            host, port = server.control_address(0)
            control0 = ControlClient(host if host not in ("0.0.0.0",) else "127.0.0.1", port)
        self.cp0 = control0                 # cluster-wide counters and barriers live on the first control server
        self.device = device if device is not None else server.device()
        self.rank = rank if rank is not None else server.rank
        self.num_workers = num_workers if num_workers is not None else server.num_workers
        self.num_owners = num_owners if num_owners is not None else max(server.num_ps, 1)
        self.worker_rank_of = worker_rank_of or (lambda i: server.worker_rank(i))
        self.p2p = p2p
        self.region = region
        self.queue = "%s/req/%d" % (self.ns, self.owner_index)
        self.ranges: List[Tuple[int, int]] = []
        self.var = self.grad = None
        self.slots: Dict[str, torch.Tensor] = {}
        self.version = 0            # updates applied by this owner
        self.step_t = None
        self.opt = None
        self.staleness_hist: Dict[int, int] = {}
        self._owns_group = False
        self.stream = None
        self.error: Optional[BaseException] = None
        self.blocks = None                  # doorbell.ControlBlocks (region mode)
        self.depth = 1
        self.pump = None
        self._owns_region = False           # a ps task maps its region itself, after it has read the layout

    # a tiny VariableStore look-alike so Optimizer.apply works on the owned shard
    class _Shard:
        def __init__(self, svc):
            self.flat = svc.var
            self.flat_grad = svc.grad
            self.flat_len = svc.var.numel()
            self.global_step = svc.step_t
            self.beta_powers = svc.beta_powers
            self._svc = svc

        def slot(self, name, initial=None):
            s = self._svc.slots
            if name not in s:
                svc = self._svc
                first = float(svc.opt.slot_initial(name))
                if svc.region is not None:
                    # slots live in the region, behind the variables, in the optimizer's order (Adam: m, then v)
                    k = svc.slot_plane(name)
                    s[name] = svc.region.local[k * svc.slot_len:k * svc.slot_len + max(svc.n, 1)]
                    s[name].fill_(first)
                else:
                    s[name] = torch.full_like(svc.var, first)
            return s[name]

    def slot_plane(self, name: str) -> int:
        """Which third of a region holds slot ``name`` (0: the variables)."""
        return 1 + self.opt.slot_order.index(name)

    def _join_data_plane(self):
        if dist.is_initialized():
            return
        from .comm import init_process_group, pick_backend
        host, port = self.server.rendezvous_address()
        backend = pick_backend(self.device)
        init_process_group(backend, rank=self.rank, world_size=self.server.world_size, master_addr=host,
                           master_port=port, device=self.device if backend == "nccl" else None)
        self._owns_group = True

    def _join_region_plane(self, layout: dict):
        """ps task: allocate this owner's region and control block and map everybody else's (collective with the other
        ps tasks and all workers; the ps tasks are the first ranks)."""
        from .doorbell import ControlBlocks
        from .peer_region import PeerRegion
        srv = self.server
        world = self.num_owners + self.num_workers
        self.depth = int(layout["depth"])
        replica = layout.get("words") == "replica"
        area = ControlBlocks.area_floats(world, self.num_workers) if replica else 0
        sizes = region_sizes(int(layout["stride"]), self.depth, self.num_owners, self.num_workers, area)
        self.region = PeerRegion(self.rank, world, sizes[self.rank], self.device, self.cp0, name=self.ns + "/region",
                                 sizes=sizes)
        self.blocks = ControlBlocks(self.rank, world, self.cp0, name=self.ns + "/ctl", lanes=self.num_workers,
                                    owner_rank=srv.ps_rank, region=self.region if replica else None)
        self._owns_region = True

    def _configure(self, layout: dict):
        from .. import optimizers
        self.layout = layout
        self.all_ranges = [[tuple(r) for r in rs] for rs in layout["ranges"]]
        self.ranges = self.all_ranges[self.owner_index]
        self.variables = [(n, int(o), int(k), tuple(s)) for n, o, k, s in layout["variables"]]
        self.flat_len = int(layout["flat_len"])
        n = self.n = _ranges_numel(self.ranges)
        if self.region is not None:
            self.slot_len = int(layout.get("stride", self.flat_len // self.num_owners))
            self.var = self.region.local[0:max(n, 1)]
            self.grad = None
        else:
            self.var = torch.zeros(max(n, 1), dtype=torch.float32, device=self.device)
            self.grad = torch.zeros_like(self.var)
        self.step_t = torch.zeros(1, dtype=torch.int64, device=self.device)
        o = layout["optimizer"]
        self.opt = optimizers.from_description(o)
        self.beta_powers = optimizers.initial_beta_powers(self.opt, self.device)
        self.shard = ParameterServerService._Shard(self)
        self.opt.create_slots(self.shard)
        self.n = n
        self.pieces = owned_pieces(self.variables, self.ranges)

    # -- the native hot path of a region owner
    def _start_pump(self):
        from .doorbell import OwnerPump
        S = self.slot_len
        self.pump = OwnerPump(device=self.device, owner=self.owner_index, workers=self.num_workers, depth=self.depth,
                              n=self.n, slot_stride=S, var=self.var, m=self._slot_at(0), v=self._slot_at(1),
                              inbox=self.region.local[3 * S:], step=self.step_t,
                              beta_powers=self.beta_powers, optimizer=self.opt,
                              ctl_address=self.blocks.block_address(self.owner_index), blocks=self.blocks)

    def _slot_at(self, i: int):
        order = self.opt.slot_order
        return self.slots.get(order[i]) if i < len(order) else None

    def _quiesce(self):
        """Stop applying (what is launched completes): the shard is stable until ``_release``."""
        if self.pump is not None:
            self.pump.pause()

    def _release(self, resync: bool = False):
        if self.pump is not None:
            self.pump.resume(resync)

    # -- transport (ps tasks)
    def _incoming(self, worker: int, wr, tag: int) -> torch.Tensor:
        """The tensor worker ``worker`` sent for this request (gradient, or initial values)."""
        buf = self.grad if tag == _TAG_PUSH else self.var
        self.p2p.recv(buf[:self.n], wr, tag)
        return buf[:self.n]

    def _send_params(self, worker: int, wr):
        self.p2p.send(self.var[:self.n], wr, _TAG_PULL)

    def _reply(self, worker: int, staleness: int, count_step: bool, seq: int = 0):
        # `seq` echoes the request's sequence number: a worker that gave up on a slow reply (ControlTimeout, then
        # in-process recovery) recognises the late answer to its abandoned request and drops it
        body = '{"version": %d, "global_step": %s, "staleness": %d, "seq": %d}' % (
            self.version, "%GS%" if count_step else "-1", staleness, seq)
        self.cp.add_set("global_step" if count_step else None, 1 if count_step else 0,
                        "%s/reply/%d/%d" % (self.ns, self.owner_index, worker), body.encode())

    def serve(self):
        """Request loop; returns after a ``shutdown`` request.  An exception ends the loop and is kept in ``error``
        (and re-raised for a ps task, whose process then exits non-zero)."""
        try:
            self._serve()
        except BaseException as e:      # noqa: BLE001 - the worker thread of a colocated owner must learn about it
            self.error = e
            try:
                self.cp0.set("%s/failed/%d" % (self.ns, self.owner_index), repr(e).encode())
            except Exception:
                pass
            raise

    def _serve(self):
        # the chief posts the layout; a cluster that trains without its ps tasks (synchronous replicas,
        # colocated owners) only ever sends "shutdown"
        layout = None
        if self.region is None:
            # what the chief needs to choose the data plane: where this ps task runs and on which kind of device
            import socket
            self.cp0.set("%s/where/%d" % (self.ns, self.owner_index),
                         ("%s@%s" % (self.device.type, socket.gethostname())).encode())
        while layout is None:
            try:
                layout = self.cp.get_json("%s/layout/%d" % (self.ns, self.owner_index), timeout_s=0.5)
            except ControlTimeout:
                if self.cp.qlen(self.queue) > 0:
                    req = self.cp.pop_json(self.queue, timeout_s=1.0)
                    if req.get("op") == "shutdown":
                        self.cp0.add(self.ns + "/stopped", 1)
                        return
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
            # ps tasks apply on this stream; a region owner's pump has a stream of its own and this one only carries
            # the rare requests (initialisation, checkpoints)
            self.stream = torch.cuda.Stream(device=self.device)
        if self.region is None and self.p2p is None:
            # only clusters that use their ps tasks put them on the data plane
            if layout.get("plane") == "region":
                self._join_region_plane(layout)
            else:
                self._join_data_plane()
                self.p2p = _P2P(self.rank)
                self.p2p.make_direction_groups()
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _NullContext()
        with ctx:
            self._configure(layout)
            self.cp0.add(self.ns + "/ready", 1)
            try:
                self._loop()
            finally:
                if self.pump is not None:
                    pump, self.pump = self.pump, None
                    pump.stop()
                if self._owns_region:
                    self.var = self.shard = None
                    self.slots = {}
                    self.blocks.close()         # (replica mode: its views point into the region)
                    self.region.close()
        self.cp0.add(self.ns + "/stopped", 1)
        if self._owns_group and dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:
                pass

    def _loop(self):
        while True:
            try:
                req = self.cp.pop_json(self.queue, timeout_s=5.0)
            except ControlTimeout:
                if self.pump is not None and self.pump.error():
                    raise RuntimeError("the owner pump failed with status %d" % self.pump.error())
                continue
            op = req["op"]
            if op == "shutdown":
                return
            w = req.get("worker")
            wr = self.worker_rank_of(w) if w is not None else None
            seq = int(req.get("seq", 0))
            if op == "init":
                self._quiesce()
                if self.n and self.region is None:
                    src = self._incoming(w, wr, _TAG_INIT)
                    if src.data_ptr() != self.var.data_ptr():
                        self.var[:self.n].copy_(src)
                # (a region owner's variables were put into place by the chief before it queued this request)
                for name, s in self.slots.items():
                    s.fill_(float(self.opt.slot_initial(name)))
                self.version = 0
                self.step_t.fill_(int(req.get("global_step", 0)))
                from .. import optimizers
                self.beta_powers.copy_(optimizers.initial_beta_powers(self.opt, self.device))
                if self.stream is not None:
                    self.stream.synchronize()
                if self.region is not None:
                    self.blocks.rebase_step(self.owner_index, int(req.get("global_step", 0)))
                    if self.pump is None:
                        self._start_pump()
                    else:
                        self._release()
                self.cp0.add(self.ns + "/init_done", 1)
            elif op == "pushpull":
                if self.n:
                    g = self._incoming(w, wr, _TAG_PUSH)
                    self.opt.apply(self.shard, 0, self.n, grad_scale=1.0, grad=g, step=self.step_t)
                    self.opt.finish_step(self.shard)
                st = self.version - int(req.get("version", self.version))
                self.staleness_hist[st] = self.staleness_hist.get(st, 0) + 1
                self.version += 1
                # the reply first: the worker reads it after its receive has completed anyway, and a send
                # that waits for the peer must not delay it
                self._reply(w, st, self.owner_index == 0, seq)
                if self.n:
                    self._send_params(w, wr)
            elif op == "rejoin_map":
                # a restarted worker has allocated a new region: with replicated control words this owner maps it, points
                # its acknowledgements at the new replica and copies its block there (the dead process's mapping goes)
                if self.blocks is not None and self.blocks.replicas is not None:
                    self._quiesce()
                    try:
                        self.region.remap_peer(wr, self.cp0)
                        addr = self.blocks.replace_replica(wr)
                        if self.pump is not None:
                            self.pump.set_replica(wr, addr)
                        self.blocks.copy_block_to(wr, self.owner_index)
                    finally:
                        self._release()
                self.cp0.add("%s/rejoin_map_done/%d" % (self.ns, w), 1)
            elif op == "clock":
                self.blocks.set_clock(w, int(req["value"]))
                self.cp0.add("%s/clock_done/%d" % (self.ns, w), 1)
            elif op == "rejoin":
                # a restarted worker takes over lane `w`: nothing is in flight from its predecessor once the pump has
                # drained, so door, acknowledgement and the pump's own count can all move to the agreed number
                self._quiesce()
                self.blocks.set_lane(self.owner_index, w, int(req["seq"]))
                self._release(resync=True)
                self.cp0.add("%s/rejoin_done/%d" % (self.ns, w), 1)
            elif op == "pull":
                self._reply(w, 0, False, seq)
                if self.n:
                    self._send_params(w, wr)
            elif op == "save":
                self._save(req)
            elif op == "restore":
                self._restore(req)
            elif op == "stats":
                if self.region is not None:
                    version = self.blocks.version(self.owner_index)
                    hist = self.pump.histogram() if self.pump is not None else {}
                else:
                    version, hist = self.version, self.staleness_hist
                self.cp.set_json("%s/stats/%d" % (self.ns, self.owner_index),
                                 {"version": version, "staleness": {str(k): v for k, v in hist.items()}})
            if self.pump is not None and self.pump.error():
                raise RuntimeError("the owner pump failed with status %d" % self.pump.error())

    # -- checkpoints
    def _save(self, req):
        """This owner's part of a checkpoint: data shard ``owner_index`` with the variables this owner *writes* - those
        whose first element it owns - and their optimizer slots.  A variable that straddles owners (flat sharding)
        is assembled by its writer with one-sided gets from the regions of the owners that hold the rest; every owner
        has stopped applying while that goes on."""
        from ..saver import BundleWriter
        token = req["token"]
        gather = bool(req.get("gather"))
        if gather and self.region is None:
            raise RuntimeError("variables split across ps tasks cannot be checkpointed by their owners")
        kinds = [("", None)] + [(suffix, slot) for slot, suffix in sorted(req["slots"].items()) if slot in self.slots]
        w = BundleWriter(req["prefix"], req["num_shards"], only_shard=self.owner_index)
        self._quiesce()
        try:
            if gather:
                self.cp0.barrier("%s/ckpt/%s/in" % (self.ns, token), self.num_owners, timeout_s=600)
            for suffix, slot in kinds:
                src = self.var if slot is None else self.slots[slot]
                plane = 0 if slot is None else self.slot_plane(slot)      # which third of a region holds it
                for p in self.pieces:
                    if p["lo"] != 0:
                        continue                                   # written by the owner of the variable's first element
                    if p["n"] == p["numel"]:
                        t = src[p["shard_off"]:p["shard_off"] + p["numel"]]
                    else:
                        t = torch.empty(p["numel"], dtype=torch.float32, device=self.device)
                        t[:p["n"]].copy_(src[p["shard_off"]:p["shard_off"] + p["n"]])
                        S = self.slot_len
                        pos = p["offset"] + p["n"]                 # flat index of the first element somebody else holds
                        while pos < p["offset"] + p["numel"]:
                            j = pos // S
                            n = min(p["offset"] + p["numel"], (j + 1) * S) - pos
                            self.region.get(j, plane * S + (pos - j * S), t[pos - p["offset"]:pos - p["offset"] + n])
                            pos += n
                        self.region.sync()
                    name = p["name"] if not suffix else "%s/%s" % (p["name"], suffix)
                    w.add(name, t.reshape(p["shape"]), self.owner_index)
            if self.owner_index == 0:
                for name, e in (req.get("extras") or {}).items():
                    w.add(name, torch.tensor(e["values"], dtype=torch.float32).reshape(e["shape"]), 0)
                w.add("global_step", torch.tensor(int(req["global_step"]), dtype=torch.int64), 0)
                if req["optimizer"] == "Adam":
                    w.add("beta1_power", self.beta_powers[0], 0)
                    w.add("beta2_power", self.beta_powers[1], 0)
            w.finish()
            if gather:
                self.cp0.barrier("%s/ckpt/%s/out" % (self.ns, token), self.num_owners, timeout_s=600)
        except Exception:
            w.abort()
            raise
        finally:
            self._release()
        self.cp0.add("%s/save_done/%s" % (self.ns, token), 1)

    def _restore(self, req):
        """Reload this owner's pieces - variables, optimizer slots, ``beta*_power``, the step - from a checkpoint."""
        from ..saver import BundleReader
        self._quiesce()
        try:
            with BundleReader(req["prefix"]) as r:
                for p in self.pieces:
                    sl = slice(p["shard_off"], p["shard_off"] + p["n"])
                    self.var[sl].copy_(r.get_tensor(p["name"]).reshape(-1)[p["lo"]:p["lo"] + p["n"]])
                    for slot, suffix in req["slots"].items():
                        key = "%s/%s" % (p["name"], suffix)
                        if slot in self.opt.slot_order and r.has_tensor(key):
                            self.shard.slot(slot)[sl].copy_(r.get_tensor(key).reshape(-1)[p["lo"]:p["lo"] + p["n"]])
                gs = None
                if r.has_tensor("global_step"):
                    gs = int(r.get_tensor("global_step").item())
                    self.step_t.fill_(gs)
                if r.has_tensor("beta1_power"):
                    self.beta_powers[0] = r.get_tensor("beta1_power")
                    self.beta_powers[1] = r.get_tensor("beta2_power")
            self.version = 0
            if self.stream is not None:
                self.stream.synchronize()
            if self.region is not None:
                if gs is not None:
                    self.blocks.rebase_step(self.owner_index, gs)
                if self.pump is None:
                    self._start_pump()
        finally:
            self._release()
        self.cp0.add("%s/restore_done/%s" % (self.ns, req["token"]), 1)


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# --------------------------------------------------------------------------- worker side
class AsyncPSStrategy:
    """Worker-side strategy for ``Sequential.compile(strategy=...)``.

    ``server``: the task's :class:`~distributed_tensorflow_amd.cluster.Server`.  With ps tasks that can each have a
    device of their own they are the owners; otherwise (no ps tasks, or more tasks than GPUs) - or with
    ``colocated=True`` - every worker also runs an owner service thread, the variables are flat-sharded over all
    workers and the ps tasks only serve the control plane.
    """

    def __init__(self, server, max_staleness: Optional[int] = None, placement="round_robin",
                 colocated: Optional[bool] = None, timeout_s: float = 300.0):
        self.server = server
        self.max_staleness = max_staleness
        self.worker_index = server.task_index
        self.num_workers = server.num_workers
        if colocated is None:
            import os
            forced = os.environ.get("DTF_COLOCATED_OWNERS")      # "1" / "0": override the device-driven choice
            colocated = (forced == "1") if forced in ("0", "1") and server.num_ps > 0 else \
                (server.num_ps == 0 or not server.ps_tasks_have_devices())
        self.colocated = bool(colocated)
        self.num_owners = self.num_workers if self.colocated else server.num_ps
        if placement == "flat" and not self.colocated:
            raise ValueError("flat sharding needs colocated owners (ps tasks own whole variables)")
        self.placement_name = "flat" if self.colocated else placement
        self.ns = "co" if self.colocated else "ps"      # ps tasks that only serve the control plane keep "ps/..."
        self.timeout_s = timeout_s
        self.world_size = 1           # gradients are NOT averaged across workers in async mode
        self.rank = 0
        self.clock = 0
        self.versions: List[int] = []
        self.staleness_hist: Dict[int, int] = {}
        self.last_global_step = 0
        self.comm_seconds = 0.0       # host time spent in the gradient push / parameter pull of every step
        self.wait_seconds = 0.0       # host time blocked by the staleness bound
        self._service = None
        self._service_thread = None
        self._seq = 0                 # sequence number of this worker's latest posting (echoed by the owners)
        self._cps: List[ControlClient] = []
        self._staging: Dict = {}
        self.region = None
        self.p2p = None
        self.plane = None             # "region" (peer regions + doorbell words) or "p2p" (send/recv), set by prepare
        self.elastic = False          # DTF_ELASTIC_WORKERS=1: workers may die and be re-admitted (ps tasks own all state)
        self._rejoined = False        # this process replaced a worker that died
        self._range_table = None      # peer_region.RangeTable of the owners' flat ranges (ps tasks on the region plane)
        self._link = None             # doorbell.PushLink: the native push of a GPU worker (stream, events, doorbells)
        self._direct_plans: Dict = {}  # batch size -> plan of the one-kernel step (train_step_direct), or None
        self._flat_stale = False      # direct steps read the owners' memory: the local copy of the variables is behind
        self.blocks = None            # doorbell.ControlBlocks: the owners' words (region plane)
        self.depth = 1                # inbox slots per (owner, worker): gradients a worker may have outstanding
        self._pulled: List[int] = []  # owner versions seen at the last pull
        self._sent_pulled: List[List[int]] = []   # [slot][owner]: the same, as sent along with the gradient in that slot
        self._acked: List[int] = []   # acknowledgements of each owner already accounted for in the histogram
        self._comm_stream = None
        self._ev_grad = self._ev_copied = None

    #: nobody but the owners holds optimizer state: ``compile`` must not allocate slot buffers on a worker
    owns_optimizer_state = True

    def var_alignment(self) -> int:
        return 64 * max(1, self.num_owners)

    # -- control plane helpers
    def _distinct_cps(self) -> List[Tuple[ControlClient, List[int]]]:
        """(client, owners whose queue lives on that control server), one entry per distinct server."""
        out: List[Tuple[ControlClient, List[int]]] = []
        for j, c in enumerate(self._cps):
            for cc, js in out:
                if cc is c:
                    js.append(j)
                    break
            else:
                out.append((c, [j]))
        return out

    def control_calls(self) -> int:
        """Control-plane round trips this worker has made so far (all of its clients)."""
        return sum(c.calls for c, _ in self._distinct_cps())

    def _post(self, op: str, extra=None, owners=None):
        """Queue one request per owner: one round trip per control server.  Every posting gets a new sequence number
        that the owners echo in their replies."""
        self._seq += 1
        for c, js in self._distinct_cps():
            items = []
            for j in js:
                if owners is not None and j not in owners:
                    continue
                req = {"op": op, "worker": self.worker_index, "seq": self._seq}
                if extra is not None:
                    req.update(extra(j) if callable(extra) else extra)
                items.append(("%s/req/%d" % (self.ns, j), req))
            if items:
                c.push_many_json(items)

    def _collect(self) -> List[dict]:
        """The owners' replies to this worker's last requests, in owner order: one round trip per control server."""
        reps: List[Optional[dict]] = [None] * self.num_owners
        for c, js in self._distinct_cps():
            pending = list(js)
            while pending:
                try:
                    got = c.take_many_json(["%s/reply/%d/%d" % (self.ns, j, self.worker_index) for j in pending],
                                           timeout_s=self.timeout_s)
                except ControlTimeout as e:
                    if self.p2p is not None:
                        # send/recv pairs with that owner are now unmatched: nothing in this process can repair the
                        # stream order of the communicator, so this is the loss of a task, not a transient fault
                        from ..failure import ClusterLostError
                        raise ClusterLostError("no reply from a parameter server within %.0f s" % self.timeout_s) from e
                    raise
                late = []
                for j, r in zip(pending, got):
                    if int(r.get("seq", self._seq)) == self._seq:
                        reps[j] = r
                    else:
                        late.append(j)       # the answer to a request this worker abandoned: take the next one
                pending = late
        self._check_service()
        return reps

    def _check_service(self):
        svc = self._service
        if svc is not None and svc.error is not None:
            raise RuntimeError("the owner service of this worker failed: %r" % (svc.error,))

    def _wait_counter(self, key: str, n: int):
        self.cp0.wait_counter(key, n, timeout_s=self.timeout_s)

    # -- topology
    def prepare(self, model):
        from .comm import init_process_group, pick_backend
        from ..cluster import PLACEMENTS
        srv = self.server
        store = model.store
        dev = store.device
        # control plane: one client per owner's server (colocated: the first server hosts every queue)
        if self.colocated:
            host, port = srv.control_address(0)
            self._cps = [ControlClient(host, port, self.timeout_s)] * self.num_owners
        else:
            self._cps = [ControlClient(*srv.control_address(j), timeout_s=self.timeout_s) for j in range(self.num_owners)]
        self.cp0 = self._cps[0]
        self.my_rank = self.worker_index if self.colocated else srv.rank
        self.owner_rank = (lambda j: j) if self.colocated else srv.ps_rank
        # layout
        variables = [(v.name, v.numel * 4) for v in store.trainable]
        placement = PLACEMENTS[self.placement_name]().assign(variables, self.num_owners)
        self.placement = placement
        self.ranges = owner_ranges(store, placement, self.num_owners)
        self.flat_len = store.flat_len
        opt = model.optimizer
        opt_desc = opt.describe()
        self.var_table = [(v.name, v.offset, v.numel, list(v.shape)) for v in store.trainable]
        # does any variable straddle two owners?  (then a checkpoint assembles it in its writer's staging area)
        self.gather = any(p["n"] != p["numel"] for j in range(self.num_owners)
                          for p in owned_pieces(self.var_table, self.ranges[j]))
        import os
        depth = 1 if (self.max_staleness == 0 or store.flat_len * 4 > (1 << 30)) else 2
        depth = max(1, min(4, int(os.environ.get("DTF_ASYNC_INBOX_DEPTH", depth))))
        if self.colocated:
            # one-sided data plane.  A rank's region: its variable shard, its two optimizer-slot shards, and
            # `depth` inbox slots per worker.  Two slots let a worker compute its next step while the owners still
            # apply the last one; a worker that may not run ahead at all (max_staleness 0) needs one, and so does a
            # model whose second set of slots would cost more than a gigabyte per rank.
            from .doorbell import ControlBlocks
            from .peer_region import PeerRegion
            S = self.slot_len = store.flat_len // self.num_owners
            self.depth = depth
            self.plane = "region"
            replica = ControlBlocks.replica_mode_available(dev)
            self.region_floats = 3 * S + self.num_workers * self.depth * S + \
                (ControlBlocks.area_floats(self.num_workers) if replica else 0)
            self.region = PeerRegion(self.worker_index, self.num_workers, self.region_floats, dev, self.cp0,
                                     name="co/region", timeout_s=self.timeout_s)
            self.blocks = ControlBlocks(self.worker_index, self.num_workers, self.cp0, name="co/ctl",
                                        timeout_s=self.timeout_s, region=self.region if replica else None)
            svc = ParameterServerService(server=srv, owner_index=self.worker_index,
                                         control=ControlClient(*srv.control_address(0), timeout_s=self.timeout_s),
                                         control0=ControlClient(*srv.control_address(0), timeout_s=self.timeout_s),
                                         rank=self.my_rank, device=dev, region=self.region,
                                         num_workers=self.num_workers, worker_rank_of=lambda i: i,
                                         num_owners=self.num_owners, ns=self.ns)
            svc.blocks, svc.depth = self.blocks, self.depth
            self._service = svc
            self._service_thread = threading.Thread(target=self._run_service, name="dtf-owner-%d" % self.worker_index,
                                                    daemon=True)
            plane = {"plane": "region", "stride": S, "depth": self.depth, "words": "replica" if replica else "shm"}
        else:
            # elastic workers: the second process to announce itself as worker i is its restarted successor, and
            # joins the running cluster instead of taking part in its start-up
            from ..failure import elastic_workers
            self.elastic = elastic_workers(srv)
            self._rejoined = self.elastic and self.cp0.add("%s/joined/%d" % (self.ns, self.worker_index), 1) > 1
            if srv.is_chief and not self._rejoined:
                plane = self._choose_plane(dev, depth)
                self.cp0.set_json(self.ns + "/plane", plane)
            else:
                plane = self.cp0.get_json(self.ns + "/plane", timeout_s=self.timeout_s)
            if self.elastic and plane["plane"] != "region":
                raise RuntimeError("DTF_ELASTIC_WORKERS=1 needs the one-sided data plane (every task on one node, one "
                                   "kind of device): a send/recv process group cannot re-admit a restarted worker")
        if srv.is_chief and not self._rejoined:
            layout = {"ranges": self.ranges, "optimizer": opt_desc, "flat_len": store.flat_len,
                      "variables": self.var_table}
            layout.update(plane)
            for c, js in self._distinct_cps():
                for j in js:
                    c.set_json("%s/layout/%d" % (self.ns, j), layout)
        if not self.colocated and plane["plane"] == "region":
            # the ps tasks' regions and control blocks (after the layout is out: they map them once they have it)
            from .doorbell import ControlBlocks
            from .peer_region import PeerRegion
            if dev.type != plane["device"]:
                raise RuntimeError("worker %d computes on %s, but the cluster's one-sided data plane is between %s "
                                   "devices: set DTF_PS_DATA_PLANE=p2p" % (self.worker_index, dev.type, plane["device"]))
            self.plane = "region"
            self.slot_len, self.depth = int(plane["stride"]), int(plane["depth"])
            replica = plane.get("words") == "replica"
            area = ControlBlocks.area_floats(srv.world_size, self.num_workers) if replica else 0
            sizes = region_sizes(self.slot_len, self.depth, self.num_owners, self.num_workers, area)
            self.region_floats = sizes[srv.rank]
            self.region = PeerRegion(srv.rank, srv.world_size, self.region_floats, dev, self.cp0,
                                     name=self.ns + "/region", timeout_s=self.timeout_s, sizes=sizes,
                                     rejoin=self._rejoined)
            self.blocks = ControlBlocks(srv.rank, srv.world_size, self.cp0, name=self.ns + "/ctl",
                                        timeout_s=self.timeout_s, lanes=self.num_workers, owner_rank=srv.ps_rank,
                                        rejoin=self._rejoined, region=self.region if replica else None)
            if replica:
                # a worker's doorbells go to the owners' replicas and its own; everything a worker READS is written by
                # the owners, who re-map a restarted worker (so nobody depends on a mapping of another worker)
                self.blocks.use_ring_ranks([srv.ps_rank(j) for j in range(self.num_owners)] + [srv.rank])
        elif not self.colocated:
            # data plane (after the layout is out: ps tasks join the process group once they have it)
            self.plane = "p2p"
            if not dist.is_initialized():
                host, port = srv.rendezvous_address()
                backend = pick_backend(dev)
                init_process_group(backend, rank=srv.rank, world_size=srv.world_size, master_addr=host,
                                   master_port=port, device=dev if backend == "nccl" else None)
            self.p2p = _P2P(self.my_rank)
            self.p2p.make_direction_groups()
        if self.region is not None:
            self._pulled = [0] * self.num_owners
            self._sent_pulled = [[0] * self.num_owners for _ in range(self.depth)]
            self._acked = [0] * self.num_owners
            if dev.type == "cuda":
                S = self.slot_len
                if self.region._kernels and (not self.colocated or (S % 4 == 0 and store.flat_len == S * self.num_owners)):
                    from .doorbell import PushLink
                    table = None if self.colocated else self._owner_ranges_table()
                    if self.colocated or table.handle:
                        self._link = PushLink(dev, self.region, table, self.blocks, self.worker_index, self.num_owners)
                if self._link is None:
                    self._comm_stream = torch.cuda.Stream(device=dev)
                    self._ev_grad, self._ev_copied = torch.cuda.Event(), torch.cuda.Event()
        if self._service_thread is not None:
            self._service_thread.start()
        if self.region is not None:
            import atexit
            atexit.register(self.close)       # the service thread must be gone before the interpreter tears down
        # chief initialises the owners, everybody else waits (chief-initialises / others-wait)
        self.versions = [0] * self.num_owners
        if self._rejoined:
            self.cp0.get(self.ns + "/initialized", timeout_s=self.timeout_s)
            self._rejoin_owners()
        elif srv.is_chief:
            self._wait_counter(self.ns + "/ready", self.num_owners)
            self._init_owners(store, int(store.global_step.item()))
            self.cp0.set(self.ns + "/initialized", b"1")
        else:
            self.cp0.get(self.ns + "/initialized", timeout_s=self.timeout_s)
        self.pull(model)
        if self.max_staleness is not None and self.region is None:
            self.cp0.clock_set_wait("workers", self.worker_index, 0, None)

    def _rejoin_owners(self):
        """Restarted worker: continue this worker's lanes.  Its predecessor may have died between two owners' doorbells,
        so every owner is told the one gradient number to continue from - nothing rung below it is waited for, and a
        gradient that was complete in some inboxes only is dropped there.  The number also serves as this worker's
        clock, so it starts no further back than the slowest running worker (it must not hold the others to a
        staleness bound measured against steps its predecessor never made)."""
        w, b = self.worker_index, self.blocks
        # first the owners map this process's new region and bring its replica of the control words up to date
        # (nothing to do for words in shared memory); only then can this process read them
        base = self.cp0.add("%s/rejoin_map_done/%d" % (self.ns, w), 0)
        self._post("rejoin_map")
        self._wait_counter("%s/rejoin_map_done/%d" % (self.ns, w), base + self.num_owners)
        seq = max(b.door(j, w) for j in range(self.num_owners))
        running = [c for c in (b.clock(i) for i in range(self.num_workers) if i != w) if c < (1 << 59)]
        if running:
            seq = max(seq, min(running))
        base = self.cp0.add("%s/rejoin_done/%d" % (self.ns, w), 0)
        self._post("rejoin", {"seq": seq})
        self._wait_counter("%s/rejoin_done/%d" % (self.ns, w), base + self.num_owners)
        self._set_clock(seq)
        self.clock = seq
        self._acked = [seq] * self.num_owners
        import sys
        sys.stderr.write("worker %d re-admitted at gradient number %d (control words: %s)\n"
                         % (w, seq, "replicas in device memory" if b.replicas is not None else "shared memory"))

    def _set_clock(self, value: int):
        """Publish this worker's clock outside a push (re-admission, end of training).  With replicated control words
        and ps tasks as owners that is owner 0's job: it holds a current mapping of every replica."""
        b = self.blocks
        if b.replicas is not None and not self.colocated:
            base = self.cp0.add("%s/clock_done/%d" % (self.ns, self.worker_index), 0)
            self._post("clock", {"value": int(value)}, owners=[0])
            self._wait_counter("%s/clock_done/%d" % (self.ns, self.worker_index), base + 1)
        else:
            b.set_clock(self.worker_index, value)

    def _choose_plane(self, dev, depth: int) -> dict:
        """Chief, ps tasks as owners: the one-sided plane (peer regions + doorbell words) when every task runs on this
        node and the ps tasks compute on the same kind of device as the workers; otherwise send/recv.
        ``DTF_PS_DATA_PLANE=p2p|region`` overrides the choice."""
        import os
        import socket
        want = os.environ.get("DTF_PS_DATA_PLANE", "")
        if want not in ("", "p2p", "region"):
            raise ValueError("DTF_PS_DATA_PLANE must be 'p2p' or 'region'")
        if want == "p2p":
            return {"plane": "p2p"}
        here = "%s@%s" % (dev.type, socket.gethostname())
        where = [self.cp0.get("%s/where/%d" % (self.ns, j), timeout_s=self.timeout_s).decode()
                 for j in range(self.num_owners)]
        hosts = {split_host(a) for job in ("ps", "worker") for a in self.server.cluster.job_tasks(job)}
        one_node = all(w == here for w in where) and (len(hosts) == 1 or hosts <= _LOOPBACK)
        if not one_node:
            if want == "region":
                raise RuntimeError("DTF_PS_DATA_PLANE=region needs every task on this node and on %s devices "
                                   "(ps tasks: %s)" % (dev.type, where))
            return {"plane": "p2p"}
        nmax = max([_ranges_numel(r) for r in self.ranges] + [1])
        from .doorbell import ControlBlocks
        return {"plane": "region", "device": dev.type, "stride": (nmax + 63) // 64 * 64, "depth": depth,
                "words": "replica" if ControlBlocks.replica_mode_available(dev) else "shm"}

    def close(self, grace_s: float = 20.0):
        """End this worker's owner thread: wait for the chief's shutdown request to reach it, and if none comes
        within ``grace_s`` (the chief is gone) queue one ourselves.  Idempotent; registered with ``atexit``."""
        t, self._service_thread = self._service_thread, None
        if t is not None and t.is_alive():
            t.join(grace_s)
            if t.is_alive():
                try:
                    self.cp0.push_json("%s/req/%d" % (self.ns, self.worker_index), {"op": "shutdown"})
                except Exception:
                    pass
                t.join(10.0)
        if self.region is not None and (t is None or not t.is_alive()):
            if self._link is not None:
                self._link.close()
                self._link = None
            if self.blocks is not None:
                self.blocks.close()             # (replica mode: its views point into the region)
                self.blocks = None
            self.region.close()
            self.region = None

    def _run_service(self):
        try:
            self._service.serve()
        except BaseException:       # noqa: BLE001 - kept in service.error; the training thread raises it
            import sys
            import traceback
            sys.stderr.write("owner service of worker %d failed:\n%s" % (self.worker_index, traceback.format_exc()))

    def _stage(self, j: int, store, kind: str = "p") -> torch.Tensor:
        key = (j, kind)
        if key not in self._staging:
            self._staging[key] = torch.empty(max(_ranges_numel(self.ranges[j]), 1), dtype=torch.float32,
                                             device=store.device)
        return self._staging[key]

    def _init_owners(self, store, gs0: int):
        """Chief: ship the initial values to every owner, reset their slots and the shared step."""
        base = self.cp0.add(self.ns + "/init_done", 0)
        if self.region is not None:
            if self.colocated:
                for j in range(self.num_owners):
                    b, e = self.ranges[j][0]
                    self.region.put(j, 0, store.flat[b:e])   # straight into the owner's variable shard
            else:
                self.region.move_ranges(self._owner_ranges_table(), False, 0, store.flat)
            self.region.sync()
            self._post("init", {"global_step": gs0})
        else:
            self._post("init", {"global_step": gs0})
            sends = []
            for j in range(self.num_owners):
                n = _ranges_numel(self.ranges[j])
                if n:
                    sends.append((_pack(store.flat, self.ranges[j], self._stage(j, store, "g"))[:n].contiguous(),
                                  self.owner_rank(j), _TAG_INIT))
            self.p2p.exchange(sends, [])
        self._wait_counter(self.ns + "/init_done", base + self.num_owners)
        if self.region is None:
            self.cp0.add("global_step", gs0 - self.cp0.add("global_step", 0))
        self.last_global_step = gs0

    # -- data movement: colocated owners (peer regions + doorbell words; nothing here crosses a socket)
    #: set by the session's heartbeat hook: raises ClusterLostError when the failure detector has lost a task
    external_health_check = None

    def _health(self):
        if self.external_health_check is not None:
            self.external_health_check()
        self._check_service()
        svc = self._service
        if svc is not None and svc.pump is not None and svc.pump.error():
            raise RuntimeError("the owner pump of this worker failed with status %d" % svc.pump.error())

    def _spin(self, pred, what: str) -> float:
        from .doorbell import spin_until
        try:
            return spin_until(pred, self.timeout_s, what, check=self._health)
        except TimeoutError as e:
            raise ControlTimeout(str(e)) from e

    def _min_ack(self) -> int:
        b, w = self.blocks, self.worker_index
        return min(b.ack(j, w) for j in range(self.num_owners))

    def _account_acks(self):
        """Histogram the staleness of every push acknowledged since the last call: owner updates applied between the
        pull a gradient was computed from and its own apply.  Must run before a slot's words are reused."""
        b, w, D = self.blocks, self.worker_index, self.depth
        for j in range(self.num_owners):
            upto = b.ack(j, w)
            for seq in range(self._acked[j] + 1, upto + 1):
                slot = (seq - 1) % D
                sent = self._sent_pulled[slot]
                pulled = sent[j] if sent is not None else b.pull_version(j, w, slot)
                st = max(0, b.ack_version(j, w, slot) - 1 - pulled)
                self.staleness_hist[st] = self.staleness_hist.get(st, 0) + 1
            self._acked[j] = upto

    def _drain(self):
        """Wait until the owners have applied everything this worker has pushed."""
        if self.clock:
            self._spin(lambda: self._min_ack() >= self.clock, "the owners to apply this worker's gradients")
        self._account_acks()

    def _owner_ranges_table(self):
        """ps tasks as owners: every flat range of every owner's variables against its place in the owner's packed
        shard - one table, so that a push or a pull is one launch whatever the placement interleaves."""
        if self._range_table is None:
            descs = []
            for j in range(self.num_owners):
                pos = 0
                for b, e in self.ranges[j]:
                    descs.append((self.owner_rank(j), pos, b, e - b))
                    pos += e - b
            self._range_table = self.region.make_ranges(descs)
        return self._range_table

    def _region_pull(self, store, stream_ptr: Optional[int] = None):
        """One-sided gets of every owner's variables straight into the flat buffer, on the current stream."""
        for j in range(self.num_owners):
            self._pulled[j] = self.blocks.version(j)
        if self.colocated:
            # owner j's shard is flat[j*S:(j+1)*S]: one launch, all links
            self.region.gather(0, store.flat, stream_ptr=stream_ptr)
        else:
            # ps tasks: the flat ranges of the variables each one owns, all owners in one launch
            self.region.move_ranges(self._owner_ranges_table(), True, 0, store.flat, stream_ptr=stream_ptr)
        self.versions = list(self._pulled)
        self.last_global_step = self.blocks.global_step()

    def _region_step(self, model):
        """Push this step's gradient shards, then fetch the variables for the next step."""
        store = model.store
        on_gpu = store.device.type == "cuda"
        W, D, w = self.num_owners, self.depth, self.worker_index
        t = self.clock                                   # gradients pushed so far; this one is number t + 1
        slot = t % D
        # the inbox slot is free once the gradient that used it last (number t + 1 - D) has been applied
        if t >= D:
            self.wait_seconds += self._spin(lambda: self._min_ack() >= t + 1 - D, "a free inbox slot")
        self._account_acks()
        inbox = 3 * self.slot_len + (w * D + slot) * self.slot_len
        self._sent_pulled[slot] = list(self._pulled)
        cur_ptr = None
        if self._link is not None:
            # one native call queues the whole push: order behind the backward pass, gradient into the owners' inboxes
            # (one launch), doorbells, release of the gradient buffer
            cur_ptr = torch.cuda.current_stream(store.device).cuda_stream
            self._link.push(cur_ptr, store.flat_grad, inbox, self.slot_len if self.colocated else 0, self._pulled, slot,
                            t + 1, t + 1)
        else:
            self._push_stepwise(store, on_gpu, inbox, slot, t + 1)
        self.clock = t + 1
        # how far this worker may run ahead of the owners: one gradient in flight per spare inbox slot
        lag = D - 1
        if self._min_ack() < self.clock - lag:
            self.wait_seconds += self._spin(lambda: self._min_ack() >= self.clock - lag, "the owners' acknowledgement")
        if self.max_staleness is not None:
            # stale-synchronous bound over the workers' clocks (completed pushes), read from owner 0's block
            s = self.max_staleness
            if self.clock - self.min_clock() > s:
                self.wait_seconds += self._spin(lambda: self.clock - self.min_clock() <= s,
                                                "the slowest worker (staleness bound)")
        self._region_pull(store, cur_ptr)
        if self.last_global_step:
            store.global_step.fill_(self.last_global_step)

    # -- the whole step as ONE kernel: variables read from, gradients written to the ps tasks' memory
    def _direct_plan(self, model, batch: int):
        """ps tasks as owners, GPU, the reference network at a batch of at most 64: the fused gradient kernel can read
        every variable straight from its owner's variable plane and write its gradient straight into this worker's
        inbox slot there (peer-mapped memory), and ring the doorbells itself.  ``None`` when that does not apply
        (``DTF_ASYNC_DIRECT=0`` switches it off)."""
        if batch in self._direct_plans:
            return self._direct_plans[batch]
        import os
        plan = None
        store = model.store
        b, w = self.blocks, self.worker_index
        gp = model.fused_grad_plan(batch) if hasattr(model, "fused_grad_plan") else None
        if (gp is not None and self.region is not None and not self.colocated and store.device.type == "cuda"
                and self.num_owners <= 8 and os.environ.get("DTF_ASYNC_DIRECT", "1") != "0" and b is not None):
            # control blocks in replica mode: the kernel rings the doorbells itself and reads the shared step from owner
            # 0's word; in shared memory: the doorbells are rung behind it and the step comes from the worker's mirror
            in_kernel = b.replicas is not None
            where = []                       # per variable: (owner, position in the owner's packed shard)
            eff_vars = [(off, n) for off, n in zip(gp["offsets"], self._plan_numels(gp))]
            for off, n in eff_vars:
                hit = None
                for j in range(self.num_owners):
                    pos = 0
                    for lo, hi in self.ranges[j]:
                        if lo <= off and off + n <= hi:
                            hit = (j, pos + off - lo)
                        pos += hi - lo
                where.append(hit)
            if all(h is not None for h in where):
                S, D = self.slot_len, self.depth
                var_ptrs = [self.region.peer_address(self.owner_rank(j), pos) for j, pos in where]
                grad_ptrs = [[self.region.peer_address(self.owner_rank(j), 3 * S + (w * D + slot) * S + pos) for j, pos in where]
                             for slot in range(D)]
                plan = dict(gp=gp, var_ptrs=var_ptrs, grad_ptrs=grad_ptrs, in_kernel=in_kernel,
                            step=b.global_step_address() if in_kernel else store.global_step.data_ptr())
        self._direct_plans[batch] = plan
        return plan

    @staticmethod
    def _plan_numels(gp):
        d0, d1, d2, d3 = gp["dims"]
        return [d0 * d1, d1, d1 * d2, d2, d2 * d3, d3]

    def train_step_direct(self, model, x, y) -> bool:
        """One training step as one kernel launch (see ``_direct_plan``); False when the step has to take the general
        path (``compute_gradients`` + ``apply_gradients``)."""
        if not (x.is_cuda and x.dim() == 2 and y.dim() == 2 and y.dtype == torch.float32 and x.shape[0] <= 64):
            return False
        plan = self._direct_plan(model, int(x.shape[0]))
        if plan is None:
            return False
        from ..ops import functional as F
        t_begin = time.perf_counter()
        w0 = self.wait_seconds
        D = self.depth
        t = self.clock
        slot = t % D
        if t >= D:
            self.wait_seconds += self._spin(lambda: self._min_ack() >= t + 1 - D, "a free inbox slot")
        self._account_acks()
        gp = plan["gp"]
        if plan["in_kernel"]:
            self._sent_pulled[slot] = None      # the kernel reads the owners' versions itself: taken from the lane words
        else:
            for j in range(self.num_owners):
                self._pulled[j] = self.blocks.version(j)
            self._sent_pulled[slot] = list(self._pulled)
            if self.last_global_step:
                model.store.global_step.fill_(self.last_global_step)
        F.mlp3_fused_grad_direct(x.contiguous(), y.contiguous(), plan["var_ptrs"], plan["grad_ptrs"][slot], gp["dims"], True,
                                 model._extra_categorical, gp["l1"].fused_keep, gp["l2"].fused_keep, gp["l1"].seed,
                                 gp["l2"].seed, plan["step"], model.stats, self.blocks if plan["in_kernel"] else None,
                                 self.worker_index, self.num_owners, slot, t + 1, t + 1)
        if not plan["in_kernel"]:
            from .doorbell import ring
            ring(torch.cuda.current_stream(model.store.device).cuda_stream, True, self.blocks, self.worker_index,
                 self.num_owners, self._pulled, slot, t + 1, t + 1)
        self.clock = t + 1
        lag = D - 1
        if self._min_ack() < self.clock - lag:
            self.wait_seconds += self._spin(lambda: self._min_ack() >= self.clock - lag, "the owners' acknowledgement")
        if self.max_staleness is not None:
            s = self.max_staleness
            if self.clock - self.min_clock() > s:
                self.wait_seconds += self._spin(lambda: self.clock - self.min_clock() <= s,
                                                "the slowest worker (staleness bound)")
        self.last_global_step = max(self.last_global_step, self.blocks.global_step())
        self._flat_stale = True
        self.comm_seconds += time.perf_counter() - t_begin - (self.wait_seconds - w0)
        return True

    def ensure_fresh(self, model):
        """Before anything reads the worker's own copy of the variables (evaluation, prediction): direct steps never
        refresh it."""
        if self._flat_stale:
            self.pull(model)
            model.store.global_step.fill_(self.last_global_step)

    def _push_stepwise(self, store, on_gpu: bool, inbox: int, slot: int, seq: int):
        """The push without the native link (CPUs; GPUs with DTF_PEER_KERNELS=0): the same steps, one call each."""
        from .doorbell import ring
        W, w = self.num_owners, self.worker_index
        if on_gpu:
            cur = torch.cuda.current_stream(store.device)
            self._ev_grad.record(cur)
            self._comm_stream.wait_event(self._ev_grad)
            ctx = torch.cuda.stream(self._comm_stream)
        else:
            ctx = _NullContext()
        with ctx:
            if self.colocated:
                # shard j of the gradient into this worker's slot of owner j's inbox: one launch, all links at once
                self.region.scatter(inbox, store.flat_grad)
            else:
                self.region.move_ranges(self._owner_ranges_table(), False, inbox, store.flat_grad)
            if on_gpu:
                self._ev_copied.record(self._comm_stream)
            ring(self._comm_stream.cuda_stream if on_gpu else 0, on_gpu, self.blocks, w, W, self._pulled, slot, seq, seq)
        if on_gpu:
            cur.wait_event(self._ev_copied)              # the next backward pass may overwrite the gradient buffer

    def min_clock(self) -> int:
        """The slowest worker's clock (pushes completed), as the staleness bound sees it."""
        if self.blocks is not None:
            # (this worker's own clock is self.clock; its word may trail it by the pushes still in flight)
            return self.blocks.min_clock(skip=self._absent_workers() + (self.worker_index,))
        return self.cp0.clock_min("workers")

    #: set by the session's heartbeat hook: the tolerated members (elastic workers) that are silent right now
    external_absent = None

    def _absent_workers(self):
        """Indices of workers the failure detector reports absent: a dead worker's clock stands still and must not
        hold the others to the staleness bound until its successor has rejoined."""
        if not self.elastic or self.external_absent is None:
            return ()
        return tuple(int(m.split(":")[1]) for m in self.external_absent() if m.startswith("worker:"))

    # -- data movement: ps tasks (torch.distributed send/recv + folded control traffic)
    def _receive_params(self, store):
        """Receive descriptors for the parameters of every owner (after the owners have been asked)."""
        recvs, bufs = [], {}
        for j in range(self.num_owners):
            n = _ranges_numel(self.ranges[j])
            if not n:
                continue
            single = len(self.ranges[j]) == 1
            buf = store.flat[self.ranges[j][0][0]:self.ranges[j][0][1]] if single else self._stage(j, store, "p")[:n]
            bufs[j] = buf
            recvs.append((buf, self.owner_rank(j), _TAG_PULL))
        return recvs, bufs

    def pull(self, model):
        """The owners' current variables into this worker's flat buffer (everything it pushed has been applied)."""
        store = model.store
        if self.region is not None:
            self._drain()
            self._region_pull(store)
            self._flat_stale = False
            return
        self._post("pull")
        recvs, bufs = self._receive_params(store)
        self.p2p.exchange([], recvs)
        for j, buf in bufs.items():
            _unpack(buf, store.flat, self.ranges[j])
        reps = self._collect()
        for j, rep in enumerate(reps):
            self.versions[j] = rep["version"]

    def apply_gradients(self, model):
        """Push this step's gradients to every owner and pull the refreshed parameters back."""
        store = model.store
        t_begin = time.perf_counter()
        if self.region is not None:
            w0 = self.wait_seconds
            self._region_step(model)
            self.comm_seconds += time.perf_counter() - t_begin - (self.wait_seconds - w0)
            return
        self._post("pushpull", lambda j: {"version": self.versions[j]})
        sends = []
        for j in range(self.num_owners):
            n = _ranges_numel(self.ranges[j])
            if not n:
                continue
            g = store.flat_grad[self.ranges[j][0][0]:self.ranges[j][0][1]] if len(self.ranges[j]) == 1 \
                else _pack(store.flat_grad, self.ranges[j], self._stage(j, store, "g"))[:n]
            sends.append((g, self.owner_rank(j), _TAG_PUSH))
        recvs, bufs = self._receive_params(store)
        self.p2p.exchange(sends, recvs)
        for j, buf in bufs.items():
            _unpack(buf, store.flat, self.ranges[j])
        reps = self._collect()
        for j, rep in enumerate(reps):
            self.versions[j] = rep["version"]
            self.staleness_hist[rep["staleness"]] = self.staleness_hist.get(rep["staleness"], 0) + 1
            if rep["global_step"] >= 0:
                self.last_global_step = rep["global_step"]
        if self.last_global_step:
            store.global_step.fill_(self.last_global_step)
        self.clock += 1
        t_comm = time.perf_counter()
        self.comm_seconds += t_comm - t_begin
        if self.max_staleness is not None:
            # publish the new clock and wait for permission to start the next step, in one round trip
            self.cp0.clock_set_wait("workers", self.worker_index, self.clock, self.max_staleness, timeout_s=self.timeout_s)
            self.wait_seconds += time.perf_counter() - t_comm

    def global_step(self) -> int:
        """Cluster-wide global step.  Colocated owners: the word owner 0 publishes, after this worker's own pushes
        have been applied (no round trip); ps tasks: one control-plane round trip."""
        if self.blocks is not None:
            self._drain()
            self.last_global_step = self.blocks.global_step()
            return self.last_global_step
        self.last_global_step = self.cp0.add("global_step", 0)
        return self.last_global_step

    def cached_global_step(self) -> int:
        """The shared step without waiting for anything: what session hooks read after every ``run``.  Colocated
        owners publish it in shared memory; with ps tasks it is the value that came back with the last push."""
        if self.blocks is not None:
            self.last_global_step = max(self.last_global_step, self.blocks.global_step())
        return self.last_global_step

    def hbm_report(self, model) -> dict:
        """Bytes this rank holds for the model: what a worker needs (variables + gradients) and what it owns."""
        F = model.store.flat_len * 4
        out = {"worker_variables": F, "worker_gradients": F,
               "worker_slots": sum(t.numel() * 4 for t in model.store.slots.values())}
        if self.region is not None and self.colocated:
            S = self.slot_len * 4
            out.update({"owner_variables": S, "owner_slots": 2 * S, "owner_inbox": self.num_workers * self.depth * S,
                        "region_total": self.region_floats * 4})
        elif self.region is not None:
            S = self.slot_len * 4        # held by each ps task, not by this worker
            out.update({"ps_variables": S, "ps_slots": 2 * S, "ps_inbox": self.num_workers * self.depth * S,
                        "region_total": self.region_floats * 4})
        return out

    def publish_state(self, model, is_chief: bool):
        """Session creation: the chief (re-)initialises the owners - from the checkpoint it has just restored, slots
        and ``beta*_power`` included, or else from its store - and the other workers wait for that, then everybody
        pulls."""
        store = model.store
        first = not getattr(self, "_published", False)
        self._published = True
        if self._rejoined:
            # the owners hold the live state (a chief that comes back must not roll them back to a checkpoint)
            model._restored_from = None
            self.pull(model)
            store.global_step.fill_(self.global_step())
            return
        if is_chief:
            prefix = getattr(model, "_restored_from", None)
            gs = int(store.global_step.item())
            if prefix:
                model._restored_from = None
                self.restore(model, prefix, uuid.uuid4().hex)
                if self.region is None:
                    self.cp0.add("global_step", gs - self.cp0.add("global_step", 0))
                self.last_global_step = gs
            else:
                self._init_owners(store, gs)
        if first:
            self.cp0.barrier(self.ns + "/publish", self.num_workers, timeout_s=self.timeout_s)
        self.pull(model)
        store.global_step.fill_(self.global_step())

    def reduce_stats(self, stats):
        return stats

    # -- checkpoints: each owner writes its own data shard, the chief merges the index
    def save(self, model, prefix: str, token: str):
        slots = getattr(model.optimizer, "slot_checkpoint_names", {})
        gs = self.global_step()
        base = self.cp0.add("%s/save_done/%s" % (self.ns, token), 0)
        # variables that are not trained (moving statistics of batch normalisation) live on the workers; the chief's go
        # into shard 0 with the request (they are a few channels' worth of numbers)
        extras = {v.name: {"shape": list(v.shape), "values": v.tensor.detach().reshape(-1).cpu().tolist()}
                  for v in model.store.variables.values()
                  if not v.trainable and v.tensor is not None and v.tensor.dtype == torch.float32}
        self._post("save", {"prefix": prefix, "num_shards": self.num_owners, "slots": slots, "global_step": gs,
                            "optimizer": model.optimizer.name, "token": token, "gather": self.gather, "extras": extras})
        self._wait_counter("%s/save_done/%s" % (self.ns, token), base + self.num_owners)
        self._check_service()
        from ..saver import merge_shards
        merge_shards(prefix, self.num_owners)
        return prefix

    def restore(self, model, prefix: str, token: str):
        """Owners reload variables, optimizer slots, ``beta*_power`` and the step from the checkpoint files."""
        slots = getattr(model.optimizer, "slot_checkpoint_names", {})
        self._post("restore", {"prefix": prefix, "slots": slots, "token": token})
        self._wait_counter("%s/restore_done/%s" % (self.ns, token), self.num_owners)
        self._check_service()

    def owner_stats(self) -> List[dict]:
        out = []
        for j in range(self.num_owners):
            self._cps[j].delete("%s/stats/%d" % (self.ns, j))
            self._cps[j].push_json("%s/req/%d" % (self.ns, j), {"op": "stats"})
            out.append(self._cps[j].get_json("%s/stats/%d" % (self.ns, j), timeout_s=self.timeout_s))
        return out

    def worker_done(self):
        """Every worker calls this once after its last step."""
        if self.blocks is not None:
            self._drain()
            self._set_clock(1 << 60)                                # a finished worker never holds others back
        else:
            self.cp0.clock_set("workers", self.worker_index, 1 << 60)
        self.cp0.add("workers_done", 1)

    def shutdown_owners(self, wait_for_workers: bool = True):
        """Chief only: ends ``server.join()`` in every ps task and every colocated owner thread (the reference has
        no such protocol).  Waits until every worker has reported ``worker_done`` so nobody is cut off mid-step."""
        if wait_for_workers:
            try:
                self.cp0.wait_counter("workers_done", self.num_workers, timeout_s=self.timeout_s)
            except ControlTimeout:
                pass
        for j in range(self.num_owners):
            self._cps[j].push_json("%s/req/%d" % (self.ns, j), {"op": "shutdown"})
        if self.colocated and self.server.num_ps > 0:
            # ps tasks that only served the control plane leave join() too
            self.server.shutdown_parameter_servers()

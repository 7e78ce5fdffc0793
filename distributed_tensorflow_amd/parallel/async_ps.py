"""Asynchronous parameter-server training with bounded staleness.

This is the reference's actual mode (/root/reference/example.py:136-139,168-170): every
worker trains on its own data, reads the variables from the parameter servers, and sends
gradients that the servers apply one by one, lock-free, with the plain (non-synchronising)
optimizer; ``global_step`` is shared, so ``StopAtStepHook(last_step=N)`` is a cluster-wide
budget.

Re-designed for one MI355X node:

* an *owner* holds ``[begin, end)`` ranges of the flat variable buffer together with the
  optimizer slots for them, resident in its HBM; owners are either the ``ps`` tasks of the
  cluster spec (separate processes, as in the reference) or - ``colocated`` - a service
  thread inside every worker process, so that 8 workers on 8 GPUs shard the variables
  across all eight HBMs;
* per step a worker sends each owner ONE packed gradient message and receives ONE packed
  parameter message back (``torch.distributed`` send/recv: RCCL over xGMI between GPUs,
  gloo between CPU processes) instead of TF's one RecvTensor RPC per variable per direction;
* the owner applies the fused apply-gradients kernel to its shard on arrival - updates from
  different workers interleave freely across owners (Hogwild-style, ``use_locking=False``
  semantics), each shard update itself is atomic;
* *bounded staleness* (stale-synchronous parallel): every worker publishes its clock (number
  of pushes) to the control plane and blocks before a step while
  ``clock - min(all clocks) > max_staleness``.  ``max_staleness=None`` is the reference's
  unbounded behaviour.  The observed staleness of every push (owner version at apply minus
  owner version at the worker's last pull) is histogrammed.
"""
from __future__ import annotations

import threading
import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from ..control import ControlClient, ControlTimeout
from ..variables import VariableStore

_REQ_QUEUE = "ps/req"
_NCCL_LOCK = threading.Lock()   # serialises the ENQUEUE of point-to-point ops of the two threads of a process


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


# --------------------------------------------------------------------------- data-plane helper
class _P2P:
    """send/recv between global ranks with the loop-back (same process) case short-circuited.

    RCCL matches the sends and receives of a pair of ranks purely by issue order (message tags are ignored), and a
    process here has two threads talking to the same peers - the worker thread (gradients out, parameters in) and,
    when owners are colocated, the owner thread (gradients in, parameters out).  Each direction therefore gets its
    own communicator: on ``push`` only worker threads send and only owner threads receive, on ``pull`` the reverse,
    so every (source, destination) pair of a communicator carries one strictly ordered stream.
    """

    def __init__(self, rank: int):
        self.rank = rank
        self.mail: Dict[Tuple[int, int], List[torch.Tensor]] = {}
        self.cv = threading.Condition()
        self.push_group = None      # worker -> owner (gradients, initial values)
        self.pull_group = None      # owner -> worker (parameters)

    def make_direction_groups(self):
        """Collective over the whole process group: every member calls this once, right after joining."""
        if self.push_group is None and dist.is_initialized() and dist.get_world_size() > 1:
            self.push_group = dist.new_group()
            self.pull_group = dist.new_group()

    def _group(self, tag: int):
        return self.pull_group if tag == _TAG_PULL else self.push_group

    def send(self, t: torch.Tensor, dst: int, tag: int):
        if dst == self.rank:
            with self.cv:
                self.mail.setdefault((tag, 0), []).append(t.clone())
                self.cv.notify_all()
            return
        if _staged(t):
            t = t.cpu()
        with _NCCL_LOCK:
            w = dist.isend(t, dst=dst, tag=tag, group=self._group(tag))
        w.wait()

    def recv(self, t: torch.Tensor, src: int, tag: int):
        if src == self.rank:
            with self.cv:
                while not self.mail.get((tag, 0)):
                    self.cv.wait()
                t.copy_(self.mail[(tag, 0)].pop(0))
            return
        h = torch.empty(t.shape, dtype=t.dtype) if _staged(t) else t
        with _NCCL_LOCK:
            w = dist.irecv(h, src=src, tag=tag, group=self._group(tag))
        w.wait()
        if h is not t:
            t.copy_(h)


    def exchange(self, sends, recvs):
        """Post every send ``(tensor, dst, tag)`` and receive ``(tensor, src, tag)`` at once, then wait for all.

        A worker must never block on one owner before it has posted its traffic to the others: owners
        serve workers in arrival order, so blocking sends issued one owner at a time can form a cycle.
        On NCCL the operations are issued as one group (they progress concurrently on the stream).
        """
        local_recvs = []
        send_ops, recv_ops = [], []
        for t, dst, tag in sends:
            if dst == self.rank:
                with self.cv:
                    self.mail.setdefault((tag, 0), []).append(t.clone())
                    self.cv.notify_all()
            else:
                send_ops.append(dist.P2POp(dist.isend, t.cpu() if _staged(t) else t, dst, group=self._group(tag), tag=tag))
        staged = []
        for t, src, tag in recvs:
            if src == self.rank:
                local_recvs.append((t, tag))
            else:
                h = t
                if _staged(t):
                    h = torch.empty(t.shape, dtype=t.dtype)
                    staged.append((t, h))
                recv_ops.append(dist.P2POp(dist.irecv, h, src, group=self._group(tag), tag=tag))
        works = []
        with _NCCL_LOCK:           # one batch per communicator (a batch may not mix groups)
            if send_ops:
                works += dist.batch_isend_irecv(send_ops)
            if recv_ops:
                works += dist.batch_isend_irecv(recv_ops)
        for t, tag in local_recvs:
            with self.cv:
                while not self.mail.get((tag, 0)):
                    self.cv.wait()
                t.copy_(self.mail[(tag, 0)].pop(0))
        for w in works:
            w.wait()
        for t, h in staged:
            t.copy_(h)


def _staged(t: torch.Tensor) -> bool:
    """GPU tensor on the gloo backend (ranks time-sharing one GPU): moved through host memory explicitly, see
    ``Communicator._staged``."""
    return t.is_cuda and dist.get_backend() == "gloo"


_TAG_PUSH, _TAG_PULL, _TAG_INIT = 11, 12, 13


# --------------------------------------------------------------------------- owner / parameter-server side
class ParameterServerService:
    """Owner of a set of flat ranges: serves init / pushpull / pull / save / restore / shutdown."""

    def __init__(self, server=None, owner_index: Optional[int] = None, control: Optional[ControlClient] = None,
                 rank: Optional[int] = None, device=None, p2p: Optional[_P2P] = None, num_workers: Optional[int] = None,
                 worker_rank_of=None):
        self.server = server
        self.owner_index = owner_index if owner_index is not None else server.task_index
        if control is None:
            host, port = server.control_address(self.owner_index)
            control = ControlClient(host if host not in ("0.0.0.0",) else "127.0.0.1", port)
        self.cp = control
        self.device = device if device is not None else server.device()
        self.rank = rank if rank is not None else server.rank
        self.num_workers = num_workers if num_workers is not None else server.num_workers
        self.worker_rank_of = worker_rank_of or (lambda i: server.worker_rank(i))
        self.p2p = p2p
        self.queue = "%s/%d" % (_REQ_QUEUE, self.owner_index)
        self.ranges: List[Tuple[int, int]] = []
        self.var = self.grad = None
        self.slots: Dict[str, torch.Tensor] = {}
        self.version = 0            # updates applied by this owner
        self.step_t = None
        self.opt = None
        self.staleness_hist: Dict[int, int] = {}
        self._owns_group = False

    # a tiny VariableStore look-alike so Optimizer.apply works on the owned shard
    class _Shard:
        def __init__(self, svc):
            self.flat = svc.var
            self.flat_grad = svc.grad
            self.flat_len = svc.var.numel()
            self.global_step = svc.step_t
            self.beta_powers = svc.beta_powers
            self._svc = svc

        def slot(self, name):
            s = self._svc.slots
            if name not in s:
                s[name] = torch.zeros_like(self._svc.var)
            return s[name]

    def _join_data_plane(self):
        if dist.is_initialized():
            return
        from .comm import init_process_group
        host, port = self.server.rendezvous_address()
        from .comm import pick_backend
        backend = pick_backend(self.device)
        init_process_group(backend, rank=self.rank, world_size=self.server.world_size, master_addr=host,
                           master_port=port, device=self.device if backend == "nccl" else None)
        self._owns_group = True

    def _configure(self, layout: dict):
        from .. import optimizers
        self.ranges = [tuple(r) for r in layout["ranges"][self.owner_index]]
        n = _ranges_numel(self.ranges)
        self.var = torch.zeros(max(n, 1), dtype=torch.float32, device=self.device)
        self.grad = torch.zeros_like(self.var)
        self.step_t = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.beta_powers = torch.ones(2, dtype=torch.float32, device=self.device)
        o = layout["optimizer"]
        self.opt = optimizers.Adam(**o["args"]) if o["name"] == "Adam" else optimizers.SGD(**o["args"])
        self.shard = ParameterServerService._Shard(self)
        self.opt.create_slots(self.shard)
        self.n = n

    def serve(self):
        """Request loop; returns after a ``shutdown`` request."""
        # the chief posts the layout; a cluster that trains without its ps tasks (synchronous replicas)
        # only ever sends "shutdown"
        layout = None
        while layout is None:
            try:
                layout = self.cp.get_json("ps/layout", timeout_s=0.5)
            except ControlTimeout:
                if self.cp.qlen(self.queue) > 0:
                    req = self.cp.pop_json(self.queue, timeout_s=1.0)
                    if req.get("op") == "shutdown":
                        self.cp.add("ps/stopped", 1)
                        return
        if self.p2p is None:
            self._join_data_plane()      # only clusters that use their ps tasks put them on the data plane
            self.p2p = _P2P(self.rank)
            self.p2p.make_direction_groups()
        self._configure(layout)
        self.cp.add("ps/ready", 1)
        while True:
            try:
                req = self.cp.pop_json(self.queue, timeout_s=5.0)
            except ControlTimeout:
                continue
            op = req["op"]
            if op == "shutdown":
                break
            wr = self.worker_rank_of(req["worker"]) if "worker" in req else None
            if op == "init":
                if self.n:
                    self.p2p.recv(self.var[:self.n], wr, _TAG_INIT)
                self.version = 0
                self.step_t.fill_(int(req.get("global_step", 0)))
                self.cp.add("ps/init_done", 1)
            elif op == "pushpull":
                if self.n:
                    self.p2p.recv(self.grad[:self.n], wr, _TAG_PUSH)
                    self.opt.apply(self.shard, 0, self.n, grad_scale=1.0, grad=self.grad, step=self.step_t)
                    self.opt.finish_step(self.shard)
                st = self.version - int(req.get("version", self.version))
                self.staleness_hist[st] = self.staleness_hist.get(st, 0) + 1
                self.version += 1
                if self.owner_index == 0:
                    gs = self.cp.add("global_step", 1)
                else:
                    gs = -1
                self.cp.set_json("ps/reply/%d/%d" % (self.owner_index, req["worker"]),
                                 {"version": self.version, "global_step": gs, "staleness": st})
                if self.n:
                    self.p2p.send(self.var[:self.n], wr, _TAG_PULL)
            elif op == "pull":
                self.cp.set_json("ps/reply/%d/%d" % (self.owner_index, req["worker"]),
                                 {"version": self.version, "global_step": -1, "staleness": 0})
                if self.n:
                    self.p2p.send(self.var[:self.n], wr, _TAG_PULL)
            elif op == "save":
                self._save(req)
            elif op == "restore":
                self._restore(req)
            elif op == "stats":
                self.cp.set_json("ps/stats/%d" % self.owner_index,
                                 {"version": self.version, "staleness": {str(k): v for k, v in self.staleness_hist.items()}})
        self.cp.add("ps/stopped", 1)
        if self._owns_group and dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:
                pass

    # sharded checkpoint: this owner writes data shard `owner_index` (var ranges + slots as flat tensors)
    def _save(self, req):
        from ..saver import BundleWriter
        w = BundleWriter(req["prefix"], req["num_shards"], only_shard=self.owner_index)
        for name, b, e, off in req["tensors"][str(self.owner_index)]:
            shape = req["shapes"][name]
            w.add(name, self.var[off:off + (e - b)][:_numel(shape)].reshape(shape), self.owner_index)
            for slot, suffix in req["slots"].items():
                if slot in self.slots:
                    w.add("%s/%s" % (name, suffix), self.slots[slot][off:off + (e - b)][:_numel(shape)].reshape(shape),
                          self.owner_index)
        if self.owner_index == 0:
            w.add("global_step", torch.tensor(int(req["global_step"]), dtype=torch.int64), 0)
            if req["optimizer"] == "Adam":
                w.add("beta1_power", self.beta_powers[0], 0)
                w.add("beta2_power", self.beta_powers[1], 0)
        w.finish()
        self.cp.add("ps/save_done/%s" % req["token"], 1)

    def _restore(self, req):
        from ..saver import BundleReader
        with BundleReader(req["prefix"]) as r:
            for name, b, e, off in req["tensors"][str(self.owner_index)]:
                n = _numel(req["shapes"][name])
                self.var[off:off + n].copy_(r.get_tensor(name).reshape(-1))
                for slot, suffix in req["slots"].items():
                    key = "%s/%s" % (name, suffix)
                    if r.has_tensor(key):
                        self.shard.slot(slot)[off:off + n].copy_(r.get_tensor(key).reshape(-1))
            if r.has_tensor("global_step"):
                self.step_t.fill_(int(r.get_tensor("global_step").item()))
            if r.has_tensor("beta1_power"):
                self.beta_powers[0] = r.get_tensor("beta1_power")
                self.beta_powers[1] = r.get_tensor("beta2_power")
        self.cp.add("ps/restore_done/%s" % req["token"], 1)


def _numel(shape) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n


# --------------------------------------------------------------------------- worker side
class AsyncPSStrategy:
    """Worker-side strategy for ``Sequential.compile(strategy=...)``.

    ``server``: the task's :class:`~distributed_tensorflow_amd.cluster.Server`.  With ps tasks in the
    cluster they are the owners; with ``colocated=True`` (or no ps tasks) every worker also runs an owner
    service thread and the variables are flat-sharded over all workers.
    """

    def __init__(self, server, max_staleness: Optional[int] = None, placement="round_robin",
                 colocated: Optional[bool] = None, timeout_s: float = 300.0):
        self.server = server
        self.max_staleness = max_staleness
        self.worker_index = server.task_index
        self.num_workers = server.num_workers
        self.colocated = (server.num_ps == 0) if colocated is None else colocated
        self.num_owners = self.num_workers if self.colocated else server.num_ps
        self.placement_name = "flat" if self.colocated else placement
        self.timeout_s = timeout_s
        self.world_size = 1           # gradients are NOT averaged across workers in async mode
        self.rank = 0
        self.clock = 0
        self.versions: List[int] = []
        self.staleness_hist: Dict[int, int] = {}
        self.last_global_step = 0
        self._service_thread = None
        self._cps: List[ControlClient] = []
        self._staging: Dict = {}

    def var_alignment(self) -> int:
        return 64 * max(1, self.num_owners)

    # -- topology
    def prepare(self, model):
        from .comm import init_process_group
        from ..cluster import PLACEMENTS
        srv = self.server
        store = model.store
        dev = store.device
        # control plane: one client per owner's server (colocated: the chief's server hosts every queue)
        if self.colocated:
            host, port = srv.control_address(0)
            self._cps = [ControlClient(host, port, self.timeout_s)] * self.num_owners
        else:
            self._cps = [ControlClient(*srv.control_address(j), timeout_s=self.timeout_s) for j in range(self.num_owners)]
        self.cp0 = self._cps[0]
        self.my_rank = self.worker_index if self.colocated else srv.rank
        self.p2p = _P2P(self.my_rank)
        owner_rank = (lambda j: j) if self.colocated else srv.ps_rank
        self.owner_rank = owner_rank
        # layout
        variables = [(v.name, v.numel * 4) for v in store.trainable]
        placement = PLACEMENTS[self.placement_name]().assign(variables, self.num_owners)
        self.placement = placement
        self.ranges = owner_ranges(store, placement, self.num_owners)
        opt = model.optimizer
        opt_desc = {"name": opt.name, "args": ({"lr": opt.lr, "beta_1": opt.beta1, "beta_2": opt.beta2,
                                                "epsilon": opt.epsilon} if opt.name == "Adam"
                                               else {"lr": opt.lr, "momentum": opt.momentum})}
        if self.colocated:
            svc = ParameterServerService(server=srv, owner_index=self.worker_index,
                                         control=ControlClient(*srv.control_address(0), timeout_s=self.timeout_s),
                                         rank=self.my_rank, device=dev, p2p=self.p2p, num_workers=self.num_workers,
                                         worker_rank_of=lambda i: i)
            self._service = svc
            self._service_thread = threading.Thread(target=svc.serve, name="dtf-owner-%d" % self.worker_index,
                                                    daemon=True)
        if srv.is_chief:
            layout = {"ranges": self.ranges, "optimizer": opt_desc, "flat_len": store.flat_len}
            for j in sorted(set(range(self.num_owners))):
                self._cps[j].set_json("ps/layout", layout)
        # data plane (after the layout is out: ps tasks join the process group once they have it)
        if not dist.is_initialized():
            host, port = srv.rendezvous_address()
            from .comm import pick_backend
            backend = pick_backend(dev)
            world = self.num_workers if self.colocated else srv.world_size
            rank = self.worker_index if self.colocated else srv.rank
            init_process_group(backend, rank=rank, world_size=world, master_addr=host, master_port=port,
                               device=dev if backend == "nccl" else None)
        self.p2p.make_direction_groups()
        if self._service_thread is not None:
            self._service_thread.start()
        # chief initialises the owners, everybody else waits (chief-initialises / others-wait)
        if srv.is_chief:
            self._wait_counter("ps/ready", self.num_owners)
            gs0 = int(store.global_step.item())
            for j in range(self.num_owners):
                self._cps[j].push_json("%s/%d" % (_REQ_QUEUE, j), {"op": "init", "worker": self.worker_index,
                                                                  "global_step": gs0})
                if _ranges_numel(self.ranges[j]):
                    self.p2p.send(_pack(store.flat, self.ranges[j], self._stage(j, store)).contiguous(),
                                  owner_rank(j), _TAG_INIT)
            self._wait_counter("ps/init_done", self.num_owners)
            self.cp0.add("global_step", gs0)
            self.cp0.set("ps/initialized", b"1")
        else:
            self.cp0.get("ps/initialized", timeout_s=self.timeout_s)
        self.versions = [0] * self.num_owners
        self.pull(model)
        self.cp0.clock_set("workers", self.worker_index, 0)

    def _stage(self, j: int, store, kind: str = "p") -> torch.Tensor:
        key = (j, kind)
        if key not in self._staging:
            self._staging[key] = torch.empty(max(_ranges_numel(self.ranges[j]), 1), dtype=torch.float32,
                                             device=store.device)
        return self._staging[key]

    def _wait_counter(self, key: str, n: int):
        t0 = time.time()
        per_owner = not self.colocated
        while True:
            if per_owner:
                tot = sum(self._cps[j].add(key, 0) for j in range(self.num_owners))
            else:
                tot = self.cp0.add(key, 0)
            if tot >= n:
                return
            if time.time() - t0 > self.timeout_s:
                raise TimeoutError("waiting for %s (%d/%d)" % (key, tot, n))
            time.sleep(0.01)

    # -- data movement
    def pull(self, model):
        store = model.store
        for j in range(self.num_owners):
            self._cps[j].push_json("%s/%d" % (_REQ_QUEUE, j), {"op": "pull", "worker": self.worker_index})
        for j in range(self.num_owners):
            n = _ranges_numel(self.ranges[j])
            if n:
                buf = self._stage(j, store)[:n] if len(self.ranges[j]) > 1 else store.flat[self.ranges[j][0][0]:self.ranges[j][0][1]]
                self.p2p.recv(buf, self.owner_rank(j), _TAG_PULL)
                _unpack(buf, store.flat, self.ranges[j])
            rep = self._reply(j)
            self.versions[j] = rep["version"]

    def _reply(self, j: int) -> dict:
        key = "ps/reply/%d/%d" % (j, self.worker_index)
        rep = self._cps[j].get_json(key, timeout_s=self.timeout_s)
        self._cps[j].delete(key)
        return rep

    def apply_gradients(self, model):
        """Push this step's gradients to every owner and pull the refreshed parameters back."""
        store = model.store
        if self.max_staleness is not None:
            self.cp0.clock_wait("workers", self.clock, self.max_staleness, timeout_s=self.timeout_s)
        for j in range(self.num_owners):
            self._cps[j].push_json("%s/%d" % (_REQ_QUEUE, j), {"op": "pushpull", "worker": self.worker_index,
                                                              "version": self.versions[j]})
        sends, recvs, bufs = [], [], {}
        for j in range(self.num_owners):
            n = _ranges_numel(self.ranges[j])
            if not n:
                continue
            single = len(self.ranges[j]) == 1
            g = store.flat_grad[self.ranges[j][0][0]:self.ranges[j][0][1]] if single \
                else _pack(store.flat_grad, self.ranges[j], self._stage(j, store, "g"))[:n]
            buf = store.flat[self.ranges[j][0][0]:self.ranges[j][0][1]] if single else self._stage(j, store, "p")[:n]
            bufs[j] = buf
            sends.append((g, self.owner_rank(j), _TAG_PUSH))
            recvs.append((buf, self.owner_rank(j), _TAG_PULL))
        self.p2p.exchange(sends, recvs)
        for j in range(self.num_owners):
            rep = self._reply(j)
            if j in bufs:
                _unpack(bufs[j], store.flat, self.ranges[j])
            self.versions[j] = rep["version"]
            self.staleness_hist[rep["staleness"]] = self.staleness_hist.get(rep["staleness"], 0) + 1
            if rep["global_step"] >= 0:
                self.last_global_step = rep["global_step"]
        if self.last_global_step:
            store.global_step.fill_(self.last_global_step)
        self.clock += 1
        self.cp0.clock_set("workers", self.worker_index, self.clock)

    def global_step(self) -> int:
        """Cluster-wide global step (one control-plane round trip)."""
        return self.cp0.add("global_step", 0)

    def publish_state(self, model, is_chief: bool):
        """Session creation: the chief (re-)initialises the owners from its store - e.g. right after it
        restored a checkpoint - and the other workers wait for that, then everybody pulls."""
        store = model.store
        first = not getattr(self, "_published", False)
        self._published = True
        if is_chief:
            gs = int(store.global_step.item())
            base = self._counter_total("ps/init_done")
            for j in range(self.num_owners):
                self._cps[j].push_json("%s/%d" % (_REQ_QUEUE, j), {"op": "init", "worker": self.worker_index,
                                                                  "global_step": gs})
                n = _ranges_numel(self.ranges[j])
                if n:
                    self.p2p.send(_pack(store.flat, self.ranges[j], self._stage(j, store, "g"))[:n].contiguous(),
                                  self.owner_rank(j), _TAG_INIT)
            self._wait_counter("ps/init_done", base + self.num_owners)
            self.cp0.add("global_step", gs - self.cp0.add("global_step", 0))
        if first:
            self.cp0.barrier("ps/publish", self.num_workers, timeout_s=self.timeout_s)
        self.pull(model)
        store.global_step.fill_(self.global_step())

    def _counter_total(self, key: str) -> int:
        if self.colocated:
            return self.cp0.add(key, 0)
        return sum(self._cps[j].add(key, 0) for j in range(self.num_owners))

    def reduce_stats(self, stats):
        return stats

    # -- checkpoints: each owner writes its own data shard, the chief merges the index
    def _tensor_table(self, store):
        table = {str(j): [] for j in range(self.num_owners)}
        shapes = {}
        if self.placement_name == "flat":
            return None, None
        tr = store.trainable
        for j in range(self.num_owners):
            off = 0
            for b, e in self.ranges[j]:
                for i, v in enumerate(tr):
                    end = tr[i + 1].offset if i + 1 < len(tr) else store.flat_len
                    if v.offset >= b and end <= e:
                        table[str(j)].append((v.name, v.offset, end, off + (v.offset - b)))
                        shapes[v.name] = list(v.shape)
                off += e - b
        return table, shapes

    def save(self, model, prefix: str, token: str):
        table, shapes = self._tensor_table(model.store)
        if table is None:
            raise NotImplementedError("flat-sharded owners checkpoint through the worker-side Saver after a pull")
        slots = getattr(model.optimizer, "slot_checkpoint_names", {})
        gs = self.global_step()
        for j in range(self.num_owners):
            self._cps[j].push_json("%s/%d" % (_REQ_QUEUE, j), {
                "op": "save", "prefix": prefix, "num_shards": self.num_owners, "tensors": table, "shapes": shapes,
                "slots": slots, "global_step": gs, "optimizer": model.optimizer.name, "token": token})
        self._wait_counter("ps/save_done/%s" % token, self.num_owners)
        from ..saver import merge_shards
        merge_shards(prefix, self.num_owners)
        return prefix

    def restore(self, model, prefix: str, token: str):
        table, shapes = self._tensor_table(model.store)
        slots = getattr(model.optimizer, "slot_checkpoint_names", {})
        for j in range(self.num_owners):
            self._cps[j].push_json("%s/%d" % (_REQ_QUEUE, j), {
                "op": "restore", "prefix": prefix, "tensors": table, "shapes": shapes, "slots": slots, "token": token})
        self._wait_counter("ps/restore_done/%s" % token, self.num_owners)

    def owner_stats(self) -> List[dict]:
        out = []
        for j in range(self.num_owners):
            self._cps[j].delete("ps/stats/%d" % j)
            self._cps[j].push_json("%s/%d" % (_REQ_QUEUE, j), {"op": "stats"})
            out.append(self._cps[j].get_json("ps/stats/%d" % j, timeout_s=self.timeout_s))
        return out

    def worker_done(self):
        """Every worker calls this once after its last step."""
        self.cp0.add("workers_done", 1)
        self.cp0.clock_set("workers", self.worker_index, 1 << 60)   # a finished worker never holds others back

    def shutdown_owners(self, wait_for_workers: bool = True):
        """Chief only: ends ``server.join()`` in every ps task (the reference has no such protocol).
        Waits until every worker has reported ``worker_done`` so nobody is cut off mid-step."""
        if wait_for_workers:
            t0 = time.time()
            while self.cp0.add("workers_done", 0) < self.num_workers and time.time() - t0 < self.timeout_s:
                time.sleep(0.02)
        for j in range(self.num_owners):
            self._cps[j].push_json("%s/%d" % (_REQ_QUEUE, j), {"op": "shutdown"})

"""Synchronous-replica data parallelism over the sharded flat variable buffer.

TensorFlow's way to make the reference's training synchronous would be
``SyncReplicasOptimizer`` around the optimizer of /root/reference/example.py:168-170: N
worker gradients are averaged on the parameter servers, the update is applied once, and
``global_step`` advances once per aggregated update.  The MI355X design keeps those
semantics but replaces the PS round trip by collectives on contiguous ranges:

    reduce-scatter(grad bucket)  ->  fused Adam on the 1/N shard this rank owns
                                 ->  all-gather(updated parameter bucket)

Each rank is therefore the "parameter server" (owner) of 1/N of every bucket: the apply work is
sharded N ways, the optimizer slots EXIST only as those shards (a rank allocates 1/N of each slot
buffer, as a reference worker holds no optimizer state at all: /root/reference/example.py:136-139,
168-170 put it on the ps tasks), and every xGMI link carries 1/N of the data in each collective.
A checkpoint gathers each bucket's slot shards to the one replica that writes that bucket.  Buckets are built in backward order and
their reduce-scatter is issued (asynchronously, on RCCL's stream) as soon as the last
gradient of the bucket has been produced, so communication overlaps the remaining
backward GEMMs.  Collectives are large by default (64 MiB buckets): ring collectives over
point-to-point xGMI are per-link bound and launch latency is tens of microseconds, so
fewer, larger messages win; a model smaller than one bucket is a single collective pair.

A model whose whole gradient is a few megabytes (the reference network is 116 KB) is latency-bound
instead: there the strategy runs "replicated" - ONE all-reduce of the flat gradient, then every rank
applies the full update to its own copy - one collective launch per step, nothing to gather, and
forward+backward can replay from a hipGraph because no collective sits inside it.

In a fully synchronous collective there is no stale gradient to drop (TF's
``local_step < global_step`` test): every rank contributes exactly one gradient per step.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from .comm import Communicator


class _Bucket:
    __slots__ = ("begin", "end", "layers", "pending", "work", "shard", "shard_begin", "shard_end", "slot_off", "writer")

    def __init__(self, begin: int, end: int):
        self.begin, self.end = begin, end
        self.layers: List = []
        self.pending = 0
        self.work = None
        self.shard: Optional[torch.Tensor] = None
        self.shard_begin = self.shard_end = 0
        self.slot_off = 0           # where this bucket's shard starts in the rank's slot-shard buffers
        self.writer = 0             # the replica that writes this bucket's variables and slots into checkpoints


class SyncReplicasStrategy:
    def __init__(self, comm: Optional[Communicator] = None, bucket_bytes: int = 64 << 20, overlap: bool = True,
                 average: bool = True, allreduce_below_bytes: int = 8 << 20):
        self.comm = comm if comm is not None else Communicator()
        self.world_size = self.comm.world_size
        self.rank = self.comm.rank
        self.bucket_bytes = int(bucket_bytes)
        self.overlap = overlap
        self.average = average
        # Models whose whole flat gradient is below this size are latency-bound: ONE all-reduce of the flat gradient
        # and a replicated apply (every rank updates every variable) beats a reduce-scatter / all-gather pair with a
        # sharded apply in between - one collective launch per step instead of two, nothing to gather afterwards.
        self.allreduce_below_bytes = int(allreduce_below_bytes)
        self.replicated = False
        self.p2p = None
        self.p2p_check_every = int(os.environ.get("DTF_P2P_CHECK_EVERY", "50"))
        # DTF_TIME_COMM=1: bracket every gradient exchange with stream events (GPU) or a host timer (CPU) and report
        # the total through comm_seconds(); events are resolved lazily, so timing adds no synchronisation
        self.time_comm = os.environ.get("DTF_TIME_COMM", "0") == "1"
        self._comm_total = 0.0
        self._comm_pending: List = []
        self.buckets: List[_Bucket] = []
        self._layer_bucket: Dict[int, _Bucket] = {}
        # exercise the real collectives even with one rank (GPU-box test of the RCCL path)
        self.force = os.environ.get("DTF_FORCE_COLLECTIVES", "0") == "1"
        self.slot_shards: Dict[str, torch.Tensor] = {}     # sharded mode: this rank's 1/N of every optimizer slot
        self.slot_shard_len = 0

    #: ``compile`` leaves the optimizer slots to ``prepare``: full-length on every replica when the apply is
    #: replicated, 1/N per rank when it is sharded
    owns_optimizer_state = True

    # ------------------------------------------------------------------ setup
    def var_alignment(self) -> int:
        """Variable offsets must be multiples of this so every bucket splits evenly into 16-byte-aligned shards."""
        return 64 * self.world_size

    def prepare(self, model):
        store = model.store
        W = self.world_size
        if store.flat_len % (64 * W) != 0:
            raise ValueError("flat buffer (%d) must be padded to 64*world_size; compile() with this strategy "
                             "before build()" % store.flat_len)
        for v in store.trainable:
            if v.offset % (64 * W) != 0:
                raise ValueError("variable %s offset %d not aligned to %d" % (v.name, v.offset, 64 * W))
        # the chief's initial values win; everyone else receives them
        self.comm.broadcast(store.flat, 0)
        self.comm.broadcast(store.global_step, 0)
        self._device = store.device
        self.replicated = store.flat_len * 4 <= self.allreduce_below_bytes
        if self.replicated:
            model.optimizer.create_slots(store)
            for s in store.slots.values():
                self.comm.broadcast(s, 0)
            self.buckets = []
            self._layer_bucket = {}
            model._after_layer_backward = None
            # DTF_ALLREDUCE=p2p: the ranks of one node add up each other's gradient buffers through mapped peer
            # memory in one hop (parallel/p2p_allreduce.py) instead of a ring all-reduce
            self.p2p = None
            if os.environ.get("DTF_ALLREDUCE", "") == "p2p" and self.comm.world_size > 1 and store.device.type == "cuda":
                from .p2p_allreduce import P2PAllReduce
                self.p2p = P2PAllReduce(store.flat_len, store.device, group=self.comm.group)
            return
        self._build_buckets(model)
        self._create_slot_shards(model)
        model._after_layer_backward = self._on_layer_backward if self.overlap else None

    def _create_slot_shards(self, model):
        """The optimizer slots of the shards this rank owns - the only place they exist."""
        from .. import optimizers
        store, opt = model.store, model.optimizer
        self.slot_shards = {}
        for name in opt.slot_order:
            self.slot_shards[name] = torch.full((max(self.slot_shard_len, 1),), float(opt.slot_initial(name)),
                                                dtype=torch.float32, device=store.device)
        if isinstance(opt, optimizers.Adam) and not getattr(store, "_beta_powers_initialised", False):
            store.beta_powers.copy_(optimizers.initial_beta_powers(opt, store.beta_powers.device))
            store._beta_powers_initialised = True

    def slot_bytes_per_rank(self) -> int:
        """Device bytes of optimizer state this rank holds (sharded mode: 1/N of every slot)."""
        return sum(t.numel() * 4 for t in self.slot_shards.values())

    def _build_buckets(self, model):
        store = model.store
        W = self.world_size
        layers_with_vars = [l for l in model.layers if l.variables]
        self.buckets = []
        self._layer_bucket = {}
        cur: Optional[_Bucket] = None
        limit = max(self.bucket_bytes // 4, 64 * W)
        # walk layers in backward order; a bucket is a contiguous flat range [begin, end)
        for layer in reversed(layers_with_vars):
            lo = min(v.offset for v in layer.variables)
            if cur is None:
                cur = _Bucket(lo, store.flat_len if not self.buckets else self.buckets[-1].begin)
                self.buckets.append(cur)
            cur.begin = lo
            cur.layers.append(layer)
            self._layer_bucket[id(layer)] = cur
            if cur.end - cur.begin >= limit:
                cur = None
        if self.buckets:
            self.buckets[-1].begin = 0
        off = 0
        load = [0] * W
        for b in self.buckets:
            n = (b.end - b.begin) // W
            assert n * W == b.end - b.begin
            b.shard_begin = b.begin + self.rank * n
            b.shard_end = b.shard_begin + n
            b.shard = torch.empty(n, dtype=torch.float32, device=store.device)
            b.pending = len(b.layers)
            b.slot_off = off
            off += n
            b.writer = load.index(min(load))        # checkpoint writers: buckets dealt out byte-balanced (greedy)
            load[b.writer] += b.end - b.begin
        self.slot_shard_len = off

    # ------------------------------------------------------------------ per step
    def _launch(self, b: _Bucket, store):
        full = store.flat_grad[b.begin:b.end]
        if self.world_size == 1 and not self.force:
            b.shard.copy_(full)
            b.work = None
        else:
            b.work = self._rs(b.shard, full)

    def _rs(self, shard, full):
        c = self.comm
        if c.world_size == 1:  # forced single-rank collective
            import torch.distributed as dist
            return dist.reduce_scatter_tensor(shard, full, group=c.group, async_op=True) \
                if c.backend == "nccl" else shard.copy_(full)
        return c.reduce_scatter_sum(shard, full, async_op=(c.backend == "nccl"))

    def _on_layer_backward(self, layer, model):
        b = self._layer_bucket.get(id(layer))
        if b is None:
            return
        b.pending -= 1
        if b.pending == 0:
            self._launch(b, model.store)

    # ------------------------------------------------------------------ communication time
    def _comm_begin(self, store):
        if not self.time_comm:
            return None
        if store.device.type == "cuda":
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    def _comm_end(self, store, begin):
        if begin is None:
            return
        if store.device.type == "cuda":
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._comm_pending.append((begin, e))
        else:
            import time
            self._comm_total += time.perf_counter() - begin

    def comm_seconds(self) -> float:
        """Time spent in gradient exchanges so far (only with ``DTF_TIME_COMM=1``); exchanges still in flight are
        counted at a later call."""
        keep = []
        for b, e in self._comm_pending:
            if e.query():
                self._comm_total += b.elapsed_time(e) * 1e-3
            else:
                keep.append((b, e))
        self._comm_pending = keep
        return self._comm_total

    def _apply_replicated(self, model, scale):
        store = model.store
        c = self.comm
        t_comm = self._comm_begin(store)
        if self.world_size > 1 or self.force:
            # a blocking-style call (async_op=False) orders the collective against the compute stream without the
            # event hops to and from RCCL's side stream that an async op plus wait() costs (12 us per step measured
            # on the latency-bound model); the host does not block either way.  DTF_ALLREDUCE_SYNC_OP=0 restores async.
            use_async = c.backend == "nccl" and os.environ.get("DTF_ALLREDUCE_SYNC_OP", "1") == "0"
            if c.world_size == 1:      # forced single-rank collective (GPU-box test of the RCCL call path)
                import torch.distributed as dist
                work = dist.all_reduce(store.flat_grad, group=c.group, async_op=use_async) if c.backend == "nccl" else None
            elif getattr(self, "p2p", None) is not None:
                self.p2p.all_reduce_sum(store.flat_grad)
                work = None
                # a reduce that gave up on a peer has poisoned the gradient with NaN and set an error word; reading
                # it synchronises the device, so it is polled every `p2p_check_every` steps (and before anything is
                # broadcast or saved) - the rank then raises and the launcher restarts the gang
                self._p2p_calls = getattr(self, "_p2p_calls", 0) + 1
                if self._p2p_calls % self.p2p_check_every == 0:
                    self.p2p.check()
            else:
                work = c.all_reduce_sum(store.flat_grad, async_op=use_async)
            if work is not None and hasattr(work, "wait"):
                work.wait()            # stream-level wait under RCCL: the host does not block
        self._comm_end(store, t_comm)
        model.optimizer.apply_and_finish(store, grad_scale=scale)

    def apply_gradients(self, model):
        store = model.store
        scale = 1.0 / self.world_size if self.average else 1.0
        if self.replicated:
            return self._apply_replicated(model, scale)
        gathers = []
        for b in self.buckets:
            if b.pending > 0:  # not launched from the backward hook
                self._launch(b, store)
            b.pending = len(b.layers)
        for b in self.buckets:
            if b.work is not None and hasattr(b.work, "wait"):
                b.work.wait()
            b.work = None
            # owner applies on its shard: var/slots are the flat ranges, the gradient is the reduced shard
            model.optimizer.apply(_ShardView(store, b, self.slot_shards), 0, b.shard_end - b.shard_begin,
                                  grad_scale=scale, grad=b.shard, step=store.global_step)
            full = store.flat[b.begin:b.end]
            mine = store.flat[b.shard_begin:b.shard_end]
            if self.world_size == 1 and not self.force:
                continue
            if self.comm.world_size == 1:
                import torch.distributed as dist
                if self.comm.backend == "nccl":
                    gathers.append(dist.all_gather_into_tensor(full, mine, group=self.comm.group, async_op=True))
                continue
            w = self.comm.all_gather(full, mine, async_op=(self.comm.backend == "nccl"))
            if w is not None:
                gathers.append(w)
        model.optimizer.finish_step(store)
        for w in gathers:
            if hasattr(w, "wait"):
                w.wait()

    def reduce_stats(self, stats: torch.Tensor) -> torch.Tensor:
        if self.world_size == 1:
            return stats
        s = stats.clone()
        self.comm.all_reduce_sum(s)
        return s

    def check_exchange(self):
        """Raise if the peer-pointer gradient exchange has lost a rank (no-op for the RCCL exchange)."""
        if getattr(self, "p2p", None) is not None:
            self.p2p.check()

    def broadcast_state(self, model):
        """Chief-initialises / others-wait: the chief's variables, slots and step win on every replica.  Sharded
        slots are not broadcast: after a restore every replica reads its own shards from the checkpoint files, else
        they are zero everywhere."""
        self.check_exchange()
        store = model.store
        self.comm.broadcast(store.flat, 0)
        self.comm.broadcast(store.global_step, 0)
        self.comm.broadcast(store.beta_powers, 0)
        for v in store.variables.values():           # moving statistics and the like (a restore gave them to the chief)
            if not v.trainable and v.tensor is not None and v.tensor.dtype == torch.float32:
                self.comm.broadcast(v.tensor, 0)
        if self.replicated or not self.buckets:
            for name in sorted(store.slots):
                self.comm.broadcast(store.slots[name], 0)
            return
        prefix = [getattr(model, "_restored_from", None) if self.rank == 0 else None]
        if self.world_size > 1:
            import torch.distributed as dist
            dist.broadcast_object_list(prefix, src=0, group=self.comm.group)
        if prefix[0]:
            self.restore_slot_shards(model, prefix[0])
        else:
            for name, t in self.slot_shards.items():
                t.fill_(float(model.optimizer.slot_initial(name)))

    def _bucket_variables(self, model, b: _Bucket):
        return [v for layer in b.layers for v in layer.variables]

    def restore_slot_shards(self, model, prefix: str):
        """This rank's slices of the optimizer slots, read from the checkpoint ``prefix`` (every replica for itself)."""
        from ..saver import BundleReader
        names = getattr(model.optimizer, "slot_checkpoint_names", {})
        with BundleReader(prefix) as r:
            for b in self.buckets:
                n = b.shard_end - b.shard_begin
                for v in self._bucket_variables(model, b):
                    lo, hi = max(v.offset, b.shard_begin), min(v.offset + v.numel, b.shard_end)
                    if lo >= hi:
                        continue
                    for slot, suffix in names.items():
                        key = "%s/%s" % (v.name, suffix)
                        if slot in self.slot_shards and r.has_tensor(key):
                            dst = self.slot_shards[slot][b.slot_off + lo - b.shard_begin:b.slot_off + hi - b.shard_begin]
                            dst.copy_(r.get_tensor(key).reshape(-1)[lo - v.offset:hi - v.offset])

    def _gather_bucket_slot(self, b: _Bucket, name: str, dst_rank: Optional[int]):
        """The whole slot range ``[b.begin, b.end)`` on ``dst_rank`` (None: on every replica); None elsewhere."""
        import torch.distributed as dist
        n = b.shard_end - b.shard_begin
        mine = self.slot_shards[name][b.slot_off:b.slot_off + n]
        W = self.world_size
        if W == 1:
            return mine
        if dst_rank is None:
            full = torch.empty(n * W, dtype=torch.float32, device=mine.device)
            self.comm.all_gather(full, mine)
            return full
        staged = self.comm.backend == "gloo" and mine.is_cuda
        src = mine.cpu() if staged else mine
        if self.rank == dst_rank:
            full = torch.empty(n * W, dtype=torch.float32, device=src.device)
            dist.gather(src, [full[r * n:(r + 1) * n] for r in range(W)], dst=dst_rank, group=self.comm.group)
            return full
        dist.gather(src, None, dst=dst_rank, group=self.comm.group)
        return None

    def save_sharded(self, model, prefix: str):
        """Collective.  Every replica writes data shard ``rank`` of the checkpoint ``prefix``: the variables of the
        buckets dealt to it, with their optimizer slots, which are gathered - bucket by bucket, to that one writer,
        not to everybody - from the ranks holding the shards.  The chief adds the step and merges the index."""
        from ..saver import BundleWriter, merge_shards
        self.check_exchange()
        store, opt = model.store, model.optimizer
        names = getattr(opt, "slot_checkpoint_names", {})
        w = BundleWriter(prefix, self.world_size, only_shard=self.rank)
        try:
            for b in self.buckets:
                mine = b.writer == self.rank
                if mine:
                    for v in self._bucket_variables(model, b):
                        w.add(v.name, v.tensor, self.rank)
                for slot in sorted(self.slot_shards):
                    full = self._gather_bucket_slot(b, slot, b.writer)
                    if mine:
                        for v in self._bucket_variables(model, b):
                            t = full[v.offset - b.begin:v.offset - b.begin + v.numel].view(v.shape)
                            w.add("%s/%s" % (v.name, names.get(slot, slot)), t, self.rank)
                    del full
            if self.rank == 0:
                for v in store.variables.values():
                    if not v.trainable:
                        w.add(v.name, v.tensor, 0)
                w.add("global_step", store.global_step.view(()), 0)
                if getattr(opt, "name", "") == "Adam":
                    w.add("beta1_power", store.beta_powers[0], 0)
                    w.add("beta2_power", store.beta_powers[1], 0)
            w.finish()
        except Exception:
            w.abort()
            raise
        self.comm.barrier()
        if self.rank == 0:
            merge_shards(prefix, self.world_size)
        self.comm.barrier()
        return prefix

    def full_slots(self, model) -> Dict[str, torch.Tensor]:
        """Collective: complete optimizer-slot buffers in the flat layout, on every replica (tests, inspection; a
        training run never needs them)."""
        store = model.store
        if self.replicated or not self.buckets:
            return dict(store.slots)
        out = {}
        for name in sorted(self.slot_shards):
            full = torch.zeros(store.flat_len, dtype=torch.float32, device=store.device)
            for b in self.buckets:
                full[b.begin:b.end].copy_(self._gather_bucket_slot(b, name, None))
            out[name] = full
        return out

    def gather_slots(self, model):
        """Collective no-op kept for callers that checkpoint through the chief: a replicated apply keeps every slot
        complete on every replica; sharded slots are written by ``save_sharded``."""
        self.check_exchange()                             # never checkpoint past a failed gradient exchange

    def agree(self, flag: bool, step: Optional[int] = None) -> bool:
        """The chief's boolean decision, on every replica (one tiny all-reduce).

        With ``step`` (the caller's ``global_step``) the same exchange checks that the replicas are in step.
        ``tf.train.SyncReplicasOptimizer`` drops a gradient whose ``local_step`` is behind ``global_step``; here a
        gradient can only be stale if a replica's counter has left the group's (replicas apply one aggregated update
        per collective, in lock step), and that replica's variables are then suspect too - so every replica raises
        ``ClusterLostError`` and the launcher restarts the gang from the last checkpoint, which drops the stale work."""
        if self.world_size == 1:
            return bool(flag)
        s = int(step) if step is not None else 0
        t = torch.tensor([1 if (flag and self.rank == 0) else 0, s, -s], dtype=torch.int64,
                         device=getattr(self, "_device", "cpu"))
        self.comm.all_reduce_max(t)
        decision, hi, lo = (int(v) for v in t.tolist())
        if step is not None and hi != -lo:
            from ..failure import ClusterLostError
            raise ClusterLostError("synchronous replicas are out of step: global_step ranges from %d to %d (this replica, "
                                   "rank %d: %d)" % (-lo, hi, self.rank, s))
        return bool(decision)

    collective_checkpoint = True


class _ShardView:
    """Presents one bucket shard of a VariableStore as a tiny store so ``Optimizer.apply`` runs on it unchanged."""

    def __init__(self, store, bucket: _Bucket, slot_shards):
        self._store = store
        self._b = bucket
        self._slots = slot_shards
        self.flat = store.flat[bucket.shard_begin:bucket.shard_end]
        self.flat_grad = None
        self.flat_len = bucket.shard_end - bucket.shard_begin
        self.global_step = store.global_step
        self.beta_powers = store.beta_powers

    def slot(self, name: str, initial: float = 0.0) -> torch.Tensor:
        b = self._b
        return self._slots[name][b.slot_off:b.slot_off + b.shard_end - b.shard_begin]

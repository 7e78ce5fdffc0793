"""Train-engine adapter: what :class:`~distributed_tensorflow_amd.session.MonitoredSession` drives.

It wraps a compiled :class:`~distributed_tensorflow_amd.models.Sequential` (variables,
optimizer, parallel strategy) and resolves *fetches*:

* ``"train_op"``            - one optimisation step on the fed batch;
* ``"loss"`` / ``"accuracy"`` - mean over the batch of this run (one device->host read);
* ``"global_step"``         - the (cluster-wide) step counter;
* a summary op (``summary.scalar`` / ``merge_all``) - evaluated into a ``Summary``;
* any callable              - called with no arguments.

The graph-style front end (:mod:`distributed_tensorflow_amd.compat`) subclasses this and maps its
symbolic nodes onto the same fetch names.
"""
from __future__ import annotations

import os

import json
import uuid
from typing import Any, Dict, List, Optional

import torch

from . import saver as saver_mod
from . import summary as summary_mod

TRAIN_OP, LOSS, ACCURACY, GLOBAL_STEP = "train_op", "loss", "accuracy", "global_step"


class FeedStager:
    """Host arrays -> device tensors through pinned staging buffers.

    One ring per (shape, dtype): ``depth`` pinned host buffers and as many device buffers.  ``put`` converts the array
    into the next pinned buffer (one pass on the CPU: dtype conversion included), enqueues one non-blocking copy on the
    current stream and records an event; a staging buffer is only rewritten after its event has completed, so a loop
    that never reads anything back cannot overwrite a copy still in flight.  The device tensor handed out is reused
    ``depth`` feeds of the same shape later - it is meant for values consumed within the step they are fed for.
    Arrays above ``max_bytes`` are not staged (``put`` returns None): they are one-off transfers of whole data sets.
    """

    def __init__(self, device, depth: int = 4, max_bytes: int = 4 << 20):
        self.device = device
        self.depth = depth
        self.max_bytes = max_bytes
        self._rings = {}

    @staticmethod
    def _target_dtype(t: torch.Tensor):
        # the rule of models.sequential._as_tensor: integer vectors are class ids, everything else is fp32
        return torch.int64 if t.dtype in (torch.int64, torch.int32, torch.int16, torch.uint8) and t.dim() == 1 \
            else torch.float32

    def put(self, a):
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(a)
        if t.device.type != "cpu":
            return None
        dtype = self._target_dtype(t)
        nbytes = t.numel() * (8 if dtype == torch.int64 else 4)
        if nbytes == 0 or nbytes > self.max_bytes:
            return None
        key = (tuple(t.shape), dtype)
        ring = self._rings.get(key)
        if ring is None:
            ring = self._rings[key] = _FeedRing(t.shape, dtype, nbytes, self.depth, self.device)
        return ring.push(t)

    def close(self):
        for ring in self._rings.values():
            ring.close()
        self._rings = {}


class _FeedRing:
    """One shape's staging ring; wait, host copy, asynchronous device copy and event record are one native call
    (``dtf_feed_ring_push``)."""

    def __init__(self, shape, dtype, nbytes, depth, device):
        import ctypes as C
        from .ops import _native as nv
        self._nv, self._C = nv, C
        self.dtype = dtype
        self.index = device.index if device.index is not None else torch.cuda.current_device()
        self.host = [torch.empty(shape, dtype=dtype).pin_memory() for _ in range(depth)]
        self.dev = [torch.empty(shape, dtype=dtype, device=device) for _ in range(depth)]
        lib = nv.kernels()
        self._push = nv.declare(lib, "dtf_feed_ring_push", [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)])
        self._acquire = nv.declare(lib, "dtf_feed_ring_acquire", [C.c_void_p, C.POINTER(C.c_int)])
        self._destroy = nv.declare(lib, "dtf_feed_ring_destroy", [C.c_void_p])
        create = nv.declare(lib, "dtf_feed_ring_create",
                            [C.c_int, C.c_longlong, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)])
        hp = (C.c_void_p * depth)(*[h.data_ptr() for h in self.host])
        dp = (C.c_void_p * depth)(*[d.data_ptr() for d in self.dev])
        handle = C.c_void_p()
        nv.check(create(depth, nbytes, hp, dp, C.byref(handle)), "dtf_feed_ring_create")
        self.handle = handle
        self._slot = C.c_int(0)

    def push(self, t: torch.Tensor) -> torch.Tensor:
        stream = torch._C._cuda_getCurrentRawStream(self.index)
        if t.dtype == self.dtype and t.is_contiguous():
            src = t.data_ptr()
        else:
            # dtype conversion (float64 / integer arrays): written into the pinned buffer by the tensor library
            self._nv.check(self._acquire(self.handle, self._slot), "dtf_feed_ring_acquire")
            self.host[self._slot.value].copy_(t)
            src = None
        self._nv.check(self._push(self.handle, src, stream, self._slot), "dtf_feed_ring_push")
        return self.dev[self._slot.value]

    def close(self):
        if self.handle is not None:
            self._destroy(self.handle)
            self.handle = None


class ModelEngine:
    # DTF_GRAPH_SMALL_STEPS=1: batches of up to 1024 rows replay forward + loss + backward from a hipGraph.  Off by
    # default: with several tasks time-sharing one GPU a graph launch costs more than the dozen kernel launches it
    # replaces (profiles/r3_async_ps_task_planes_b50.txt); one process per GPU may gain from it.
    _graph_small_steps = os.environ.get("DTF_GRAPH_SMALL_STEPS", "0") == "1"

    def __init__(self, model, x_key="x", y_key="y", training_key="training"):
        self.model = model
        self.x_key, self.y_key, self.training_key = x_key, y_key, training_key
        self._last_logs: Dict[str, float] = {}

    # ---- structure
    @property
    def store(self):
        return self.model.store

    @property
    def optimizer(self):
        return self.model.optimizer

    @property
    def strategy(self):
        return self.model.strategy

    def ensure_built(self):
        if not self.model.compiled:
            raise RuntimeError("compile() the model before opening a session")

    def describe(self) -> str:
        return json.dumps(self.model.get_config(), indent=1, default=str)

    def variable_device_fn(self, worker_device: str = ""):
        """name -> device string of a variable under the model's strategy (``/job:ps/task:k`` for whole variables
        owned by ps task k; the worker's own device otherwise)."""
        placement = getattr(self.model.strategy, "placement", None) or {}
        colocated = bool(getattr(self.model.strategy, "colocated", False))

        def dev(name: str) -> str:
            k = placement.get(name, placement.get(name.rsplit("/", 1)[0]))
            if k is None or k < 0 or colocated:
                return worker_device
            return "/job:ps/task:%d" % k
        return dev

    def graph_def_nodes(self):
        """GraphDef nodes of the compiled model (for ``FileWriter.add_graph`` and ``graph.pbtxt``)."""
        from . import graph_def as gd
        from .cluster import current_server
        srv = current_server()
        wdev = "/job:worker/task:%d" % srv.task_index if srv is not None and srv.job_name == "worker" else ""
        return gd.model_nodes(self.model, worker_device=wdev, variable_device=self.variable_device_fn(wdev),
                              loss_name=getattr(self.model.loss, "name", None),
                              optimizer_name=getattr(self.model.optimizer, "name", None))

    # ---- global step
    def global_step_value(self) -> int:
        """The (cluster-wide) step counter without a device synchronisation: single-replica and synchronous-replica
        training advance the counter exactly once per local step, so the model's host mirror is exact; asynchronous
        workers take the value that came back with their last push (see ``AsyncPSStrategy.cached_global_step``)."""
        st = self.model.strategy
        cached = getattr(st, "cached_global_step", None)
        if callable(cached):
            return int(cached())
        gs = getattr(st, "global_step", None)
        if callable(gs):
            return int(gs())
        return int(self.model._host_step)

    def device_global_step(self) -> int:
        """The device-resident counter itself (one device->host read); what checkpoints record."""
        return int(self.model.store.global_step.item())

    # ---- checkpoints
    def _sharded_sync_save(self) -> bool:
        """Synchronous replicas of a model large enough to be sharded: every replica writes its share of the files."""
        st = self.model.strategy
        return bool(getattr(st, "collective_checkpoint", False)) and not getattr(st, "replicated", True)

    def make_saver(self, max_to_keep: int = 5):
        st = self.model.strategy
        store = self.model.store
        if self._sharded_sync_save():
            # one data shard per replica: the strategy deals its buckets (whole variables with their slots) out
            return saver_mod.Saver(store, self.model.optimizer, max_to_keep=max_to_keep, num_shards=st.world_size)
        placement = getattr(st, "placement", None) or {}
        shards = getattr(st, "num_owners", 1) if placement and not all(v == -1 for v in placement.values()) else 1
        return saver_mod.Saver(store, self.model.optimizer, max_to_keep=max_to_keep,
                               placement=placement if shards > 1 else None, num_shards=shards)

    def save_checkpoint(self, saver, save_path: str, step: int) -> str:
        """One data shard per variable owner, index merged by the chief - whatever the strategy
        (/root/reference/example.py:191: TF's sharded saver writes one shard per parameter-server task)."""
        st = self.model.strategy
        prefix = "%s-%d" % (save_path, int(step))
        if hasattr(st, "save"):
            # asynchronous parameter servers: every owner writes its own shard (variables and slots), the chief merges
            st.save(self.model, prefix, uuid.uuid4().hex)
            saver._record(prefix)
            return prefix
        if self._sharded_sync_save():
            # every replica holds all variables (all-gathered each step) but only its shards of the optimizer slots:
            # each bucket's slot shards are gathered to the one replica that writes that bucket, and the device-to-host
            # copies and file writes run in parallel on all replicas instead of funnelling through the chief
            st.save_sharded(self.model, prefix)
            if st.rank == 0:
                saver._record(prefix)
            return prefix
        if hasattr(st, "gather_slots"):
            st.gather_slots(self.model)          # collective: every replica takes part
            if st.rank != 0:
                return prefix
        return saver.save(save_path, global_step=step)

    def restore_checkpoint(self, saver, prefix: str) -> int:
        step = saver.restore(prefix)
        self.model._host_step = step
        self.model._restored_from = prefix      # owners of an asynchronous cluster reload their shards from it too
        return step

    def synchronize_state(self, is_chief: bool):
        st = self.model.strategy
        if hasattr(st, "broadcast_state"):
            st.broadcast_state(self.model)
        elif hasattr(st, "publish_state"):
            st.publish_state(self.model, is_chief)
        # session creation is the one place where the host mirror is re-read from the device (a replica just
        # received the chief's counter; a restore just set it)
        self.model._host_step = int(self.model.store.global_step.item())

    # ---- running
    def _feed(self, feed_dict, key, default=None):
        if feed_dict is None:
            return default
        return feed_dict.get(key, default)

    def _to_device(self, a, labels=False):
        """A fed value on the model's device.  Host arrays of a session-style loop (``sess.run(train_op, {x: batch})``
        at /root/reference/example.py:207-213) arrive once per step: a pageable host-to-device copy costs ~40 us here
        (it synchronises the stream), so small feeds go through a ring of pinned staging buffers and one asynchronous
        copy each (`FeedStager`); the tensors are consumed inside the same `run`."""
        dev = self.model.device
        if isinstance(a, (tuple, list)) and a and all(isinstance(t, torch.Tensor) for t in a):
            return tuple(self._to_device(t) for t in a)      # a model with several fed placeholders
        if dev.type == "cuda" and not (isinstance(a, torch.Tensor) and a.is_cuda):
            stager = getattr(self, "_stager", None)
            if stager is None:
                stager = self._stager = FeedStager(dev)
            t = stager.put(a)
            if t is not None:
                return t
        from .models.sequential import _as_tensor
        return _as_tensor(a, dev)

    def resolve(self, fetch):
        """Map a caller-level fetch object to an engine fetch name / callable (overridden by the graph engine)."""
        return fetch

    def run(self, fetch_struct: Dict[str, Any], feed_dict: Optional[Dict] = None) -> Dict[str, Any]:
        flat: List[Any] = []

        def walk(f):
            if isinstance(f, (list, tuple)):
                for g in f:
                    walk(g)
            elif isinstance(f, dict):
                for g in f.values():
                    walk(g)
            elif f is not None:
                flat.append(self.resolve(f))

        walk(fetch_struct)
        names = [f for f in flat if isinstance(f, str)]
        x = self._feed(feed_dict, self.x_key)
        y = self._feed(feed_dict, self.y_key)
        training = bool(self._feed(feed_dict, self.training_key, TRAIN_OP in names))
        need_stats = LOSS in names or ACCURACY in names or any(
            isinstance(f, (summary_mod.MergedSummaryOp, summary_mod.ScalarSummaryOp)) for f in flat)
        m = self.model
        if TRAIN_OP in names:
            if x is None or y is None:
                raise ValueError("train_op needs both inputs and labels in feed_dict")
            xd, yd = self._to_device(x), self._to_device(y)
            if need_stats:
                m.stats.zero_()
            if training:
                plan = m.fused_step_plan(int(xd.shape[0])) if hasattr(m, "fused_step_plan") and not isinstance(xd, tuple) \
                    else None
                if plan is not None:
                    m.train_steps_fused(xd, yd, int(xd.shape[0]), 1, plan)     # the whole step in one launch
                elif self._graph_small_steps and not isinstance(xd, tuple) and xd.is_cuda and xd.shape[0] <= 1024 \
                        and hasattr(m, "train_step_graphed"):
                    m.train_step_graphed(xd, yd)      # cluster strategies at launch-bound batch sizes
                else:
                    m.train_step(xd, yd)
            else:
                # learning phase 0 with a training op: update without dropout
                out = m.forward(xd, training=False)
                g = m._loss_and_metrics(out, yd, True, m.stats)
                m.backward(g)
                m.strategy.apply_gradients(m)
                m._host_step += 1
            if need_stats:
                self._last_logs = m.read_stats()
        elif need_stats:
            if x is None or y is None:
                raise ValueError("evaluating loss/accuracy needs inputs and labels in feed_dict")
            xd, yd = self._to_device(x), self._to_device(y)
            from .ops import functional as F
            st = F.new_stats(m.device)
            out = m.forward(xd, training=training)
            m._loss_and_metrics(out, yd, False, st)
            self._last_logs = m.read_stats(stats=st)

        def value(f):
            f = self.resolve(f)
            if isinstance(f, str):
                if f == TRAIN_OP:
                    return None
                if f == LOSS:
                    return self._last_logs["loss"]
                if f == ACCURACY:
                    return self._last_logs.get("binary_rounded_accuracy", self._last_logs.get("acc"))
                if f == GLOBAL_STEP:
                    return self.global_step_value()
                if f == "predictions":
                    return m.predict(x, batch_size=len(x))
                raise KeyError("unknown fetch %r" % f)
            if isinstance(f, summary_mod.ScalarSummaryOp):
                v = value(f.source)
                if hasattr(v, "item"):
                    v = v.item()
                return summary_mod.Summary([(f.name, float(v))])
            if isinstance(f, summary_mod.MergedSummaryOp):
                merged = summary_mod.Summary()
                for op in f.ops:
                    merged.value.extend(value(op).value)
                return merged
            if callable(f):
                return f()
            return f

        def rebuild(f):
            if isinstance(f, list):
                return [rebuild(g) for g in f]
            if isinstance(f, tuple):
                return tuple(rebuild(g) for g in f)
            if isinstance(f, dict):
                return {k: rebuild(g) for k, g in f.items()}
            if f is None:
                return None
            return value(f)

        return rebuild(fetch_struct)

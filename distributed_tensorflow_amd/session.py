"""``MonitoredTrainingSession`` and the session-run hook protocol.

What one call gives the reference (/root/reference/example.py:187-192): the chief
initialises or restores the variables and everybody else waits for it; a checkpoint is
written at the first step, every ``save_checkpoint_secs`` (600) and at the end; summaries
every ``save_summaries_steps`` (100); ``global_step/sec`` every 100 steps; hooks run around
every ``run`` call; ``should_stop()`` turns true when a hook asks for it
(``StopAtStepHook(last_step=...)`` counts *absolute* global steps, so a restarted job
continues instead of starting over); recoverable errors make the session re-create itself
from the last checkpoint and retry.

The session drives a *train engine*: any object with ``store`` (VariableStore),
``optimizer``, ``strategy``, ``run(fetches, feed_dict)`` and ``global_step_value()`` - the
graph-style front end in :mod:`distributed_tensorflow_amd.compat` and
:class:`~distributed_tensorflow_amd.models.Sequential` both provide it.
"""
from __future__ import annotations

import os
import time
import uuid
from typing import Any, Dict, List, Optional, Sequence

from . import saver as saver_mod
from . import summary as summary_mod


# --------------------------------------------------------------------------- errors
class AbortedError(RuntimeError):
    """A peer task or the transport went away mid-step; the session may recover from a checkpoint."""


class UnavailableError(RuntimeError):
    """A peer task cannot be reached (yet); recoverable."""


class OutOfRangeError(RuntimeError):
    """Input exhausted: ends the monitored loop cleanly."""


RECOVERABLE_ERRORS = (AbortedError, UnavailableError)


def _transient_errors():
    """Real faults that are worth one in-process recovery: a control-plane request that timed out (an owner was slow
    to answer, a connection hiccup).  They are treated like TF's ``UnavailableError``: the session re-creates itself
    from the last checkpoint and retries; if the peer is really gone the recovery itself fails - or the failure
    detector reports ``ClusterLostError``, which is never retried in process (a static process group cannot re-admit
    a member: the launcher restarts the gang)."""
    from .control import ControlTimeout
    return (ControlTimeout,)


# --------------------------------------------------------------------------- hook protocol
class SessionRunArgs:
    def __init__(self, fetches=None, feed_dict=None):
        self.fetches = fetches
        self.feed_dict = feed_dict


class SessionRunValues:
    def __init__(self, results=None):
        self.results = results


class SessionRunContext:
    def __init__(self, original_args: SessionRunArgs, session: "MonitoredSession"):
        self.original_args = original_args
        self.session = session
        self._stop_requested = False

    @property
    def stop_requested(self) -> bool:
        return self._stop_requested

    def request_stop(self):
        self._stop_requested = True


class SessionRunHook:
    def begin(self): pass
    def after_create_session(self, session, coord): pass
    def before_run(self, run_context: SessionRunContext) -> Optional[SessionRunArgs]: return None
    def after_run(self, run_context: SessionRunContext, run_values: SessionRunValues): pass
    def end(self, session): pass


class StopAtStepHook(SessionRunHook):
    """Request a stop once ``global_step >= last_step`` (or after ``num_steps`` more steps)."""

    def __init__(self, num_steps: Optional[int] = None, last_step: Optional[float] = None):
        if (num_steps is None) == (last_step is None):
            raise ValueError("exactly one of num_steps and last_step must be given")
        self._num_steps, self._last_step = num_steps, last_step

    def after_create_session(self, session, coord):
        if self._last_step is None:
            self._last_step = session.global_step_value() + self._num_steps
        if session.global_step_value() >= self._last_step:
            session.request_stop()

    def after_run(self, run_context, run_values):
        if run_context.session.global_step_value() >= self._last_step:
            run_context.request_stop()


class StepCounterHook(SessionRunHook):
    """``global_step/sec`` (and examples/sec when the batch size is known) every N steps."""

    def __init__(self, every_n_steps: int = 100, summary_writer=None, output_dir: Optional[str] = None,
                 batch_size: Optional[int] = None, log=print):
        self.every, self.writer, self.output_dir, self.batch_size, self.log = every_n_steps, summary_writer, output_dir, batch_size, log
        self._last_step = None
        self._last_time = None
        self.last_rate = None

    def after_create_session(self, session, coord):
        if self.writer is None and self.output_dir:
            self.writer = summary_mod.FileWriterCache.get(self.output_dir)
        self._last_step, self._last_time = session.global_step_value(), time.time()

    def after_run(self, run_context, run_values):
        step = run_context.session.global_step_value()
        if step - self._last_step >= self.every:
            now = time.time()
            rate = (step - self._last_step) / max(now - self._last_time, 1e-9)
            self.last_rate = rate
            if self.writer is not None:
                vals = [("global_step/sec", rate)]
                if self.batch_size:
                    vals.append(("examples/sec", rate * self.batch_size))
                self.writer.add_summary(summary_mod.Summary(vals), step)
            if self.log:
                self.log("INFO: global_step/sec: %g" % rate)
            self._last_step, self._last_time = step, now

    def end(self, session):
        if self.writer is not None:
            self.writer.flush()


class StepLatencyHook(SessionRunHook):
    """Step-latency percentiles and communication time, every N steps, as scalars next to ``global_step/sec``:
    ``step_latency_ms/p50|p90|p99|max`` (host wall time of one ``run``: exact when the run fetches something from the
    device - the reference fetches loss and accuracy every step - and the enqueue time of the step otherwise) and, when
    the strategy keeps the counters, ``comm_ms/step`` (time in the gradient exchange per step) and
    ``staleness_wait_ms/step`` (time blocked by the staleness bound)."""

    def __init__(self, every_n_steps: int = 100, summary_writer=None, output_dir: Optional[str] = None, log=None):
        self.every, self.writer, self.output_dir, self.log = every_n_steps, summary_writer, output_dir, log
        self._lat: List[float] = []
        self._t0 = None
        self._comm0 = self._wait0 = 0.0
        self.last: Dict[str, float] = {}

    def after_create_session(self, session, coord):
        if self.writer is None and self.output_dir:
            self.writer = summary_mod.FileWriterCache.get(self.output_dir)

    def before_run(self, run_context):
        self._t0 = time.perf_counter()
        return None

    @staticmethod
    def percentiles(samples: Sequence[float]) -> Dict[str, float]:
        xs = sorted(samples)
        pick = lambda q: xs[min(len(xs) - 1, int(q * len(xs)))]
        return {"p50": pick(0.50), "p90": pick(0.90), "p99": pick(0.99), "max": xs[-1]}

    def after_run(self, run_context, run_values):
        self._lat.append((time.perf_counter() - self._t0) * 1e3)
        if len(self._lat) < self.every:
            return
        sess = run_context.session
        vals = [("step_latency_ms/" + k, v) for k, v in self.percentiles(self._lat).items()]
        st = getattr(sess.engine, "strategy", None)
        for name, attr, prev in (("comm_ms/step", "comm_seconds", "_comm0"), ("staleness_wait_ms/step", "wait_seconds", "_wait0")):
            tot = getattr(st, attr, None)
            if tot is not None:
                tot = float(tot() if callable(tot) else tot)
                vals.append((name, (tot - getattr(self, prev)) * 1e3 / len(self._lat)))
                setattr(self, prev, tot)
        self.last = dict(vals)
        if self.writer is not None:
            self.writer.add_summary(summary_mod.Summary(vals), sess.global_step_value())
        if self.log:
            self.log("INFO: " + ", ".join("%s = %.3f" % kv for kv in vals))
        self._lat = []

    def end(self, session):
        if self.writer is not None:
            self.writer.flush()


class SummarySaverHook(SessionRunHook):
    """Evaluate ``summary_op`` (default: ``merge_all()``) every N steps and write it at the true global step."""

    def __init__(self, save_steps: int = 100, output_dir: Optional[str] = None, summary_writer=None, summary_op=None):
        self.save_steps, self.output_dir, self.writer, self.summary_op = save_steps, output_dir, summary_writer, summary_op
        self._next = None
        self._own_writer = False

    def begin(self):
        if self.summary_op is None:
            self.summary_op = summary_mod.merge_all()

    def after_create_session(self, session, coord):
        if self.writer is None and self.output_dir:
            self.writer = summary_mod.FileWriterCache.get(self.output_dir)
        self._next = session.global_step_value()

    def before_run(self, run_context):
        self._pre_step = run_context.session.global_step_value()
        self._want = self.summary_op is not None and self._pre_step >= self._next
        return SessionRunArgs(fetches=self.summary_op) if self._want else None

    def after_run(self, run_context, run_values):
        if self._want and self.writer is not None and run_values.results is not None:
            step = run_context.session.global_step_value()
            self.writer.add_summary(run_values.results, step)
            self._next = self._pre_step + self.save_steps

    def end(self, session):
        if self.writer is not None:
            self.writer.flush()


class CheckpointSaverHook(SessionRunHook):
    """Chief-only: save at the first step, every ``save_secs`` / ``save_steps``, and at ``end``."""

    def __init__(self, checkpoint_dir: str, save_secs: Optional[float] = 600.0, save_steps: Optional[int] = None,
                 saver=None, checkpoint_basename: str = "model.ckpt"):
        self.dir, self.save_secs, self.save_steps, self.saver = checkpoint_dir, save_secs, save_steps, saver
        self.basename = checkpoint_basename
        self.poll_steps = 100
        self._last_time = None
        self._last_step = None
        self.saved_steps: List[int] = []

    def after_create_session(self, session, coord):
        os.makedirs(self.dir, exist_ok=True)
        # like TF's CheckpointSaverHook: the GraphDef in text format next to the checkpoints
        try:
            from . import graph_def as gd
            with open(os.path.join(self.dir, "graph.pbtxt"), "w") as f:
                f.write(gd.graph_def_text(gd.nodes_for(session.graph)))
        except OSError:
            pass

    def _save(self, session, step: int):
        prefix = session.save_checkpoint(os.path.join(self.dir, self.basename), step, saver=self.saver)
        self._last_time, self._last_step = time.time(), step
        self.saved_steps.append(step)
        if session.is_chief and prefix:
            self._write_meta(session, prefix)

    def _write_meta(self, session, prefix: str):
        """``<prefix>.meta``: the MetaGraphDef TF's saver writes with every checkpoint (graph, SaverDef, the variable
        collections).  Encoded once per session; the saver's clean-up removes it with the checkpoint it belongs to."""
        try:
            if getattr(self, "_meta", None) is None:
                from . import graph_def as gd
                store = session.engine.store
                variables = [(v.name, bool(v.trainable)) for v in store.variables.values()] + [("global_step", False)]
                nodes = gd.nodes_for(session.graph)
                train_ops = [n.name for n in nodes if n.op == "AssignAdd" and "global_step" in n.inputs]
                saver = session._saver(self.saver)
                self._meta = gd.encode_meta_graph_def(nodes, variables, train_ops,
                                                      max_to_keep=getattr(saver, "max_to_keep", 5) or 0,
                                                      sharded=getattr(saver, "num_shards", 1) > 1)
            with open(prefix + ".meta", "wb") as f:
                f.write(self._meta)
        except (OSError, AttributeError, KeyError, TypeError):
            pass        # the file is not needed to restore; a model the GraphDef cannot be derived for goes without

    def after_run(self, run_context, run_values):
        s = run_context.session
        step = s.global_step_value()
        if self._last_step is None:
            if step > 0:
                self._save(s, step)
            return
        due = (self.save_secs is not None and time.time() - self._last_time >= self.save_secs) or \
              (self.save_steps is not None and step - self._last_step >= self.save_steps)
        agree = getattr(s.engine.strategy, "agree", None)
        if agree is not None and getattr(s.engine.strategy, "collective_checkpoint", False):
            # synchronous replicas save collectively: every replica runs this hook and follows the chief's
            # clock, consulted every `poll_steps` steps so that the broadcast stays off the per-step path
            if step % self.poll_steps != 0:
                return
            due = agree(due, step)
        if due and step != self._last_step:
            self._save(s, step)

    def end(self, session):
        step = session.global_step_value()
        if step != self._last_step:
            self._save(session, step)


class LoggingTensorHook(SessionRunHook):
    def __init__(self, tensors: Dict[str, Any], every_n_iter: int = 100, log=print):
        self.tensors, self.every, self.log = tensors, every_n_iter, log
        self._iter = 0

    def before_run(self, run_context):
        self._want = self._iter % self.every == 0
        return SessionRunArgs(fetches=list(self.tensors.values())) if self._want else None

    def after_run(self, run_context, run_values):
        if self._want and run_values.results is not None:
            self.log("INFO: " + ", ".join("%s = %s" % (k, v) for k, v in zip(self.tensors, run_values.results)))
        self._iter += 1


class NanLossHook(SessionRunHook):
    """Stop (or raise) when the watched loss becomes NaN."""

    def __init__(self, loss_tensor, fail_on_nan_loss: bool = True):
        self.loss, self.fail = loss_tensor, fail_on_nan_loss

    def before_run(self, run_context):
        return SessionRunArgs(fetches=self.loss)

    def after_run(self, run_context, run_values):
        v = run_values.results
        if v is not None and v != v:
            if self.fail:
                raise FloatingPointError("NaN loss during training")
            run_context.request_stop()


class FaultInjectionHook(SessionRunHook):
    """Test hook: raise a recoverable (or fatal) error when ``global_step`` reaches ``at_step``, once."""

    def __init__(self, at_step: int, error: type = AbortedError, kill_process: bool = False):
        self.at_step, self.error, self.kill = at_step, error, kill_process
        self.fired = False

    def before_run(self, run_context):
        if not self.fired and run_context.session.global_step_value() >= self.at_step:
            self.fired = True
            if self.kill:
                os._exit(17)
            raise self.error("injected fault at global_step %d" % self.at_step)
        return None


# --------------------------------------------------------------------------- scaffold
class Scaffold:
    """Holds the pieces a session needs once the model is final (saver, summary op, init callbacks)."""

    def __init__(self, saver=None, summary_op=None, init_fn=None):
        self.saver = saver
        self.summary_op = summary_op
        self.init_fn = init_fn


# --------------------------------------------------------------------------- sessions
class MonitoredSession:
    def __init__(self, engine, is_chief: bool = True, master: str = "", checkpoint_dir: Optional[str] = None,
                 hooks: Optional[Sequence[SessionRunHook]] = None, scaffold: Optional[Scaffold] = None,
                 max_recoveries: int = 8, recovery_wait_secs: float = 1.0, stop_grace_period_secs: float = 120.0):
        self.engine = engine
        self.is_chief = is_chief
        self.master = master
        self.checkpoint_dir = checkpoint_dir
        self.hooks: List[SessionRunHook] = list(hooks or [])
        self.scaffold = scaffold or Scaffold()
        self.max_recoveries = max_recoveries
        self.recovery_wait_secs = recovery_wait_secs
        self._stop = False
        self._closed = False
        self._recoveries = 0
        self._step_cache: Optional[int] = None
        self.step_reads = 0           # engine reads of global_step (at most one per run() and one per session creation)
        self.graph = getattr(engine, "graph", engine)

    # -- engine pass-throughs
    def global_step_value(self) -> int:
        """``global_step`` as of the end of the last ``run`` (or of session creation).  Every hook of one ``run``
        shares ONE read of the engine's counter - TF fetches ``global_step`` once per ``run`` for all hooks too -
        and the engine serves that read from its host mirror, so hooks cost no device synchronisation."""
        if self._step_cache is None:
            self._step_cache = int(self.engine.global_step_value())
            self.step_reads += 1
        return self._step_cache

    def describe_graph(self) -> str:
        d = getattr(self.engine, "describe", None)
        return d() if d else repr(self.engine)

    def _saver(self, saver=None):
        if saver is not None:
            return saver
        if self.scaffold.saver is None:
            self.scaffold.saver = self.engine.make_saver()
        return self.scaffold.saver

    def save_checkpoint(self, save_path: str, step: int, saver=None) -> str:
        return self.engine.save_checkpoint(self._saver(saver), save_path, step)

    # -- lifecycle
    def _create(self, first: bool):
        """Chief: restore from the latest checkpoint or initialise; then publish the state to the other tasks."""
        self.engine.ensure_built()
        restored = False
        if self.is_chief and self.checkpoint_dir:
            latest = saver_mod.latest_checkpoint(self.checkpoint_dir)
            if latest:
                print("INFO: Restoring parameters from %s" % latest, flush=True)
                self.engine.restore_checkpoint(self._saver(), latest)
                restored = True
        if self.is_chief and not restored and self.scaffold.init_fn is not None and first:
            self.scaffold.init_fn(self)
        self.engine.synchronize_state(self.is_chief)   # chief's values win everywhere; non-chiefs wait here
        self._step_cache = None
        return restored

    def __enter__(self):
        for h in self.hooks:
            h.begin()
        self._create(first=True)
        for h in self.hooks:
            h.after_create_session(self, None)
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is not None and issubclass(exc_type, (OutOfRangeError, StopIteration)):
            self.close()
            return True
        self.close(run_end_hooks=exc_type is None)
        return False

    def should_stop(self) -> bool:
        return self._stop or self._closed

    def request_stop(self):
        self._stop = True

    def run(self, fetches, feed_dict=None):
        """Run ``fetches`` with the hooks' extra fetches; recoverable errors trigger restore-and-retry."""
        if self._closed:
            raise RuntimeError("session is closed")
        while True:
            try:
                return self._run_once(fetches, feed_dict)
            except RECOVERABLE_ERRORS + _transient_errors() as e:
                self._recoveries += 1
                if self._recoveries > self.max_recoveries:
                    raise
                print("INFO: recoverable error (%s); re-creating the session from the last checkpoint" % e)
                time.sleep(self.recovery_wait_secs)
                self._create(first=False)

    def _run_once(self, fetches, feed_dict):
        ctx = SessionRunContext(SessionRunArgs(fetches, feed_dict), self)
        extra = []
        for h in self.hooks:
            req = h.before_run(ctx)
            extra.append(req.fetches if req is not None else None)
            if req is not None and req.feed_dict:
                feed_dict = dict(feed_dict or {}, **req.feed_dict)
        hook_fetches = [f for f in extra if f is not None]
        results = self.engine.run({"caller": fetches, "hooks": hook_fetches}, feed_dict)
        self._step_cache = None       # the run may have advanced the counter: the first hook that asks re-reads it
        hook_results = iter(results["hooks"])
        for h, f in zip(self.hooks, extra):
            h.after_run(ctx, SessionRunValues(next(hook_results) if f is not None else None))
        if ctx.stop_requested:
            self._stop = True
        return results["caller"]

    def check_health(self):
        """Raise ``ClusterLostError`` if the failure detector of this session has lost a task (for loops that do not
        go through ``run``, such as ``model.fit`` inside a session)."""
        self._step_cache = None       # training outside run() (model.fit) has moved the counter since the last read
        for h in self.hooks:
            d = getattr(h, "detector", None)
            if d is not None:
                d.check()

    def close(self, run_end_hooks: bool = True):
        if self._closed:
            return
        # training that never went through run() (model.fit inside the session) has advanced the counter: the end
        # hooks - the final checkpoint's name among them - must see the real step, not the one cached at creation
        self._step_cache = None
        if run_end_hooks:
            for h in self.hooks:
                h.end(self)
        self._closed = True


def MonitoredTrainingSession(master: str = "", is_chief: bool = True, checkpoint_dir: Optional[str] = None,
                             scaffold: Optional[Scaffold] = None, hooks: Optional[Sequence[SessionRunHook]] = None,
                             chief_only_hooks: Optional[Sequence[SessionRunHook]] = None,
                             save_checkpoint_secs: Optional[float] = 600.0, save_summaries_steps: Optional[int] = 100,
                             save_summaries_secs: Optional[float] = None, config=None,
                             stop_grace_period_secs: float = 120.0, log_step_count_steps: Optional[int] = 100,
                             engine=None, max_recoveries: int = 8) -> MonitoredSession:
    """``tf.train.MonitoredTrainingSession`` (/root/reference/example.py:189-192).

    ``engine`` defaults to the default graph of :mod:`distributed_tensorflow_amd.compat`.  When
    ``checkpoint_dir`` is given the chief gets the TF default hooks: step counter (every
    ``log_step_count_steps``), summary saver (every ``save_summaries_steps``) and checkpoint saver
    (every ``save_checkpoint_secs``, plus first step and end).
    """
    if engine is None:
        from . import compat
        engine = compat.get_default_graph().engine()
    all_hooks: List[SessionRunHook] = []
    if is_chief and chief_only_hooks:
        all_hooks.extend(chief_only_hooks)
    collective = bool(getattr(getattr(engine, "strategy", None), "collective_checkpoint", False))
    if (not is_chief) and checkpoint_dir and collective and save_checkpoint_secs:
        all_hooks.append(CheckpointSaverHook(checkpoint_dir, save_secs=save_checkpoint_secs,
                                             saver=scaffold.saver if scaffold else None))
    if is_chief and checkpoint_dir:
        if log_step_count_steps:
            all_hooks.append(StepCounterHook(every_n_steps=log_step_count_steps, output_dir=checkpoint_dir, log=None))
            all_hooks.append(StepLatencyHook(every_n_steps=log_step_count_steps, output_dir=checkpoint_dir))
        if save_summaries_steps:
            all_hooks.append(SummarySaverHook(save_steps=save_summaries_steps, output_dir=checkpoint_dir,
                                              summary_op=scaffold.summary_op if scaffold else None))
        if save_checkpoint_secs:
            all_hooks.append(CheckpointSaverHook(checkpoint_dir, save_secs=save_checkpoint_secs,
                                                 saver=scaffold.saver if scaffold else None))
    all_hooks.extend(hooks or [])
    from .cluster import current_server
    srv = current_server()
    if master and srv is not None and srv.world_size > 1 and master == srv.target \
            and os.environ.get("DTF_HEARTBEAT", "1") != "0":
        from .failure import HeartbeatHook
        all_hooks.append(HeartbeatHook(srv))      # a lost task surfaces as ClusterLostError instead of a hang
    return MonitoredSession(engine, is_chief=is_chief, master=master, checkpoint_dir=checkpoint_dir, hooks=all_hooks,
                            scaffold=scaffold, max_recoveries=max_recoveries,
                            stop_grace_period_secs=stop_grace_period_secs)

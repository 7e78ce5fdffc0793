"""Keras-style callbacks for ``Sequential.fit`` (``callbacks=[TensorBoard(log_dir=...)]`` at
/root/reference/example2.py:197-200)."""
from __future__ import annotations

import sys
import time
from typing import Dict, List, Optional


class Callback:
    def __init__(self):
        self.model = None
        self.params: Dict = {}

    def set_model(self, model):
        self.model = model

    def set_params(self, params):
        self.params = params

    def on_train_begin(self, logs=None): pass
    def on_train_end(self, logs=None): pass
    def on_epoch_begin(self, epoch, logs=None): pass
    def on_epoch_end(self, epoch, logs=None): pass
    def on_batch_begin(self, batch, logs=None): pass
    def on_batch_end(self, batch, logs=None): pass


class History(Callback):
    """Per-epoch metric lists, returned by ``fit`` (keys ``loss``/``acc``/``val_loss``/``val_acc``)."""

    def on_train_begin(self, logs=None):
        self.epoch: List[int] = []
        self.history: Dict[str, List[float]] = {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class ProgbarLogger(Callback):
    """One console line per epoch in the Keras verbose=2 layout."""

    def __init__(self, stream=None):
        super().__init__()
        self.stream = stream or sys.stdout

    def on_epoch_begin(self, epoch, logs=None):
        self._t0 = time.time()
        self.stream.write("Epoch %d/%d\n" % (epoch + 1, self.params.get("epochs", 0)))

    def on_epoch_end(self, epoch, logs=None):
        dt = time.time() - self._t0
        parts = ["%s: %.4f" % (k, v) for k, v in (logs or {}).items()]
        self.stream.write(" - %.1fs - %s\n" % (dt, " - ".join(parts)))
        self.stream.flush()


class TensorBoard(Callback):
    """Writes per-epoch scalars (and the model description once) to a TensorBoard event file."""

    def __init__(self, log_dir: str = "./logs", write_graph: bool = True):
        super().__init__()
        self.log_dir = log_dir
        self.write_graph = write_graph
        self.writer = None

    def on_train_begin(self, logs=None):
        from .summary import FileWriter
        self.writer = FileWriter(self.log_dir)
        if self.write_graph and self.model is not None:
            self.writer.add_graph(self.model)

    def on_epoch_end(self, epoch, logs=None):
        if self.writer is None:
            return
        for k, v in (logs or {}).items():
            self.writer.add_scalar(k, float(v), epoch)
        self.writer.flush()

    def on_train_end(self, logs=None):
        if self.writer is not None:
            self.writer.close()
            self.writer = None


def _improved(value: float, best: Optional[float], mode: str, min_delta: float = 0.0) -> bool:
    if best is None:
        return True
    return value < best - min_delta if mode == "min" else value > best + min_delta


def _mode_for(monitor: str, mode: str) -> str:
    if mode in ("min", "max"):
        return mode
    return "max" if "acc" in monitor else "min"


class ModelCheckpoint(Callback):
    """Save the model after every ``period`` epochs as a TensorBundle checkpoint (``Sequential.save_weights``).
    ``filepath`` may contain ``{epoch}`` (1-based, Keras convention) and metric names, e.g.
    ``"ckpt/weights.{epoch:02d}-{val_loss:.3f}"``.  ``save_best_only`` keeps only improvements of ``monitor``."""

    def __init__(self, filepath: str, monitor: str = "val_loss", save_best_only: bool = False, mode: str = "auto",
                 period: int = 1, verbose: int = 0):
        super().__init__()
        self.filepath, self.monitor, self.save_best_only = filepath, monitor, save_best_only
        self.mode = _mode_for(monitor, mode)
        self.period, self.verbose = max(1, int(period)), verbose
        self.best: Optional[float] = None
        self.saved: List[str] = []
        self._since = 0

    def on_epoch_end(self, epoch, logs=None):
        import os
        logs = logs or {}
        self._since += 1
        if self._since < self.period:
            return
        self._since = 0
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None or not _improved(float(cur), self.best, self.mode):
                return
            self.best = float(cur)
        path = self.filepath.format(epoch=epoch + 1, **logs)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self.saved.append(self.model.save_weights(path))
        if self.verbose:
            sys.stdout.write("Epoch %05d: saving model to %s\n" % (epoch + 1, path))


class EarlyStopping(Callback):
    """Stop ``fit`` when ``monitor`` has not improved by ``min_delta`` for ``patience`` epochs;
    ``restore_best_weights`` puts the best epoch's variables back."""

    def __init__(self, monitor: str = "val_loss", min_delta: float = 0.0, patience: int = 0, mode: str = "auto",
                 restore_best_weights: bool = False, verbose: int = 0):
        super().__init__()
        self.monitor, self.min_delta, self.patience = monitor, abs(float(min_delta)), int(patience)
        self.mode = _mode_for(monitor, mode)
        self.restore_best_weights, self.verbose = restore_best_weights, verbose

    def on_train_begin(self, logs=None):
        self.best: Optional[float] = None
        self.wait = 0
        self.stopped_epoch: Optional[int] = None
        self._best_flat = None

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if _improved(float(cur), self.best, self.mode, self.min_delta):
            self.best, self.wait = float(cur), 0
            if self.restore_best_weights:
                self._best_flat = self.model.store.flat.clone()
            return
        self.wait += 1
        if self.wait > self.patience or (self.patience == 0 and self.wait >= 1):
            self.stopped_epoch = epoch
            self.model.stop_training = True
            if self.restore_best_weights and self._best_flat is not None:
                self.model.store.flat.copy_(self._best_flat)
            if self.verbose:
                sys.stdout.write("Epoch %05d: early stopping\n" % (epoch + 1))


class LearningRateScheduler(Callback):
    """``schedule(epoch, lr) -> lr`` (or ``schedule(epoch)``) evaluated at the start of every epoch."""

    def __init__(self, schedule, verbose: int = 0):
        super().__init__()
        self.schedule, self.verbose = schedule, verbose

    def on_epoch_begin(self, epoch, logs=None):
        lr = self.model.optimizer.lr
        try:
            new = self.schedule(epoch, lr)
        except TypeError:
            new = self.schedule(epoch)
        self.model.set_learning_rate(float(new))
        if self.verbose:
            sys.stdout.write("Epoch %05d: learning rate %.6g\n" % (epoch + 1, float(new)))

    def on_epoch_end(self, epoch, logs=None):
        if logs is not None:
            logs["lr"] = self.model.optimizer.lr


class ReduceLROnPlateau(Callback):
    """Multiply the learning rate by ``factor`` when ``monitor`` has not improved for ``patience`` epochs."""

    def __init__(self, monitor: str = "val_loss", factor: float = 0.1, patience: int = 10, min_delta: float = 1e-4,
                 min_lr: float = 0.0, mode: str = "auto", verbose: int = 0):
        super().__init__()
        if factor >= 1.0:
            raise ValueError("ReduceLROnPlateau needs factor < 1")
        self.monitor, self.factor, self.patience = monitor, float(factor), int(patience)
        self.min_delta, self.min_lr, self.verbose = abs(float(min_delta)), float(min_lr), verbose
        self.mode = _mode_for(monitor, mode)

    def on_train_begin(self, logs=None):
        self.best: Optional[float] = None
        self.wait = 0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if _improved(float(cur), self.best, self.mode, self.min_delta):
            self.best, self.wait = float(cur), 0
            return
        self.wait += 1
        if self.wait >= self.patience:
            old = self.model.optimizer.lr
            new = max(old * self.factor, self.min_lr)
            if new < old:
                self.model.set_learning_rate(new)
                if self.verbose:
                    sys.stdout.write("Epoch %05d: reducing learning rate to %.6g\n" % (epoch + 1, new))
            self.wait = 0


class CSVLogger(Callback):
    """One CSV row of the epoch's metrics per epoch."""

    def __init__(self, filename: str, separator: str = ",", append: bool = False):
        super().__init__()
        self.filename, self.sep, self.append = filename, separator, append
        self._f = None
        self._keys: Optional[List[str]] = None

    def on_train_begin(self, logs=None):
        self._f = open(self.filename, "a" if self.append else "w")
        self._keys = None

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        if self._keys is None:
            self._keys = sorted(logs)
            if not (self.append and self._f.tell() > 0):
                self._f.write(self.sep.join(["epoch"] + self._keys) + "\n")
        self._f.write(self.sep.join([str(epoch)] + ["%.8g" % float(logs.get(k, float("nan"))) for k in self._keys]) + "\n")
        self._f.flush()

    def on_train_end(self, logs=None):
        if self._f is not None:
            self._f.close()
            self._f = None


class LambdaCallback(Callback):
    def __init__(self, on_epoch_end=None, on_batch_end=None, on_train_begin=None, on_train_end=None):
        super().__init__()
        self._oe, self._ob, self._tb, self._te = on_epoch_end, on_batch_end, on_train_begin, on_train_end

    def on_epoch_end(self, epoch, logs=None):
        if self._oe:
            self._oe(epoch, logs)

    def on_batch_end(self, batch, logs=None):
        if self._ob:
            self._ob(batch, logs)

    def on_train_begin(self, logs=None):
        if self._tb:
            self._tb(logs)

    def on_train_end(self, logs=None):
        if self._te:
            self._te(logs)


class ClusterHealth(Callback):
    """Inside a cluster session ``model.fit`` never passes through ``session.run``, so the session's heartbeat hook
    gets no chance to report a lost task; this callback asks the session at every epoch boundary instead and lets the
    resulting ``ClusterLostError`` end ``fit``.  (A worker that is stuck mid-epoch in a transfer with the dead task is
    ended by the failure detector itself.)"""

    def __init__(self, session):
        super().__init__()
        self.session = session

    def on_epoch_begin(self, epoch, logs=None):
        self.session.check_health()

    def on_epoch_end(self, epoch, logs=None):
        self.session.check_health()


class CallbackList:
    def __init__(self, callbacks: Optional[List[Callback]], model, params):
        self.callbacks = list(callbacks or [])
        for c in self.callbacks:
            c.set_model(model)
            c.set_params(params)
        self.wants_batch_logs = any(type(c).on_batch_end is not Callback.on_batch_end for c in self.callbacks)

    def call(self, hook: str, *args):
        for c in self.callbacks:
            getattr(c, hook)(*args)

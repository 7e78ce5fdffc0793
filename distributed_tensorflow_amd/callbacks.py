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

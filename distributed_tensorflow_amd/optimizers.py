"""Optimizers over the flat variable buffer.

``Adam`` reproduces ``tf.train.AdamOptimizer()`` (/root/reference/example.py:168) and the
Keras ``'adam'`` string (/root/reference/example2.py:165): lr 1e-3, beta1 .9, beta2 .999,
eps 1e-8, TensorFlow epsilon placement.  ``apply`` takes a ``[begin, end)`` range so the
same fused kernel serves "all variables" (single process), "my shard" (sync replicas
after a reduce-scatter) and "this owner's shard" (parameter-server owner).
"""
from __future__ import annotations

from typing import Dict, Optional

from .ops import functional as F
from .variables import VariableStore


class Optimizer:
    #: slot buffer name -> suffix used in TF checkpoints ("<var>/<suffix>")
    slot_checkpoint_names: Dict[str, str] = {}
    name = "optimizer"

    def apply(self, store: VariableStore, begin: int = 0, end: Optional[int] = None, grad_scale: float = 1.0,
              grad=None, step=None):
        """Update ``store.flat[begin:end]`` from ``grad`` (default: ``store.flat_grad``) on that range."""
        raise NotImplementedError

    def finish_step(self, store: VariableStore):
        """Advance ``global_step`` (and optimizer-wide scalars) once per applied update."""
        F.step_advance(store.global_step, None)

    def create_slots(self, store: VariableStore):
        pass

    def apply_and_finish(self, store: VariableStore, grad_scale: float = 1.0):
        """Whole-buffer update plus the ``global_step`` increment (one launch where the optimizer can fuse them)."""
        self.apply(store, grad_scale=grad_scale)
        self.finish_step(store)


class Adam(Optimizer):
    slot_checkpoint_names = {"m": "Adam", "v": "Adam_1"}
    name = "Adam"

    def __init__(self, lr: float = 1e-3, beta_1: float = 0.9, beta_2: float = 0.999, epsilon: float = 1e-8,
                 learning_rate: Optional[float] = None, beta1: Optional[float] = None,
                 beta2: Optional[float] = None):
        self.lr = float(learning_rate if learning_rate is not None else lr)
        self.beta1 = float(beta1 if beta1 is not None else beta_1)
        self.beta2 = float(beta2 if beta2 is not None else beta_2)
        self.epsilon = float(epsilon)

    def create_slots(self, store):
        store.slot("m")
        store.slot("v")
        # TF's non-slot variables beta1_power / beta2_power START at beta (not 1) and are multiplied once per applied
        # update, so a checkpoint written after t updates holds beta^(t+1) - what tf.train.AdamOptimizer expects to
        # find when it resumes from one of these files.  (The kernels' own bias correction uses global_step.)
        if not getattr(store, "_beta_powers_initialised", False) and getattr(store, "beta_powers", None) is not None:
            store.beta_powers.copy_(initial_beta_powers(self, store.beta_powers.device))
            store._beta_powers_initialised = True

    def apply(self, store, begin=0, end=None, grad_scale=1.0, grad=None, step=None):
        end = store.flat_len if end is None else end
        g = store.flat_grad if grad is None else grad
        F.adam_apply(store.flat[begin:end], store.slot("m")[begin:end], store.slot("v")[begin:end], g[begin:end],
                     store.global_step if step is None else step, self.lr, self.beta1, self.beta2, self.epsilon,
                     grad_scale)

    def finish_step(self, store):
        F.step_advance(store.global_step, store.beta_powers, self.beta1, self.beta2)

    def apply_and_finish(self, store, grad_scale=1.0):
        F.adam_apply(store.flat, store.slot("m"), store.slot("v"), store.flat_grad, store.global_step, self.lr, self.beta1,
                     self.beta2, self.epsilon, grad_scale, advance=store.global_step, beta_powers=store.beta_powers)


class SGD(Optimizer):
    slot_checkpoint_names = {"momentum": "Momentum"}
    name = "SGD"

    def __init__(self, lr: float = 0.01, momentum: float = 0.0, learning_rate: Optional[float] = None):
        self.lr = float(learning_rate if learning_rate is not None else lr)
        self.momentum = float(momentum)

    def create_slots(self, store):
        if self.momentum > 0.0:
            store.slot("momentum")

    def apply(self, store, begin=0, end=None, grad_scale=1.0, grad=None, step=None):
        end = store.flat_len if end is None else end
        g = store.flat_grad if grad is None else grad
        mom = store.slot("momentum")[begin:end] if self.momentum > 0.0 else None
        F.sgd_apply(store.flat[begin:end], g[begin:end], self.lr, mom, self.momentum, grad_scale)

    def apply_and_finish(self, store, grad_scale=1.0):
        mom = store.slot("momentum") if self.momentum > 0.0 else None
        F.sgd_apply(store.flat, store.flat_grad, self.lr, mom, self.momentum, grad_scale, advance=store.global_step)


def initial_beta_powers(optimizer, device):
    """Initial value of the ``[beta1_power, beta2_power]`` pair for ``optimizer`` (ones for optimizers without them)."""
    import torch
    if isinstance(optimizer, Adam):
        return torch.tensor([optimizer.beta1, optimizer.beta2], dtype=torch.float32, device=device)
    return torch.ones(2, dtype=torch.float32, device=device)


# TF 1.x class names (tf.train.*)
AdamOptimizer = Adam
GradientDescentOptimizer = SGD


def MomentumOptimizer(learning_rate, momentum):
    return SGD(lr=learning_rate, momentum=momentum)


_REGISTRY = {"adam": Adam, "sgd": SGD}


def get(spec) -> Optimizer:
    """Resolve ``compile(optimizer=...)``.  The strings are the Keras ones (/root/reference/example2.py:165).  The
    reference pins keras==2.0.8 (/root/reference/requirements.txt:7), whose ``Adam`` signature still has
    ``epsilon=1e-8`` - the same value as ``tf.train.AdamOptimizer()`` (/root/reference/example.py:168); the
    ``epsilon=None -> K.epsilon() == 1e-7`` default only arrived in Keras 2.1.3.  So ``'adam'`` is ``Adam()``."""
    if isinstance(spec, Optimizer):
        return spec
    if isinstance(spec, str) and spec.lower() in _REGISTRY:
        return _REGISTRY[spec.lower()]()
    raise ValueError("unknown optimizer %r (have: adam, sgd)" % (spec,))

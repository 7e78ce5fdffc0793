"""Optimizers over the flat variable buffer.

``Adam`` reproduces ``tf.train.AdamOptimizer()`` (/root/reference/example.py:168) and the
Keras ``'adam'`` string (/root/reference/example2.py:165): lr 1e-3, beta1 .9, beta2 .999,
eps 1e-8, TensorFlow epsilon placement.  ``apply`` takes a ``[begin, end)`` range so the
same fused kernel serves "all variables" (single process), "my shard" (sync replicas
after a reduce-scatter) and "this owner's shard" (parameter-server owner).

``SGD`` (with momentum), ``RMSProp`` and ``Adagrad`` are the other rules a script written against
``tf.train.*Optimizer`` or Keras' optimizer strings may name; each comes in the TensorFlow form (slot names, initial
slot values and epsilon placement of TF's Apply* kernels, so checkpoints carry the names TF expects) and, through
``get('rmsprop')`` / ``get('adagrad')``, in the form of Keras 2.0.8's Python formulas.
"""
from __future__ import annotations

from typing import Dict, Optional

from .ops import functional as F
from .variables import VariableStore


class Optimizer:
    #: slot buffer name -> suffix used in TF checkpoints ("<var>/<suffix>")
    slot_checkpoint_names: Dict[str, str] = {}
    name = "optimizer"

    @property
    def slot_order(self):
        """Slot buffers in the order a variable owner lays them out behind its variables."""
        return list(self.slot_checkpoint_names)

    def slot_initial(self, slot_name: str) -> float:
        """Value a fresh slot buffer starts from."""
        return 0.0

    def describe(self) -> dict:
        """``{"name", "args"}`` from which ``from_description`` rebuilds this optimizer in another process."""
        raise NotImplementedError

    def pump_parameters(self):
        """``(code, lr, b1, b2, eps, momentum)`` of the native owner service (`csrc/kernels/owner_pump.hip`)."""
        raise NotImplementedError

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

    def describe(self):
        return {"name": "Adam", "args": {"lr": self.lr, "beta_1": self.beta1, "beta_2": self.beta2, "epsilon": self.epsilon}}

    def pump_parameters(self):
        return 0, self.lr, self.beta1, self.beta2, self.epsilon, 0.0

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

    @property
    def slot_order(self):
        return ["momentum"] if self.momentum > 0.0 else []

    def describe(self):
        return {"name": "SGD", "args": {"lr": self.lr, "momentum": self.momentum}}

    def pump_parameters(self):
        return 1, self.lr, 0.0, 0.0, 0.0, self.momentum

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


class RMSProp(Optimizer):
    """``tf.train.RMSPropOptimizer(learning_rate, decay=0.9, momentum=0.0, epsilon=1e-10)``: TF's ApplyRMSProp
    (``var -= lr g / sqrt(ms + eps)``, ``ms`` starting at ONE, slots ``<var>/RMSProp`` and - always present in TF's
    checkpoints, kept here only with ``momentum > 0`` - ``<var>/RMSProp_1``).  ``keras=True``: Keras 2.0.8's
    ``RMSprop(lr=0.001, rho=0.9, epsilon=1e-8)``, accumulator from zero and epsilon outside the root."""

    slot_checkpoint_names = {"rms": "RMSProp", "momentum": "RMSProp_1"}
    name = "RMSProp"

    def __init__(self, lr: float = 0.001, decay: float = 0.9, momentum: float = 0.0, epsilon: Optional[float] = None,
                 learning_rate: Optional[float] = None, rho: Optional[float] = None, keras: bool = False):
        self.lr = float(learning_rate if learning_rate is not None else lr)
        self.rho = float(rho if rho is not None else decay)
        self.momentum = float(momentum)
        self.keras = bool(keras)
        self.epsilon = float(epsilon if epsilon is not None else (1e-8 if keras else 1e-10))

    @property
    def slot_order(self):
        return ["rms", "momentum"] if self.momentum > 0.0 else ["rms"]

    def slot_initial(self, slot_name):
        return 1.0 if slot_name == "rms" and not self.keras else 0.0

    def describe(self):
        return {"name": "RMSProp", "args": {"lr": self.lr, "decay": self.rho, "momentum": self.momentum,
                                            "epsilon": self.epsilon, "keras": self.keras}}

    def pump_parameters(self):
        # b1 carries rho, b2 the epsilon placement
        return 2, self.lr, self.rho, 0.0 if self.keras else 1.0, self.epsilon, self.momentum

    def create_slots(self, store):
        for name in self.slot_order:
            store.slot(name, self.slot_initial(name))

    def apply(self, store, begin=0, end=None, grad_scale=1.0, grad=None, step=None):
        end = store.flat_len if end is None else end
        g = store.flat_grad if grad is None else grad
        mom = store.slot("momentum")[begin:end] if self.momentum > 0.0 else None
        F.rmsprop_apply(store.flat[begin:end], store.slot("rms", self.slot_initial("rms"))[begin:end], g[begin:end], self.lr,
                        self.rho, mom, self.momentum, self.epsilon, not self.keras, grad_scale)

    def apply_and_finish(self, store, grad_scale=1.0):
        mom = store.slot("momentum") if self.momentum > 0.0 else None
        F.rmsprop_apply(store.flat, store.slot("rms", self.slot_initial("rms")), store.flat_grad, self.lr, self.rho, mom,
                        self.momentum, self.epsilon, not self.keras, grad_scale, advance=store.global_step)


class Adagrad(Optimizer):
    """``tf.train.AdagradOptimizer(learning_rate, initial_accumulator_value=0.1)``: TF's ApplyAdagrad
    (``acc += g^2; var -= lr g / sqrt(acc)``, slot ``<var>/Adagrad``).  ``keras=True``: Keras 2.0.8's
    ``Adagrad(lr=0.01, epsilon=1e-8)``, accumulator from zero and ``sqrt(acc) + epsilon``."""

    slot_checkpoint_names = {"accumulator": "Adagrad"}
    name = "Adagrad"

    def __init__(self, lr: float = 0.01, initial_accumulator_value: Optional[float] = None, epsilon: Optional[float] = None,
                 learning_rate: Optional[float] = None, keras: bool = False):
        self.lr = float(learning_rate if learning_rate is not None else lr)
        self.keras = bool(keras)
        self.initial = float(initial_accumulator_value if initial_accumulator_value is not None else (0.0 if keras else 0.1))
        self.epsilon = float(epsilon if epsilon is not None else (1e-8 if keras else 0.0))

    def slot_initial(self, slot_name):
        return self.initial

    def describe(self):
        return {"name": "Adagrad", "args": {"lr": self.lr, "initial_accumulator_value": self.initial,
                                            "epsilon": self.epsilon, "keras": self.keras}}

    def pump_parameters(self):
        return 3, self.lr, 0.0, 0.0 if self.keras else 1.0, self.epsilon, 0.0

    def create_slots(self, store):
        store.slot("accumulator", self.initial)

    def apply(self, store, begin=0, end=None, grad_scale=1.0, grad=None, step=None):
        end = store.flat_len if end is None else end
        g = store.flat_grad if grad is None else grad
        F.adagrad_apply(store.flat[begin:end], store.slot("accumulator", self.initial)[begin:end], g[begin:end], self.lr,
                        self.epsilon, not self.keras, grad_scale)

    def apply_and_finish(self, store, grad_scale=1.0):
        F.adagrad_apply(store.flat, store.slot("accumulator", self.initial), store.flat_grad, self.lr, self.epsilon,
                        not self.keras, grad_scale, advance=store.global_step)


class Adadelta(Optimizer):
    """``tf.train.AdadeltaOptimizer(learning_rate=0.001, rho=0.95, epsilon=1e-8)`` (slots ``<var>/Adadelta`` and
    ``<var>/Adadelta_1``); ``get('adadelta')`` is Keras' ``Adadelta(lr=1.0, rho=0.95, epsilon=1e-8)`` - one formula,
    different default rates."""

    slot_checkpoint_names = {"accum": "Adadelta", "accum_update": "Adadelta_1"}
    name = "Adadelta"

    def __init__(self, lr: float = 0.001, rho: float = 0.95, epsilon: float = 1e-8, learning_rate: Optional[float] = None):
        self.lr = float(learning_rate if learning_rate is not None else lr)
        self.rho, self.epsilon = float(rho), float(epsilon)

    def describe(self):
        return {"name": "Adadelta", "args": {"lr": self.lr, "rho": self.rho, "epsilon": self.epsilon}}

    def pump_parameters(self):
        return 4, self.lr, self.rho, 0.0, self.epsilon, 0.0

    def create_slots(self, store):
        store.slot("accum")
        store.slot("accum_update")

    def apply(self, store, begin=0, end=None, grad_scale=1.0, grad=None, step=None):
        end = store.flat_len if end is None else end
        g = store.flat_grad if grad is None else grad
        F.adadelta_apply(store.flat[begin:end], store.slot("accum")[begin:end], store.slot("accum_update")[begin:end],
                         g[begin:end], self.lr, self.rho, self.epsilon, grad_scale)

    def apply_and_finish(self, store, grad_scale=1.0):
        F.adadelta_apply(store.flat, store.slot("accum"), store.slot("accum_update"), store.flat_grad, self.lr, self.rho,
                         self.epsilon, grad_scale, advance=store.global_step)


def from_description(desc: dict) -> Optimizer:
    """Inverse of ``Optimizer.describe`` (how a parameter-server task learns which rule its shard is updated by)."""
    classes = {"Adam": Adam, "SGD": SGD, "RMSProp": RMSProp, "Adagrad": Adagrad, "Adadelta": Adadelta}
    return classes[desc["name"]](**desc["args"])


def initial_beta_powers(optimizer, device):
    """Initial value of the ``[beta1_power, beta2_power]`` pair for ``optimizer`` (ones for optimizers without them)."""
    import torch
    if isinstance(optimizer, Adam):
        return torch.tensor([optimizer.beta1, optimizer.beta2], dtype=torch.float32, device=device)
    return torch.ones(2, dtype=torch.float32, device=device)


# TF 1.x class names (tf.train.*)
AdamOptimizer = Adam
GradientDescentOptimizer = SGD


RMSPropOptimizer = RMSProp
AdagradOptimizer = Adagrad
AdadeltaOptimizer = Adadelta


def MomentumOptimizer(learning_rate, momentum):
    return SGD(lr=learning_rate, momentum=momentum)


# the Keras strings build the Keras forms
_REGISTRY = {"adam": Adam, "sgd": SGD, "rmsprop": lambda: RMSProp(keras=True), "adagrad": lambda: Adagrad(keras=True),
             "adadelta": lambda: Adadelta(lr=1.0)}


def get(spec) -> Optimizer:
    """Resolve ``compile(optimizer=...)``.  The strings are the Keras ones (/root/reference/example2.py:165).  The
    reference pins keras==2.0.8 (/root/reference/requirements.txt:7), whose ``Adam`` signature still has
    ``epsilon=1e-8`` - the same value as ``tf.train.AdamOptimizer()`` (/root/reference/example.py:168); the
    ``epsilon=None -> K.epsilon() == 1e-7`` default only arrived in Keras 2.1.3.  So ``'adam'`` is ``Adam()``."""
    if isinstance(spec, Optimizer):
        return spec
    if isinstance(spec, str) and spec.lower() in _REGISTRY:
        return _REGISTRY[spec.lower()]()
    raise ValueError("unknown optimizer %r (have: %s)" % (spec, ", ".join(sorted(_REGISTRY))))

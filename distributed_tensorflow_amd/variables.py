"""Variables and their flat device storage.

TensorFlow keeps every variable as its own tensor on a parameter-server task
(/root/reference/example.py:136-139 places them with ``replica_device_setter``).  Here
all trainable variables of a model are *views into one flat fp32 buffer*, with matching
flat buffers for gradients and optimizer slots.  That layout is what makes the MI355X
design cheap:

* one fused apply-gradients launch covers every variable (or exactly one 1/N shard);
* sync-replica training is a reduce-scatter / all-gather over contiguous ranges;
* parameter-server "owners" are just ``[begin, end)`` ranges of the flat buffer.

Variable names follow the Keras-on-TF naming the reference's checkpoints would carry
(``dense_1/kernel``, ``dense_1/bias``, ``global_step``, ``beta1_power``, ``<var>/Adam``...).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

ALIGN = 64  # floats: every variable starts on a 256-byte boundary of the flat buffer


def _round_up(n: int, a: int) -> int:
    return (n + a - 1) // a * a


# --------------------------------------------------------------------------- initializers
def glorot_uniform(shape: Sequence[int], generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Keras ``glorot_uniform``: U(-l, l) with l = sqrt(6 / (fan_in + fan_out)).

    For a conv kernel [kh, kw, cin, cout] the receptive field multiplies both fans.
    """
    if len(shape) == 2:
        fan_in, fan_out = shape
    elif len(shape) == 4:
        rf = shape[0] * shape[1]
        fan_in, fan_out = shape[2] * rf, shape[3] * rf
    else:
        fan_in = fan_out = int(math.prod(shape)) if shape else 1
    limit = math.sqrt(6.0 / float(fan_in + fan_out))
    return (torch.rand(tuple(shape), generator=generator, dtype=torch.float32) * 2.0 - 1.0) * limit


def _fans(shape: Sequence[int]) -> Tuple[int, int]:
    if len(shape) == 2:
        return int(shape[0]), int(shape[1])
    if len(shape) == 4:
        rf = shape[0] * shape[1]
        return int(shape[2] * rf), int(shape[3] * rf)
    n = int(math.prod(shape)) if shape else 1
    return n, n


def he_uniform(shape: Sequence[int], generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Keras ``he_uniform``: U(-l, l) with l = sqrt(6 / fan_in) - keeps the activation scale of ReLU stacks constant
    with depth, which very wide layers need."""
    limit = math.sqrt(6.0 / float(_fans(shape)[0]))
    return (torch.rand(tuple(shape), generator=generator, dtype=torch.float32) * 2.0 - 1.0) * limit


def uniform_limit(initializer, shape) -> Optional[float]:
    """``l`` when ``initializer`` draws from U(-l, l) (0 for zeros); None for anything else.  What the device-side
    initialisation of large models needs to know about an initializer."""
    fan_in, fan_out = _fans(shape)
    if initializer is glorot_uniform:
        return math.sqrt(6.0 / float(fan_in + fan_out))
    if initializer is he_uniform:
        return math.sqrt(6.0 / float(fan_in))
    if initializer is zeros:
        return 0.0
    return None


def zeros(shape, generator=None):
    return torch.zeros(tuple(shape), dtype=torch.float32)


def ones(shape, generator=None):
    return torch.ones(tuple(shape), dtype=torch.float32)


INITIALIZERS: Dict[str, Callable] = {"glorot_uniform": glorot_uniform, "he_uniform": he_uniform, "zeros": zeros,
                                     "ones": ones}

# models with more parameters than this are initialised by a kernel on their device (a counter-based generator
# keyed by seed and variable index) instead of the host generator followed by a host-to-device copy
DEVICE_INIT_MIN_PARAMS = 1 << 26


def get_initializer(spec) -> Callable:
    if callable(spec):
        return spec
    if spec not in INITIALIZERS:
        raise ValueError("unknown initializer %r" % (spec,))
    return INITIALIZERS[spec]


# --------------------------------------------------------------------------- variables
class Variable:
    """A named tensor.  ``tensor``/``grad`` become views of the store's flat buffers after ``finalize``."""

    def __init__(self, name: str, shape: Sequence[int], initializer="zeros", trainable: bool = True,
                 dtype: torch.dtype = torch.float32):
        self.name = name
        self.shape = tuple(int(s) for s in shape)
        self.initializer = get_initializer(initializer) if dtype == torch.float32 else initializer
        self.trainable = trainable
        self.dtype = dtype
        self.offset = -1          # element offset in the flat buffer (trainable fp32 only)
        self.tensor: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None

    @property
    def numel(self) -> int:
        return int(math.prod(self.shape)) if self.shape else 1

    def __repr__(self):
        return "Variable(%r, shape=%s, trainable=%s)" % (self.name, self.shape, self.trainable)


class VariableStore:
    """All variables of one model plus their flat storage on one device."""

    def __init__(self):
        self.variables: "OrderedDict[str, Variable]" = OrderedDict()
        self.device: Optional[torch.device] = None
        self.flat: Optional[torch.Tensor] = None       # trainable parameters
        self.flat_grad: Optional[torch.Tensor] = None
        self.slots: Dict[str, torch.Tensor] = {}       # optimizer slot name -> flat buffer
        self.global_step: Optional[torch.Tensor] = None  # int64 [1], device resident
        self.beta_powers: Optional[torch.Tensor] = None  # float32 [2]: beta1_power, beta2_power
        self.flat_len = 0
        self.pad_multiple = ALIGN
        self.finalized = False
        self._name_counts: Dict[str, int] = {}

    # ---- naming (Keras style: dense_1, dense_2, ... per layer class)
    def unique_layer_name(self, prefix: str) -> str:
        n = self._name_counts.get(prefix, 0) + 1
        self._name_counts[prefix] = n
        return "%s_%d" % (prefix, n)

    def create(self, name: str, shape, initializer="zeros", trainable=True, dtype=torch.float32) -> Variable:
        if self.finalized:
            raise RuntimeError("VariableStore is finalized; cannot create %r" % name)
        if name in self.variables:
            raise ValueError("duplicate variable name %r" % name)
        v = Variable(name, shape, initializer, trainable, dtype)
        self.variables[name] = v
        return v

    @property
    def trainable(self) -> List[Variable]:
        return [v for v in self.variables.values() if v.trainable]

    @property
    def num_params(self) -> int:
        return sum(v.numel for v in self.trainable)

    def finalize(self, device, seed: Optional[int] = None, pad_multiple: int = ALIGN,
                 device_init: Optional[bool] = None) -> "VariableStore":
        """Lay out the flat buffers on ``device`` and run the initializers: the host generator and one host-to-device
        copy for ordinary models (the values then do not depend on the device), a device-side counter-based generator
        for models above ``DEVICE_INIT_MIN_PARAMS`` parameters whose initializers are uniform (``device_init`` forces
        either): a billion parameters take milliseconds instead of a 4 GB host pass and copy."""
        if self.finalized:
            return self
        self.device = torch.device(device)
        self.pad_multiple = _round_up(max(int(pad_multiple), ALIGN), ALIGN)
        off = 0
        for v in self.trainable:
            v.offset = off
            off += _round_up(v.numel, self.pad_multiple)
        self.flat_len = _round_up(max(off, ALIGN), self.pad_multiple)
        gen = torch.Generator()
        if seed is None:
            gen.seed()
        else:
            gen.manual_seed(int(seed))
        limits = [uniform_limit(v.initializer, v.shape) for v in self.trainable]
        if device_init is None:
            device_init = self.num_params >= DEVICE_INIT_MIN_PARAMS
        self.device_initialised = bool(device_init) and all(l is not None for l in limits)
        if self.device_initialised:
            from .ops import functional as F
            base = int(gen.initial_seed()) & 0x7FFFFFFF
            self.flat = torch.zeros(self.flat_len, dtype=torch.float32, device=self.device)
            for k, (v, l) in enumerate(zip(self.trainable, limits)):
                if l:
                    F.init_uniform_(self.flat[v.offset:v.offset + v.numel], base + 7919 * (k + 1), -l, l)
        else:
            host = torch.zeros(self.flat_len, dtype=torch.float32)
            for v in self.trainable:
                host[v.offset:v.offset + v.numel] = v.initializer(v.shape, gen).reshape(-1)
            self.flat = host.to(self.device)
        self.flat_grad = torch.zeros(self.flat_len, dtype=torch.float32, device=self.device)
        for v in self.trainable:
            v.tensor = self.flat[v.offset:v.offset + v.numel].view(v.shape)
            v.grad = self.flat_grad[v.offset:v.offset + v.numel].view(v.shape)
        for v in self.variables.values():
            if not v.trainable:
                if v.dtype == torch.float32 and callable(v.initializer):
                    v.tensor = v.initializer(v.shape, gen).to(self.device)      # e.g. a moving variance that starts at one
                else:
                    v.tensor = torch.zeros(v.shape, dtype=v.dtype, device=self.device)
        self.global_step = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.beta_powers = torch.ones(2, dtype=torch.float32, device=self.device)
        self.finalized = True
        return self

    def slot(self, slot_name: str, initial: float = 0.0) -> torch.Tensor:
        """Flat optimizer slot buffer (filled with ``initial`` on first use), same layout as ``flat``."""
        if slot_name not in self.slots:
            self.slots[slot_name] = torch.full((self.flat_len,), float(initial), dtype=torch.float32, device=self.device)
        return self.slots[slot_name]

    def slot_view(self, slot_name: str, var: Variable) -> torch.Tensor:
        return self.slot(slot_name)[var.offset:var.offset + var.numel].view(var.shape)

    def shard_range(self, rank: int, world: int) -> Tuple[int, int]:
        """``[begin, end)`` of the flat buffer owned by ``rank`` out of ``world`` (equal, aligned shards)."""
        if self.flat_len % world != 0:
            raise ValueError("flat length %d is not divisible by world size %d; finalize with pad_multiple"
                             % (self.flat_len, world))
        per = self.flat_len // world
        return rank * per, (rank + 1) * per

    def zero_grad(self):
        self.flat_grad.zero_()

    # ---- (de)serialisation used by the checkpoint saver
    def named_tensors(self, include_slots: bool = True, slot_names: Optional[Dict[str, str]] = None
                      ) -> "OrderedDict[str, torch.Tensor]":
        """Every persistent tensor by its TF-compatible checkpoint name."""
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for v in self.variables.values():
            out[v.name] = v.tensor
        out["global_step"] = self.global_step.view(())
        if include_slots:
            for slot_name in self.slots:
                suffix = (slot_names or {}).get(slot_name, slot_name)
                for v in self.trainable:
                    out["%s/%s" % (v.name, suffix)] = self.slot_view(slot_name, v)
        return out

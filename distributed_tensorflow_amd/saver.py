"""Checkpoint save / restore in TensorFlow's TensorBundle format (native writer/reader in
``csrc/runtime/tensor_bundle.cc``).

Behaviour behind ``checkpoint_dir=FLAGS.log_dir`` at /root/reference/example.py:191 (TF's
default ``Saver(sharded=True, max_to_keep=5)`` + ``CheckpointSaverHook``):

* files ``model.ckpt-<global_step>.index`` / ``.data-0000i-of-0000N`` plus the text-proto
  ``checkpoint`` state file (``model_checkpoint_path`` / ``all_model_checkpoint_paths``);
* one data shard per variable owner (parameter-server task); the chief merges the index;
* the newest ``max_to_keep`` checkpoints are kept;
* ``latest_checkpoint(dir)`` + ``restore`` resume training, ``global_step`` included.

Variable names are the TF/Keras ones (``dense_1/kernel``, ``dense_1/kernel/Adam``,
``beta1_power``, ``global_step``...), so ``tf.train.load_checkpoint`` can read the files.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .ops import _native as nv

# tensorflow DataType enum values
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_INT16, DT_INT8, DT_INT64, DT_BOOL, DT_BFLOAT16, DT_HALF = 1, 2, 3, 4, 5, 6, 9, 10, 14, 19
_TORCH_TO_DT = {torch.float32: DT_FLOAT, torch.float64: DT_DOUBLE, torch.int32: DT_INT32, torch.uint8: DT_UINT8,
                torch.int16: DT_INT16, torch.int8: DT_INT8, torch.int64: DT_INT64, torch.bool: DT_BOOL,
                torch.bfloat16: DT_BFLOAT16, torch.float16: DT_HALF}
_DT_TO_TORCH = {v: k for k, v in _TORCH_TO_DT.items()}

_lib = None


def _rt():
    global _lib
    if _lib is not None:
        return _lib
    lib = nv.runtime()
    nv.declare(lib, "dtf_last_error", [], C.c_char_p)
    nv.declare(lib, "dtf_bundle_writer_create", [C.c_char_p, C.c_int, C.c_int], C.c_void_p)
    nv.declare(lib, "dtf_bundle_writer_add", [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.c_int,
                                              C.c_void_p, C.c_int64, C.c_int])
    nv.declare(lib, "dtf_bundle_writer_finish", [C.c_void_p])
    nv.declare(lib, "dtf_bundle_writer_abort", [C.c_void_p], None)
    nv.declare(lib, "dtf_bundle_merge", [C.c_char_p, C.c_int])
    nv.declare(lib, "dtf_bundle_reader_open", [C.c_char_p], C.c_void_p)
    nv.declare(lib, "dtf_bundle_reader_close", [C.c_void_p], None)
    nv.declare(lib, "dtf_bundle_reader_num_entries", [C.c_void_p])
    nv.declare(lib, "dtf_bundle_reader_num_shards", [C.c_void_p])
    nv.declare(lib, "dtf_bundle_reader_entry", [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int),
                                                C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int),
                                                C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                                C.POINTER(C.c_uint32)])
    nv.declare(lib, "dtf_bundle_reader_read", [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int])
    _lib = lib
    return lib


def _err() -> str:
    return (_rt().dtf_last_error() or b"").decode()


# --------------------------------------------------------------------------- bundle level
class BundleWriter:
    """Writes one checkpoint.  ``only_shard >= 0`` writes just that data shard and a partial index."""

    def __init__(self, prefix: str, num_shards: int = 1, only_shard: int = -1):
        os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
        self.prefix = prefix
        self.num_shards = num_shards
        self._h = _rt().dtf_bundle_writer_create(prefix.encode(), num_shards, only_shard)
        if not self._h:
            raise IOError("cannot create bundle writer for %s" % prefix)

    def add(self, name: str, tensor: torch.Tensor, shard: int = 0):
        t = tensor.detach().to("cpu").contiguous()
        if t.dtype not in _TORCH_TO_DT:
            raise TypeError("unsupported dtype %s for %s" % (t.dtype, name))
        arr = t.view(torch.uint8).numpy() if t.dtype == torch.bfloat16 else t.numpy()
        dims = (C.c_int64 * max(t.dim(), 1))(*t.shape)
        rc = _rt().dtf_bundle_writer_add(self._h, name.encode(), _TORCH_TO_DT[t.dtype], dims, t.dim(),
                                         arr.ctypes.data_as(C.c_void_p), t.numel() * t.element_size(), shard)
        if rc != 0:
            raise IOError("bundle add failed: " + _err())

    def finish(self):
        h, self._h = self._h, None
        if h and _rt().dtf_bundle_writer_finish(h) != 0:
            raise IOError("bundle finish failed: " + _err())

    def abort(self):
        h, self._h = self._h, None
        if h:
            _rt().dtf_bundle_writer_abort(h)


def merge_shards(prefix: str, num_shards: int):
    """Chief side of a sharded save: merge ``<prefix>.index.part-<i>`` into ``<prefix>.index``."""
    if _rt().dtf_bundle_merge(prefix.encode(), num_shards) != 0:
        raise IOError("bundle merge failed: " + _err())


class BundleReader:
    def __init__(self, prefix: str):
        self.prefix = prefix
        self._h = _rt().dtf_bundle_reader_open(prefix.encode())
        if not self._h:
            raise IOError("cannot open checkpoint %s: %s" % (prefix, _err()))
        self.entries: "OrderedDict[str, Dict]" = OrderedDict()
        n = _rt().dtf_bundle_reader_num_entries(self._h)
        name = C.create_string_buffer(4096)
        dims = (C.c_int64 * 16)()
        dtype, ndims, shard = C.c_int(), C.c_int(), C.c_int()
        nbytes, offset = C.c_int64(), C.c_int64()
        crc = C.c_uint32()
        for i in range(n):
            rc = _rt().dtf_bundle_reader_entry(self._h, i, name, 4096, C.byref(dtype), dims, 16, C.byref(ndims),
                                               C.byref(nbytes), C.byref(shard), C.byref(offset), C.byref(crc))
            if rc != 0:
                raise IOError("bad checkpoint entry %d" % i)
            self.entries[name.value.decode()] = {
                "dtype": dtype.value, "shape": tuple(dims[d] for d in range(ndims.value)), "nbytes": nbytes.value,
                "shard": shard.value, "offset": offset.value, "crc32c": crc.value}
        self.num_shards = _rt().dtf_bundle_reader_num_shards(self._h)

    def has_tensor(self, name: str) -> bool:
        return name in self.entries

    def get_variable_to_shape_map(self) -> Dict[str, Tuple[int, ...]]:
        return {k: v["shape"] for k, v in self.entries.items()}

    def get_tensor(self, name: str, verify_crc: bool = True) -> torch.Tensor:
        e = self.entries[name]
        if e["dtype"] not in _DT_TO_TORCH:
            raise TypeError("unsupported checkpoint dtype %d for %s" % (e["dtype"], name))
        dt = _DT_TO_TORCH[e["dtype"]]
        buf = np.empty(max(e["nbytes"], 1), dtype=np.uint8)
        rc = _rt().dtf_bundle_reader_read(self._h, name.encode(), buf.ctypes.data_as(C.c_void_p), e["nbytes"],
                                          int(verify_crc))
        if rc != 0:
            raise IOError("cannot read %s: %s" % (name, _err()))
        t = torch.from_numpy(buf[:e["nbytes"]]).view(dt)
        return t.reshape(e["shape"]).clone()

    def close(self):
        h, self._h = self._h, None
        if h:
            _rt().dtf_bundle_reader_close(h)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def load_checkpoint(prefix: str) -> BundleReader:
    """``tf.train.load_checkpoint`` analogue."""
    return BundleReader(prefix)


def list_variables(prefix: str) -> List[Tuple[str, Tuple[int, ...]]]:
    """``tf.train.list_variables`` analogue: sorted (name, shape) pairs."""
    with BundleReader(prefix) as r:
        return sorted((k, v["shape"]) for k, v in r.entries.items())


# --------------------------------------------------------------------------- checkpoint state file
_STATE_FILE = "checkpoint"


def _read_state(directory: str) -> Tuple[Optional[str], List[str]]:
    path = os.path.join(directory, _STATE_FILE)
    if not os.path.exists(path):
        return None, []
    latest, all_paths = None, []
    with open(path, "r") as f:
        for line in f:
            m = re.match(r'\s*(model_checkpoint_path|all_model_checkpoint_paths)\s*:\s*"(.*)"\s*$', line)
            if not m:
                continue
            if m.group(1) == "model_checkpoint_path":
                latest = m.group(2)
            else:
                all_paths.append(m.group(2))
    return latest, all_paths


def _write_state(directory: str, latest: str, all_paths: List[str]):
    tmp = os.path.join(directory, _STATE_FILE + ".tmp%d" % os.getpid())
    with open(tmp, "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % latest)
        for p in all_paths:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)
    os.replace(tmp, os.path.join(directory, _STATE_FILE))


def latest_checkpoint(checkpoint_dir: str) -> Optional[str]:
    """Prefix of the newest checkpoint recorded in ``<dir>/checkpoint`` (None if there is none)."""
    latest, _ = _read_state(checkpoint_dir)
    if latest is None:
        return None
    prefix = latest if os.path.isabs(latest) else os.path.join(checkpoint_dir, latest)
    return prefix if os.path.exists(prefix + ".index") else None


def checkpoint_exists(prefix: str) -> bool:
    return os.path.exists(prefix + ".index")


def _delete_checkpoint(prefix: str):
    d, base = os.path.dirname(prefix), os.path.basename(prefix)
    for fn in os.listdir(d or "."):
        if fn == base + ".index" or fn == base + ".meta" or fn.startswith(base + ".data-") \
                or fn.startswith(base + ".index.part-"):
            try:
                os.remove(os.path.join(d, fn))
            except OSError:
                pass


# --------------------------------------------------------------------------- Saver
class Saver:
    """Saves / restores a ``VariableStore`` (all variables, ``global_step`` and optimizer slots)."""

    def __init__(self, store=None, optimizer=None, max_to_keep: int = 5, sharded: bool = True,
                 placement: Optional[Dict[str, int]] = None, num_shards: int = 1):
        self.store = store
        self.optimizer = optimizer
        self.max_to_keep = max_to_keep
        self.sharded = sharded
        self.placement = placement or {}
        self.num_shards = max(1, num_shards)
        self.last_checkpoints: List[str] = []

    # name -> tensor for everything persistent
    def _named_tensors(self) -> "OrderedDict[str, torch.Tensor]":
        st = self.store
        slot_names = getattr(self.optimizer, "slot_checkpoint_names", {}) if self.optimizer is not None else {}
        out = st.named_tensors(include_slots=True, slot_names=slot_names)
        if self.optimizer is not None and getattr(self.optimizer, "name", "") == "Adam":
            out["beta1_power"] = st.beta_powers[0]
            out["beta2_power"] = st.beta_powers[1]
        return out

    def _shard_of(self, name: str) -> int:
        if name in self.placement:
            return self.placement[name] % self.num_shards
        suffixes = set(getattr(self.optimizer, "slot_checkpoint_names", {}).values()) | {"Adam", "Adam_1", "Momentum"}
        base = name.rsplit("/", 1)[0] if name.rsplit("/", 1)[-1] in suffixes else name
        return self.placement.get(base, 0) % self.num_shards

    def save(self, save_path: str, global_step: Optional[int] = None, write_state: bool = True,
             only_shard: int = -1, merge: bool = True) -> str:
        """Write ``<save_path>-<global_step>``; returns the checkpoint prefix.

        ``only_shard >= 0``: write only the tensors placed on that shard (called by each owner in
        a multi-process save); the chief then calls with ``merge=True`` after all owners finished.
        """
        prefix = save_path if global_step is None else "%s-%d" % (save_path, int(global_step))
        tensors = self._named_tensors()
        w = BundleWriter(prefix, self.num_shards, only_shard)
        try:
            for name, t in tensors.items():
                shard = self._shard_of(name) if self.sharded else 0
                if only_shard >= 0 and shard != only_shard:
                    continue
                w.add(name, t, shard)
            w.finish()
        except Exception:
            w.abort()
            raise
        if only_shard >= 0:
            return prefix
        if write_state:
            self._record(prefix)
        return prefix

    def finish_sharded_save(self, prefix: str, write_state: bool = True) -> str:
        merge_shards(prefix, self.num_shards)
        if write_state:
            self._record(prefix)
        return prefix

    def _record(self, prefix: str):
        d = os.path.dirname(os.path.abspath(prefix))
        base = os.path.basename(prefix)
        _, all_paths = _read_state(d)
        all_paths = [p for p in all_paths if p != base] + [base]
        while self.max_to_keep and len(all_paths) > self.max_to_keep:
            victim = all_paths.pop(0)
            _delete_checkpoint(os.path.join(d, victim))
        _write_state(d, base, all_paths)
        self.last_checkpoints = [os.path.join(d, p) for p in all_paths]

    def restore(self, prefix: str, strict: bool = True) -> int:
        """Load every tensor found in the checkpoint into the store; returns the restored global_step."""
        st = self.store
        tensors = self._named_tensors()
        with BundleReader(prefix) as r:
            missing = [n for n in tensors if not r.has_tensor(n)]
            if strict and missing:
                raise KeyError("checkpoint %s lacks %s" % (prefix, ", ".join(missing[:8])))
            for name, dst in tensors.items():
                if not r.has_tensor(name):
                    continue
                src = r.get_tensor(name)
                if tuple(src.shape) != tuple(dst.shape):
                    raise ValueError("shape mismatch for %s: checkpoint %s vs variable %s"
                                     % (name, tuple(src.shape), tuple(dst.shape)))
                dst.copy_(src.to(dst.dtype))
        return int(st.global_step.item())

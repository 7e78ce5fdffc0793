"""ctypes bindings to the in-tree native libraries.

``kernels()`` returns ``libdtf_kernels.so`` (HIP, gfx950) and ``runtime()`` returns
``libdtf_runtime.so`` (host C++).  Both are loaded lazily.  A CUDA/HIP tensor reaching an
op without the kernel library present is an error (never a silent PyTorch fallback):
the CPU reference implementations in this package only run for CPU tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch

LIB_DIR = Path(__file__).resolve().parent.parent / "_lib"

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3
ACT_CODES = {None: ACT_NONE, "linear": ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "sigmoid": ACT_SIGMOID,
             "tanh": ACT_TANH}

_kernels = None
_runtime = None

c_f = C.c_float
c_i = C.c_int
c_u32 = C.c_uint32
c_ll = C.c_longlong
c_p = C.c_void_p


class NativeLibraryError(RuntimeError):
    pass


def _load(name: str):
    path = LIB_DIR / name
    if not path.exists():
        if os.environ.get("DTF_AUTO_BUILD", "1") == "1":
            from .. import _build
            (_build.build_kernels if "kernels" in name else _build.build_runtime)(verbose=False)
    if not path.exists():
        raise NativeLibraryError(
            "%s is missing: run `python -m distributed_tensorflow_amd._build` (needs hipcc / g++)" % path)
    return C.CDLL(str(path))


def kernels():
    """The HIP kernel library; argument types are declared once here."""
    global _kernels
    if _kernels is not None:
        return _kernels
    lib = _load("libdtf_kernels.so")
    sig = {
        "dtf_dense_fwd": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_u32, c_p, c_p],
        "dtf_dense_fwd_ws": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_u32, c_p, c_p, c_ll, c_p],
        "dtf_dense_fwd_splits": [c_i, c_i, c_i],
        "dtf_dense_bf16_supported": [c_i, c_i, c_i],
        "dtf_gemm_set_split_products": [c_i],
        "dtf_gemm_get_split_products": [],
        "dtf_dense_fwd_bf16": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_u32, c_p, c_p],
        "dtf_dense_bwd_dx_bf16": [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_f, c_p],
        "dtf_dense_bwd_dw_bf16": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
        "dtf_dense_bwd_dx": [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_f, c_p],
        "dtf_dense_bwd_dw_splits": [c_i, c_i, c_i],
        "dtf_conv_wgrad_splits": [c_i, c_i, c_i],
        "dtf_dense_bwd_dw": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p],
        "dtf_mse_loss": [c_p, c_p, c_p, c_i, c_i, c_f, c_i, c_p, c_p],
        "dtf_bce_loss": [c_p, c_p, c_p, c_i, c_i, c_f, c_i, c_p, c_p],
        "dtf_softmax_xent": [c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p, c_p],
        "dtf_categorical_accuracy": [c_p, c_p, c_i, c_i, c_p, c_p],
        "dtf_adam_apply": [c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_p, c_p, c_p],
        "dtf_sgd_apply": [c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_p, c_p, c_p],
        "dtf_rmsprop_apply": [c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_f, c_i, c_f, c_p, c_p, c_p],
        "dtf_adagrad_apply": [c_p, c_p, c_p, c_ll, c_f, c_f, c_i, c_f, c_p, c_p, c_p],
        "dtf_adadelta_apply": [c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_f, c_p, c_p, c_p],
        "dtf_step_advance": [c_p, c_p, c_f, c_f, c_p],
        "dtf_act_grad": [c_p, c_p, c_p, c_ll, c_i, c_f, c_p],
        "dtf_dropout": [c_p, c_p, c_ll, c_f, c_u32, c_p, c_p],
        "dtf_xor_data": [c_p, c_p, c_i, c_i, c_u32, c_p],
        "dtf_scale": [c_p, c_ll, c_f, c_p],
        "dtf_init_uniform": [c_p, c_ll, c_u32, c_f, c_f, c_p],
        "dtf_activation": [c_p, c_p, c_ll, c_i, c_p],
        "dtf_batchnorm_fwd": [c_p, c_p, c_ll, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_p, c_p, c_p, c_p],
        "dtf_batchnorm_bwd": [c_p, c_p, c_p, c_ll, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
        "dtf_softmax_rows": [c_p, c_p, c_i, c_i, c_p],
        "dtf_im2col": [c_p, c_p] + [c_i] * 12 + [c_p],
        "dtf_col2im": [c_p, c_p, c_p] + [c_i] * 12 + [c_p],
        "dtf_avgpool_fwd": [c_p, c_p] + [c_i] * 12 + [c_p],
        "dtf_avgpool_bwd": [c_p, c_p] + [c_i] * 12 + [c_p],
        "dtf_maxpool_fwd": [c_p, c_p, c_p] + [c_i] * 6 + [c_p],
        "dtf_maxpool_strided_fwd": [c_p, c_p, c_p] + [c_i] * 12 + [c_p],
        "dtf_maxpool_strided_bwd": [c_p, c_p, c_p, c_p] + [c_i] * 12 + [c_p],
        "dtf_conv_pool_supported": [c_i] * 4,
        "dtf_conv_pool_workspace_floats": [c_i] * 4,
        "dtf_conv_pool_set_staged": [c_i],
        "dtf_conv_pool_staged_fits": [c_i] * 11,
        "dtf_conv_pool_fwd": [c_p] * 5 + [c_i] * 12 + [c_p],
        "dtf_conv_pool_wgrad": [c_p] * 6 + [c_i] * 12 + [c_p, c_p],
        "dtf_maxpool_bwd": [c_p, c_p, c_p, c_p] + [c_i] * 7 + [c_p],
        "dtf_conv_pad": [c_p, c_p] + [c_i] * 8 + [c_p],
        "dtf_conv_rowtab": [c_p] + [c_i] * 8 + [c_p],
        "dtf_conv_flip_weights": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
        "dtf_conv_rowtab_masked": [c_p, c_p] + [c_i] * 13 + [c_p],
        "dtf_conv_gemm_fwd": [c_p, c_ll, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
        "dtf_conv_gemm_dgrad": [c_p, c_ll, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p],
        "dtf_conv_gemm_wgrad": [c_p, c_ll, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p],
        "dtf_mlp_tail_supported": [c_i, c_i, c_i],
        "dtf_mlp_tail": [c_p] * 12 + [c_ll, c_i, c_i, c_i, c_i, c_f, c_u32, c_p, c_f, c_f, c_p, c_p],
        "dtf_mlp3_fused_supported": [c_i, c_i, c_i, c_i, c_i],
        "dtf_mlp3_fused_train": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                 c_f, c_f, c_u32, c_u32, c_f, c_f, c_f, c_f, c_p, c_p, c_p, c_p],
    }
    for name, argtypes in sig.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = c_i
    if hasattr(lib, "dtf_batchnorm_workspace_floats"):
        lib.dtf_batchnorm_workspace_floats.argtypes = [c_ll, c_i]
        lib.dtf_batchnorm_workspace_floats.restype = c_ll
    _kernels = lib
    return lib


def declare(lib, name, argtypes, restype=c_i):
    fn = getattr(lib, name)
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


def runtime():
    """The host C++ runtime library (checkpoints, event files, control plane)."""
    global _runtime
    if _runtime is None:
        _runtime = _load("libdtf_runtime.so")
    return _runtime


def kernels_available() -> bool:
    try:
        kernels()
        return True
    except (NativeLibraryError, OSError):
        return False


def stream_ptr() -> int:
    """Raw hipStream_t of the current PyTorch stream (the capture stream under graph capture).  Every native op asks
    for it: the binding's own entry point is used, `torch.cuda.current_stream()` builds a Python stream object and
    resolves the device through several layers, ~3 us per call."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError("%s failed with HIP/launch status %d" % (what, rc))


def on_gpu(t: torch.Tensor) -> bool:
    return t.is_cuda


def require_f32_contig(*tensors):
    for t in tensors:
        if t is None:
            continue
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("native kernels take contiguous float32 tensors, got %s %s" % (t.dtype, tuple(t.stride())))

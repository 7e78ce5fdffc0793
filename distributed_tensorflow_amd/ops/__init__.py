"""Hot-path ops: ctypes bindings to the HIP kernels plus CPU reference implementations."""
from . import functional  # noqa: F401
from ._native import kernels_available  # noqa: F401

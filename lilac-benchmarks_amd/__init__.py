"""hip-spmv: an MI355X-native backend for the lilac-benchmarks `libspmv` C-ABI.

The product is the shared library `lib/libhip-spmv.so` (C host code + HIP kernels, see
`csrc/`); this Python package only builds it and binds it with ctypes for tests and
measurement.  It deliberately contains no compute path of its own and never touches
`oracle/`: if the library is missing or no GPU is present, calls fail loudly.
"""
from .build import build_all, LIB_DIR, lib_path  # noqa: F401
from .abi import Backend, Matrix, Info, Stats, PLAN_AUTO, PLAN_ORDERED, PLAN_STREAM, PLAN_VECTOR, PLAN_BINNED, PLAN_TILED, PLAN_STRIPED, PLAN_BLOCKED, PLAN_HUB  # noqa: F401
from . import npb  # noqa: F401
from . import callers  # noqa: F401

__all__ = [
    "build_all", "LIB_DIR", "lib_path", "Backend", "Matrix", "Info", "Stats", "npb", "callers",
    "PLAN_AUTO", "PLAN_ORDERED", "PLAN_STREAM", "PLAN_VECTOR", "PLAN_BINNED", "PLAN_TILED", "PLAN_STRIPED",
]

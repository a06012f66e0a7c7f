"""big-linear-algebra on MI355X: hand-written gfx950 HIP kernels behind a C-ABI
(include/bla.h), a C host layer that is API-identical to the reference's
lib/matrix.h / conv.h / norm.h / util.h / layer.h, and this thin ctypes binding
used by the tests and bench.py.  There is no CPU compute path in this package:
if libbla_hip.so or a gfx950 device is missing, calls raise.
"""
from . import build as _build  # noqa: F401
from .native import (BlaError, DeviceArray, lib, init, is_available, gemm, Epilogue,  # noqa: F401
                     to_device, empty, zeros, sync, ACT_NONE, ACT_RELU)

from . import native, mnist_nn  # noqa: F401,E402

build_native = _build.build_native
__all__ = ["BlaError", "DeviceArray", "lib", "init", "is_available", "gemm", "Epilogue", "to_device", "empty",
           "zeros", "sync", "build_native", "ACT_NONE", "ACT_RELU"]

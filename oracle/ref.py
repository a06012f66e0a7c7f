"""ctypes binding of the REFERENCE ITSELF (oracle/_ref/libref.so = /root/reference/lib/
{matrix,conv,norm,util,csv}.c compiled as-is by oracle/Makefile).

Checker only: used by oracle/gen_golden.py (to produce tests/golden/) and by
tests/test_oracle_vs_ref.py (to pin the restatement).  `available()` is False on
machines where neither the reference nor a prebuilt _ref/ exists.
The reference is fp64 (`typedef double matrix_float_t`, lib/matrix.h:4).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libref.so")
_lib = None


class Matrix(C.Structure):
    """struct Matrix, lib/matrix.h:6-11."""
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("data", C.POINTER(C.c_double))]


PM = C.POINTER(Matrix)


def available():
    return os.path.exists(_PATH)


def lib(opt="O2"):
    global _lib
    if opt != "O2":
        return C.CDLL(os.path.join(_HERE, "_ref", "libref_O0.so"))
    if _lib is None:
        L = C.CDLL(_PATH)
        L.matrix_multiply.restype = PM
        L.matrix_multiply.argtypes = [Matrix, Matrix]
        L.matrix_row_sum.restype = PM
        L.matrix_row_sum.argtypes = [Matrix]
        L.matrix_col_sum.restype = PM
        L.matrix_col_sum.argtypes = [Matrix]
        L.clone_matrix.restype = PM
        L.clone_matrix.argtypes = [Matrix]
        L.frobenius_norm.restype = C.c_double
        L.frobenius_norm.argtypes = [Matrix]
        L.max_value.restype = C.c_double
        L.max_value.argtypes = [Matrix]
        L.matrix_scale.argtypes = [PM, C.c_double]
        L.free_matrix.argtypes = [PM]
        _lib = L
    return _lib


def mat(a):
    """Wrap a 2-D float64 numpy array (kept alive by the caller) as a struct Matrix."""
    assert a.dtype == np.float64 and a.flags.c_contiguous and a.ndim == 2
    return Matrix(a.shape[0], a.shape[1], a.ctypes.data_as(C.POINTER(C.c_double)))


def mats(a3):
    """[C][H][W] array -> C array of struct Matrix sharing its memory (conv.c's `Matrix* in`)."""
    assert a3.dtype == np.float64 and a3.flags.c_contiguous and a3.ndim == 3
    arr = (Matrix * a3.shape[0])()
    for c in range(a3.shape[0]):
        arr[c] = mat(a3[c])
    return arr


def kernel_ptrs(k4):
    """[F][C][k][k] array -> Matrix** as conv.c's `Matrix** kernels` (kernels[f][c])."""
    rows = [mats(k4[f]) for f in range(k4.shape[0])]
    pp = (PM * k4.shape[0])(*[C.cast(r, PM) for r in rows])
    return pp, rows


def take(pm):
    """Copy a library-allocated Matrix* into numpy and free it."""
    m = pm.contents
    out = np.ctypeslib.as_array(m.data, shape=(m.rows, m.cols)).copy()
    lib().free_matrix(pm)
    return out


# ---- convenience wrappers (float64 in/out, inputs never modified) ------------
def matmul(a, b):
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    return take(lib().matrix_multiply(mat(a), mat(b)))


def matmul_inplace(a, b):
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    c = np.zeros((a.shape[0], b.shape[1]))
    ma, mb, mc = mat(a), mat(b), mat(c)
    lib().matrix_multiply_inplace(C.byref(ma), C.byref(mb), C.byref(mc))
    return c


def inplace1(fn, a, *extra):
    a = np.ascontiguousarray(a, np.float64).copy()
    m = mat(a)
    getattr(lib(), fn)(C.byref(m), *extra)
    return a.reshape(m.rows, m.cols)


def inplace2(fn, a, b):
    a = np.ascontiguousarray(a, np.float64).copy()
    b = np.ascontiguousarray(b, np.float64)
    ma, mb = mat(a), mat(b)
    getattr(lib(), fn)(C.byref(ma), C.byref(mb))
    return a


def data_fn(fn, a, *dims):
    a = np.ascontiguousarray(a, np.float64).copy()
    getattr(lib(), fn)(a.ctypes.data_as(C.POINTER(C.c_double)), *dims)
    return a

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


# ---- model/mnist_nn.c:218-315 driven call by call through the reference's matrix.h ----------------------------
def mnist_step(params, x_raw, y, intended_colsum):
    """One SGD step of the reference's trainer: every matrix.h call of model/mnist_nn.c:218-315 in its order, on the reference's own
    objects (relu / softmax are lib/util.c's, identical to the model-local copies :38-73).  Returns (new params, activations, gradients).
    oracle/gen_golden.py records it as tests/golden/mnist_step.npz; bench.py times it as the secondary workload's cpu_baseline."""
    L = lib()
    w1, b1, w2, b2, w3, b3 = [np.ascontiguousarray(p, np.float64).copy() for p in params]
    B = x_raw.shape[1]
    n0 = w1.shape[1]
    x = inplace1("matrix_scale", x_raw, C.c_double(np.float32(1) / np.float32(255.0)))   # :218 (1/255.0F)

    def fwd(w, a, b):
        z = matmul(w, a)                                    # matrix_multiply
        z = inplace2("matrix_add_tile_columns", z, b)
        return z

    def colsum(m):
        if intended_colsum:   # true row sums via the reference's own row_sum of the transpose
            mt = np.ascontiguousarray(inplace1("matrix_transpose", m))   # keep alive across the call
            return take(L.matrix_row_sum(mat(mt))).reshape(-1, 1)
        assert m.shape[0] <= m.shape[1]
        mc = np.ascontiguousarray(m)
        return take(L.matrix_col_sum(mat(mc)))

    z1 = fwd(w1, x, b1); a1 = data_fn("relu", z1, z1.size)
    z2 = fwd(w2, a1, b2); a2 = data_fn("relu", z2, z2.size)
    z3 = fwd(w3, a2, b3); a3 = data_fn("softmax", z3, z3.shape[0], B)
    scale = 1 / float(n0)                                        # :260
    ny = inplace1("matrix_scale", y, C.c_double(-1.0))
    dz3 = inplace2("matrix_add", a3, ny)
    dz3 = inplace1("matrix_scale", dz3, C.c_double(scale))
    dw3 = matmul(dz3, inplace1("matrix_transpose", a2)); db3 = colsum(dz3)
    da2 = matmul(inplace1("matrix_transpose", w3), dz3)
    dz2 = inplace2("matrix_multiply_elementwise", (z2 > 0).astype(np.float64), da2)
    dw2 = matmul(dz2, inplace1("matrix_transpose", a1)); db2 = colsum(dz2)
    da1 = matmul(inplace1("matrix_transpose", w2), dz2)
    dz1 = inplace2("matrix_multiply_elementwise", (z1 > 0).astype(np.float64), da1)
    dw1 = matmul(dz1, inplace1("matrix_transpose", x)); db1 = colsum(dz1)
    grads = [dw1, db1, dw2, db2, dw3, db3]
    lr = float(np.float32(-0.02))                                # float epoch_learn_rate, :186
    new = []
    for p, g in zip([w1, b1, w2, b2, w3, b3], grads):
        new.append(inplace2("matrix_add", p, inplace1("matrix_scale", g, C.c_double(lr))))
    return new, dict(z1=z1, a1=a1, z2=z2, a2=a2, z3=z3, a3=a3), grads


# ---- lib/conv.c:205-229, the reference's own stages in the intended order (stride 1) ---------------------------
def conv_fwd_bwd(x, kern, del_y):
    """conv() + conv_ddx() of one image, stride 1, by the reference's own functions called in the order lib/conv.c:205-229 calls them, with
    the two direction-swapped reshapes given the operands that make the composition the intended one (the same driving
    oracle/gen_golden.py uses for tests/golden/conv.npz).  x [C][H][W], kern [F][C][k][k], del_y [F][H][W], float64.
    Returns (output [F][H][W], del_kern [F][C][k][k], del_x [C][H][W])."""
    L = lib()
    x = np.ascontiguousarray(x, np.float64); kern = np.ascontiguousarray(kern, np.float64); del_y = np.ascontiguousarray(del_y, np.float64)
    cin, h, w = x.shape; cout, _, k, _ = kern.shape
    xm = mats(x); kp, _keep = kernel_ptrs(kern)
    im = np.zeros((h * w, k * k * cin)); imm = mat(im)
    L._im2col(xm, C.byref(imm), k, cin, 1)                                # :208
    km = np.zeros((k * k * cin, cout)); kmm = mat(km)
    L._reshape_kernels_matrix(kp, C.byref(kmm))                           # :209
    prod = matmul_inplace(im, km)                                         # :210
    out = np.zeros((cout, h, w)); om = mats(out); pm = mat(prod)
    L.reshape_channels_matrix(om, C.byref(pm))                            # :211 (as written this call performs channels <- matrix)
    dq = np.zeros((h * w, cout)); dqm = mat(dq)
    L.reshape_matrix_channels(C.byref(dqm), mats(del_y))                  # :219 del_Q <- del_Y
    imt = inplace1("matrix_transpose", im)                                # :221
    dkm = matmul_inplace(imt, dq)                                         # :222
    dkern = np.zeros_like(kern); dkp, _k3 = kernel_ptrs(dkern); dkmm = mat(dkm)
    L._reshape_matrix_kernels(C.byref(dkmm), dkp)                         # :223
    kmt = inplace1("matrix_transpose", km)                                # :225
    dcol = matmul_inplace(dq, kmt)                                        # :226
    dx = np.zeros((cin, h, w)); dcm = mat(dcol)
    L._col2im(C.byref(dcm), mats(dx), k, cin, 1)                          # :228
    return out, dkern, dx

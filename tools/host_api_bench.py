#!/usr/bin/env python3
"""SURVEY 8(d) cfg 2 (ii): through-API rate of the drop-in `matrix_multiply_inplace` (big-linear-algebra_amd/lib/matrix.c,
same C signature as lib/matrix.c:47-57): pageable host operands in, host result out, i.e. including the H2D / D2H copies and the
synchronisation the reference's stateless contract forces on every call.  Beside it: the device-resident kernel rate."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_pkg
from inputs import uniform

bla = load_pkg(); bla.build_native(); bla.init(0)
LIB = os.path.join(ROOT, "big-linear-algebra_amd", "lib", "libbla_host.so")
H = C.CDLL(LIB)


class Matrix(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("data", C.POINTER(C.c_float))]


def mat(a):
    return Matrix(a.shape[0], a.shape[1], a.ctypes.data_as(C.POINTER(C.c_float)))


H.matrix_multiply_inplace.argtypes = [C.POINTER(Matrix)] * 3
H.matrix_multiply_inplace.restype = None
for n in [int(x) for x in (sys.argv[1:] or ["1024", "2048", "4096", "8192"])]:
    a = uniform(0xB1A5, (n, n), dtype=np.float32); b = uniform(0xB1A6, (n, n), dtype=np.float32); c = np.empty((n, n), np.float32)
    ma, mb, mc = mat(a), mat(b), mat(c)
    H.matrix_multiply_inplace(C.byref(ma), C.byref(mb), C.byref(mc))   # warm-up (staging buffers, first-touch)
    reps = 5 if n <= 4096 else 2
    t0 = time.perf_counter()
    for _ in range(reps):
        H.matrix_multiply_inplace(C.byref(ma), C.byref(mb), C.byref(mc))
    t = (time.perf_counter() - t0) / reps
    da, db, dc = bla.to_device(a), bla.to_device(b), bla.empty((n, n))
    L = bla.lib(); st = L.bla_default_stream()
    for _ in range(3):
        bla.gemm(da, db, dc, stream=st)
    bla.native.sync(st)
    t1 = time.perf_counter()
    for _ in range(10):
        bla.gemm(da, db, dc, stream=st)
    bla.native.sync(st)
    tk = (time.perf_counter() - t1) / 10
    ok = np.allclose(c[:4], dc.numpy()[:4], rtol=1e-4, atol=1e-3)
    fl = 2.0 * n ** 3
    print(f"N={n:5d}  through the host API (pageable in/out, 3 x {n*n*4/1e6:6.1f} MB over PCIe) {t*1e3:9.2f} ms = {fl/t/1e9:10.1f} GFLOP/s   |   "
          f"device-resident kernel {tk*1e3:8.3f} ms = {fl/tk/1e9:10.1f} GFLOP/s   (results agree: {ok})", flush=True)

"""The -DBLA_FP64 build (SURVEY 7.0(1), 8(a) `matrix_float_t`): the matrix.h functions in the reference's own element type, fp64 on the
device (MFMA f64 GEMM), against the golden vectors the reference produced -- to 1e-12 normwise (only the order of additions differs).
CPU part: the reference's mnist_nn.c compiles and links against this repo's matrix.h / csv.h / mnist_csv2.h with the double typedef."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden
from inputs import uniform

LIB = os.path.join(ROOT, "big-linear-algebra_amd", "lib")
CSRC = os.path.join(ROOT, "big-linear-algebra_amd", "csrc")
REF = "/root/reference"
TOL = 1e-12


class Matrix(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("data", C.POINTER(C.c_double))]


PM = C.POINTER(Matrix)


def mat(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return Matrix(a.shape[0], a.shape[1], a.ctypes.data_as(C.POINTER(C.c_double)))


def close(got, want, tol=TOL):
    return np.linalg.norm(np.asarray(got, np.float64) - want) <= tol * max(np.linalg.norm(want), 1e-300)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference sources only exist in the build container")
def test_reference_mnist_program_links_with_double_typedef(tmp_path, pkg):
    """model/mnist_nn.c, unchanged, against this repo's headers under -DBLA_FP64 (typedef double matrix_float_t, as in the reference)."""
    pkg.build_native()
    tree = tmp_path / "tree"; (tree / "model").mkdir(parents=True); (tree / "lib").mkdir()
    for f in os.listdir(LIB):
        if f.endswith((".h", ".c")):
            os.symlink(os.path.join(LIB, f), tree / "lib" / f)
    os.symlink(os.path.join(REF, "model", "mnist_nn.c"), tree / "model" / "mnist_nn.c")
    r = subprocess.run(["gcc", "-std=c99", "-DBLA_FP64", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-sign-compare", "-Wno-unused-variable",
                        "-Wno-unused-but-set-variable", "-Werror=incompatible-pointer-types", "-I", os.path.join(ROOT, "include"), "model/mnist_nn.c", "lib/matrix.c",
                        "lib/csv.c", "lib/mnist_csv2.c", "lib/bla_host.c", "-o", str(tmp_path / "prog"), "-L", CSRC, "-l:libbla_hip.so", f"-Wl,-rpath,{CSRC}", "-lm"],
                       cwd=str(tree), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "bla_gemm_f64" in subprocess.run(["nm", "-D", str(tmp_path / "prog")], stdout=subprocess.PIPE, text=True).stdout


@pytest.fixture(scope="module")
def dev(pkg):
    pkg.init(0)
    return pkg


def d64(dev, a):
    return dev.DeviceArray(np.asarray(a).shape, np.float64).copy_from(np.ascontiguousarray(a, np.float64))


@pytest.mark.gpu
def test_gemm_f64_against_the_reference_vectors(dev):
    g = golden("gemm"); L = dev.lib(); chk = dev.native.check
    cases = [(g["kat_main_a"], g["kat_main_b"], g["kat_main_c"]), (g["csv_a"], g["csv_b"], g["csv_c"])]
    for i, (m, k, n) in enumerate(g["shapes"]):
        cases.append((uniform(100 + i, (int(m), int(k))), uniform(200 + i, (int(k), int(n))), None if g.is_digest(f"rand{i}_c") else g[f"rand{i}_c"]))
    for i, (a, b, want) in enumerate(cases):
        da, db = d64(dev, a), d64(dev, b); dc = dev.DeviceArray((a.shape[0], b.shape[1]), np.float64)
        chk(L.bla_gemm_f64(None, 0, 0, a.shape[0], b.shape[1], a.shape[1], da.ptr, a.shape[1], db.ptr, b.shape[1], dc.ptr, b.shape[1], 1.0, 0.0))
        got = dc.numpy()
        if want is None:
            g.check(f"rand{i - 2}_c", got, rtol=0, atol=TOL * np.abs(a).sum(1).max() * np.abs(b).max())
        else:
            assert close(got, want), i
        # transposed operands and alpha / beta
        dat, dbt = d64(dev, a.T), d64(dev, b.T)
        chk(L.bla_gemm_f64(None, 1, 1, a.shape[0], b.shape[1], a.shape[1], dat.ptr, a.shape[0], dbt.ptr, a.shape[1], dc.ptr, b.shape[1], 0.5, 2.0))
        assert close(dc.numpy(), 2.5 * (a @ b), 1e-11)


@pytest.mark.gpu
def test_matrix_ops_f64_against_the_reference_vectors(dev):
    g = golden("matrix_ops"); L = dev.lib(); chk = dev.native.check
    for i, (r, c) in enumerate(g["shapes"]):
        r, c = int(r), int(c)
        a = uniform(300 + i, (r, c), -2, 2); b = uniform(400 + i, (r, c), -2, 2)
        def run(fn, *args):
            da = d64(dev, a); chk(fn(None, da.ptr, *args)); return da.numpy()
        db = d64(dev, b)
        assert close(run(L.bla_scale_f64, a.size, -0.37), g[f"s{i}_scale"]) and close(run(L.bla_add_f64, db.ptr, a.size), g[f"s{i}_add"])
        assert close(run(L.bla_hadamard_f64, db.ptr, a.size), g[f"s{i}_hadamard"])
        o = dev.DeviceArray((c, r), np.float64); chk(L.bla_transpose_f64(None, d64(dev, a).ptr, o.ptr, r, c))
        assert np.array_equal(o.numpy(), g[f"s{i}_transpose"])
        o = dev.DeviceArray((1, c), np.float64); chk(L.bla_row_sum_f64(None, d64(dev, a).ptr, r, c, o.ptr)); assert close(o.numpy(), g[f"s{i}_row_sum"])
        o = dev.DeviceArray((r, 1), np.float64)
        if r <= c:
            chk(L.bla_col_sum_f64(None, d64(dev, a).ptr, r, c, o.ptr, 0)); assert close(o.numpy(), g[f"s{i}_col_sum"])
        else:
            assert L.bla_col_sum_f64(None, d64(dev, a).ptr, r, c, o.ptr, 0) == 5          # undefined in the reference (Q2)
        chk(L.bla_col_sum_f64(None, d64(dev, a).ptr, r, c, o.ptr, 1)); assert close(o.numpy().ravel(), a.sum(1))
        s = dev.DeviceArray((1,), np.float64)
        chk(L.bla_frobenius_f64(None, d64(dev, a).ptr, a.size, s.ptr)); assert abs(s.numpy()[0] - float(g[f"s{i}_frobenius"])) <= TOL * float(g[f"s{i}_frobenius"])
        chk(L.bla_max_f64(None, d64(dev, a).ptr, a.size, s.ptr)); assert s.numpy()[0] == float(g[f"s{i}_max"])
        assert close(run(L.bla_zscore_f64, a.size), g[f"s{i}_zscore"], 1e-11)
        bc = uniform(500 + i, (r, 1)); br = uniform(600 + i, (1, c))
        assert close(run(L.bla_add_tile_columns_f64, r, c, d64(dev, bc).ptr, 1), g[f"s{i}_tile_cols"])
        assert close(run(L.bla_add_tile_rows_f64, r, c, d64(dev, br).ptr), g[f"s{i}_tile_rows"])


@pytest.mark.gpu
def test_host_layer_in_double(dev):
    """lib/libbla_host_f64.so: the drop-in matrix.h with `double` data, through its C signatures."""
    H = C.CDLL(os.path.join(LIB, "libbla_host_f64.so"))
    H.matrix_multiply.restype = PM; H.matrix_multiply.argtypes = [Matrix, Matrix]
    H.frobenius_norm.restype = C.c_double; H.frobenius_norm.argtypes = [Matrix]
    H.matrix_scale.argtypes = [PM, C.c_double]; H.free_matrix.argtypes = [PM]
    g = golden("gemm")
    a, b = np.ascontiguousarray(g["kat_main_a"]), np.ascontiguousarray(g["kat_main_b"])
    pm = H.matrix_multiply(mat(a), mat(b))
    got = np.ctypeslib.as_array(pm.contents.data, shape=(2, 2)).copy(); H.free_matrix(pm)
    assert close(got, g["kat_main_c"]) and np.allclose(got, [[1.4, 8.5], [5.0, 19.0]], atol=1e-12)      # main.c:39-40
    x = uniform(9, (37, 53), -2, 2); want = np.sqrt((x * x).sum())
    assert abs(H.frobenius_norm(mat(x)) - want) <= 1e-12 * want
    y = x.copy(); m = mat(y); H.matrix_scale(C.byref(m), -0.37); assert np.array_equal(y, x * -0.37)
    m = mat(y); H.matrix_transpose(C.byref(m)); assert (m.rows, m.cols) == (53, 37) and np.array_equal(y.reshape(53, 37), (x * -0.37).T)

"""The drop-in host layer (big-linear-algebra_amd/lib: matrix.h, conv.h, norm.h, util.h, layer.h).

CPU part: the reference's own programs compile and link against it unchanged (only where /root/reference
exists), error behaviour matches the reference's (message + exit(1)), and with no GPU it fails loudly.
GPU part: every host entry point, called through its C signature with struct Matrix arguments, against
the golden vectors."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden
from inputs import uniform

LIB = os.path.join(ROOT, "big-linear-algebra_amd", "lib")
CSRC = os.path.join(ROOT, "big-linear-algebra_amd", "csrc")
REF = "/root/reference"
F32 = np.float32


class Matrix(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("data", C.POINTER(C.c_float))]


PM = C.POINTER(Matrix)


class ConvData(C.Structure):
    _fields_ = [("im2col", PM), ("kernel_matrix", PM), ("product", PM), ("output", PM)]


def mat(a):
    assert a.dtype == F32 and a.flags.c_contiguous
    a2 = a.reshape(a.shape[0], -1) if a.ndim != 2 else a
    return Matrix(a2.shape[0], a2.shape[1], a.ctypes.data_as(C.POINTER(C.c_float)))


def mats(a3):
    arr = (Matrix * a3.shape[0])()
    for c in range(a3.shape[0]):
        arr[c] = Matrix(a3.shape[1], a3.shape[2], a3[c].ctypes.data_as(C.POINTER(C.c_float)))
    return arr


def kernel_ptrs(k4):
    rows = [mats(k4[f]) for f in range(k4.shape[0])]
    return (PM * k4.shape[0])(*[C.cast(r, PM) for r in rows]), rows


@pytest.fixture(scope="module")
def host(pkg):
    pkg.build_native()
    L = C.CDLL(os.path.join(LIB, "libbla_host.so"))
    L.matrix_multiply.restype = PM; L.matrix_multiply.argtypes = [Matrix, Matrix]
    L.matrix_row_sum.restype = PM; L.matrix_row_sum.argtypes = [Matrix]
    L.matrix_col_sum.restype = PM; L.matrix_col_sum.argtypes = [Matrix]
    L.clone_matrix.restype = PM; L.clone_matrix.argtypes = [Matrix]
    L.frobenius_norm.restype = C.c_float; L.frobenius_norm.argtypes = [Matrix]
    L.max_value.restype = C.c_float; L.max_value.argtypes = [Matrix]
    L.matrix_scale.argtypes = [PM, C.c_float]
    L.free_matrix.argtypes = [PM]
    L.group_norm.argtypes = [PM, PM, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int]
    L.group_norm_ddx.argtypes = [PM, PM, PM, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int]
    return L


def take(L, pm):
    m = pm.contents
    out = np.ctypeslib.as_array(m.data, shape=(m.rows, m.cols)).copy()
    L.free_matrix(pm)
    return out


# ---------------------------------------------------------------- CPU ------------------------------------------
def _cc(args, **kw):
    return subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **kw)


def test_print_matrix_stdout_matches_the_reference(host):
    """print_matrix / print_matrix_dim (lib/matrix.c:71-93) write the same BYTES as the reference did for the same matrices
    (tests/golden/print_matrix.npz holds the reference's captured stdout): exact zeros, negatives (which take the `< 0.01` branch and
    print as %.2e), small positives, ordinary values, the 2 x 2 known answer of main.c:39-40.  Host-only: no device needed."""
    import tempfile
    g = golden("print_matrix")
    host.print_matrix.argtypes = [Matrix]; host.print_matrix_dim.argtypes = [Matrix]
    libc = C.CDLL(None)
    for name in ("zeros", "mixed", "main_kat", "column", "row"):
        a = np.ascontiguousarray(g[name], F32)
        assert np.array_equal(a.astype(np.float64), g[name])       # fp32-representable by construction
        sys.stdout.flush(); libc.fflush(None)
        with tempfile.TemporaryFile() as tmp:
            saved = os.dup(1); os.dup2(tmp.fileno(), 1)
            try:
                host.print_matrix(mat(a)); host.print_matrix_dim(mat(a)); libc.fflush(None)
            finally:
                os.dup2(saved, 1); os.close(saved)
            tmp.seek(0); got = tmp.read()
        assert got == bytes(g[name + "__stdout"].astype(np.uint8)), (name, got)


def test_exports_reference_symbols(host):
    """Every function the reference headers declare (lib/matrix.h:13-32, conv.h:13-16, norm.h:6-7, util.h:7-11,
    layer.h:17-21, csv.h) plus the four extern helpers of lib/conv.c is exported."""
    names = ("make_matrix clone_matrix free_matrix_data free_matrix matrix_multiply matrix_scale matrix_add print_matrix "
             "print_matrix_dim matrix_multiply_elementwise matrix_transpose matrix_row_sum matrix_col_sum frobenius_norm "
             "max_value matrix_z_score_normalize matrix_add_tile_columns matrix_add_tile_rows matrix_multiply_inplace "
             "conv reshape_channels_matrix reshape_matrix_channels conv_ddx _im2col _col2im _reshape_kernels_matrix "
             "_reshape_matrix_kernels group_norm group_norm_ddx relu softmax softmax_row_wise load_matrix_from_csv "
             "random_gaussian feed_forward free_layer_data load_weights_from_csv load_biases_from_csv back_propagate_errors "
             "read_csv_contents read_csv_contents_file write_csv_contents count_num_lines").split()
    for n in names:
        assert hasattr(host, n), n


def test_error_behaviour_matches_reference(tmp_path, host):
    exe = str(tmp_path / "host_errors")
    r = _cc(["gcc", "-std=c99", "-I", LIB, os.path.join(ROOT, "tests", "c", "host_errors.c"), "-o", exe,
             "-L", LIB, "-l:libbla_host.so", f"-Wl,-rpath,{LIB}", f"-Wl,-rpath,{CSRC}"])
    assert r.returncode == 0, r.stderr
    r = _cc([exe, "mul"])
    assert r.returncode == 1 and r.stdout == "Attempted to multiply 2x3 matrix by 2x3 matrix, exiting\n"        # lib/matrix.c:37
    r = _cc([exe, "had"])
    assert r.returncode == 1 and r.stdout == "Attempted to multiply elements of 2x3 matrix by 3x2 matrix, exiting\n"   # :97
    import torch
    if not torch.cuda.is_available():
        r = _cc([exe, "dev"])
        assert r.returncode == 1 and "scaled" not in r.stdout and "no CPU path" in r.stderr    # loud, no fallback


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference sources only exist in the build container")
@pytest.mark.parametrize("prog,extra_ref,ours", [
    ("model/mnist_nn.c", [], ["matrix.c", "csv.c", "mnist_csv2.c", "bla_host.c"]),
    ("model/cifar_unet.c", [], ["matrix.c", "conv.c", "norm.c", "util.c", "csv.c", "cifar10.c", "bmp.c", "bla_host.c"]),
    ("model/mnist_hinge.c", [], ["matrix.c", "layer.c", "csv.c", "mnist_csv.c", "bla_host.c"]),
    ("main.c", [], ["matrix.c", "layer.c", "csv.c", "bla_host.c"]),
    ("model/my_first_model.c", [], ["matrix.c", "layer.c", "csv.c", "bla_host.c"]),
])
def test_reference_programs_link_unchanged(tmp_path, pkg, prog, extra_ref, ours):
    """The model sources are compiled where they are, from a scratch tree that lays our lib/ next to them
    (they include "../lib/matrix.h").  All five link against this repo's units ONLY (mnist_hinge.c takes the legacy
    streaming reader lib/mnist_csv.c, which clashes with mnist_csv2.h by the reference's design and is therefore a
    per-program unit here as there)."""
    pkg.build_native()
    tree = tmp_path / "tree"
    (tree / "model").mkdir(parents=True)
    (tree / "lib").mkdir()
    for f in os.listdir(LIB):            # our drop-in layer ...
        if f.endswith((".h", ".c")):
            os.symlink(os.path.join(LIB, f), tree / "lib" / f)
    for f in os.listdir(os.path.join(REF, "lib")):   # ... plus the reference's out-of-scope IO units it does not replace
        if not (tree / "lib" / f).exists():
            os.symlink(os.path.join(REF, "lib", f), tree / "lib" / f)
    src = tree / prog
    os.symlink(os.path.join(REF, prog), src)
    objs = [str(src)] + [str(tree / "lib" / o) for o in ours] + [str(tree / e) for e in extra_ref]
    flags = ["-std=c99", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-sign-compare", "-Wno-unused-variable",
             "-Wno-unused-but-set-variable", "-Werror=incompatible-pointer-types", "-Werror=implicit-function-declaration"]
    exe = str(tmp_path / "prog")
    r = _cc(["gcc"] + flags + ["-I", os.path.join(ROOT, "include")] + objs + ["-o", exe, "-L", CSRC, "-l:libbla_hip.so",
             f"-Wl,-rpath,{CSRC}", "-lm"], cwd=str(tree))
    assert r.returncode == 0, r.stderr[-3000:]
    # the binary really is bound to the HIP backend, not to a CPU matrix.c
    nm = _cc(["nm", "-D", exe]).stdout
    assert "bla_gemm_f32" in nm


# ---------------------------------------------------------------- GPU ------------------------------------------
gpu = pytest.mark.gpu


def close(got, ref, rtol=2e-6, scale=0.0):
    ref = np.asarray(ref, np.float64); got = np.asarray(got, np.float64).reshape(ref.shape)
    s = np.maximum(np.abs(ref), scale)
    assert (np.abs(got - ref) <= rtol * s + 1e-30).all(), (np.abs(got - ref) / (s + 1e-300)).max()


@gpu
def test_matrix_h_against_golden(host, ora):
    g = golden("gemm")
    a, b = g["kat_main_a"].astype(F32), g["kat_main_b"].astype(F32)
    c = take(host, host.matrix_multiply(mat(a), mat(b)))
    assert np.allclose(c, [[1.4, 8.5], [5.0, 19.0]], rtol=1e-6)                      # main.c:20-41
    for i, (m, k, n) in enumerate(g["shapes"]):
        a = uniform(100 + i, (m, k), dtype=F32); b = uniform(200 + i, (k, n), dtype=F32)
        c = np.full((int(m), int(n)), np.nan, F32)
        ma, mb, mc = mat(a), mat(b), mat(c)
        host.matrix_multiply_inplace(C.byref(ma), C.byref(mb), C.byref(mc))
        bound = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))
        assert (np.abs(c - g[f"rand{i}_c"]) <= 1e-5 * bound + 1e-30).all()
    g = golden("matrix_ops")
    for i, (r, c) in enumerate(g["shapes"]):
        r, c = int(r), int(c)
        a = uniform(300 + i, (r, c), -2, 2, F32); b = uniform(400 + i, (r, c), -2, 2, F32)
        x = a.copy(); m = mat(x); host.matrix_scale(C.byref(m), -0.37); close(x, g[f"s{i}_scale"])
        x = a.copy(); m, mb = mat(x), mat(b); host.matrix_add(C.byref(m), C.byref(mb)); close(x, g[f"s{i}_add"], scale=2)
        x = a.copy(); m = mat(x); host.matrix_multiply_elementwise(C.byref(m), C.byref(mb)); close(x, g[f"s{i}_hadamard"])
        x = a.copy(); m = mat(x); host.matrix_transpose(C.byref(m))
        assert (m.rows, m.cols) == (c, r) and np.array_equal(x.reshape(c, r), a.T)
        close(take(host, host.matrix_row_sum(mat(a))), g[f"s{i}_row_sum"], scale=np.abs(a).sum(0, keepdims=True))
        cs = take(host, host.matrix_col_sum(mat(a)))
        if r <= c:
            close(cs, g[f"s{i}_col_sum"], scale=np.abs(a).sum())                       # as written where defined
        else:
            close(cs, ora.col_sum_intended(a.astype(np.float64)), scale=np.abs(a).sum(1, keepdims=True))   # intent where UB
        close(host.frobenius_norm(mat(a)), g[f"s{i}_frobenius"])
        assert host.max_value(mat(a)) == F32(g[f"s{i}_max"])
        x = a.copy(); m = mat(x); host.matrix_z_score_normalize(C.byref(m)); close(x, g[f"s{i}_zscore"], rtol=1e-5, scale=1)
        bias = uniform(500 + i, (r, 1), dtype=F32); x = a.copy(); m, mb2 = mat(x), mat(bias)
        host.matrix_add_tile_columns(C.byref(m), C.byref(mb2)); close(x, g[f"s{i}_tile_cols"], scale=2)
        bias = uniform(600 + i, (1, c), dtype=F32); x = a.copy(); m, mb2 = mat(x), mat(bias)
        host.matrix_add_tile_rows(C.byref(m), C.byref(mb2)); close(x, g[f"s{i}_tile_rows"], scale=2)
        x = a.copy(); host.relu(x.ctypes.data_as(C.POINTER(C.c_float)), x.size); assert np.array_equal(x, g[f"s{i}_relu"].astype(F32))
        x = a * 4; host.softmax(x.ctypes.data_as(C.POINTER(C.c_float)), r, c); close(x, g[f"s{i}_softmax_cols"], rtol=1e-5, scale=1e-6)
        x = a * 4; host.softmax_row_wise(x.ctypes.data_as(C.POINTER(C.c_float)), r, c); close(x, g[f"s{i}_softmax_rows"], rtol=1e-5, scale=1e-6)
    ex = np.array([[1, 2, 3], [4, 5, 6]], F32)
    assert take(host, host.matrix_col_sum(mat(ex))).ravel().tolist() == [6.0, 12.0]     # the documented as-written quirk


@gpu
def test_conv_h_against_golden(host, ora):
    g = golden("conv")
    for i, cfg in enumerate(g["cfgs"][:6]):
        h, w, cin, cout, k, s = [int(v) for v in cfg]
        seed = 1000 + 10 * i
        x = uniform(seed, (cin, h, w), -1, 1, F32); kern = uniform(seed + 1, (cout, cin, k, k), -0.3, 0.3, F32)
        ho, wo = ora.out_hw(h, w, s); hw, kkc = ho * wo, k * k * cin
        im, km, pr = np.zeros((hw, kkc), F32), np.zeros((kkc, cout), F32), np.zeros((hw, cout), F32)
        out = np.full((cout, ho, wo), -777, F32)
        mim, mkm, mpr, mout = mat(im), mat(km), mat(pr), mats(out)
        cd = ConvData(C.pointer(mim), C.pointer(mkm), C.pointer(mpr), C.cast(mout, PM))
        kp, _keep = kernel_ptrs(kern)
        host.conv(mats(x), kp, C.byref(cd), cin, cout, s)
        g.check(f"c{i}_im2col", im, exact=True); g.check(f"c{i}_kmat", km, exact=True)
        bound = (np.abs(im.astype(np.float64)) @ np.abs(km.astype(np.float64))).max()
        g.check(f"c{i}_product", pr, rtol=0, atol=1e-5 * bound)
        g.check(f"c{i}_output", out, rtol=0, atol=1e-5 * bound)                           # intended: output is written
        if s == 1:
            del_y = uniform(seed + 3, (cout, h, w), -1, 1, F32)
            gim, gkm, gpr = np.zeros_like(im), np.zeros_like(km), np.zeros_like(pr)
            mgim, mgkm, mgpr = mat(gim), mat(gkm), mat(gpr)
            gcd = ConvData(C.pointer(mgim), C.pointer(mgkm), C.pointer(mgpr), None)
            dkern = np.zeros_like(kern); dkp, _k2 = kernel_ptrs(dkern); dx = np.zeros_like(x)
            host.conv_ddx(mats(del_y), C.byref(cd), C.byref(gcd), dkp, mats(dx), cin, 1)
            g.check(f"c{i}_ddx_del_q", gpr, exact=True)
            b1 = (np.abs(im.astype(np.float64).T) @ np.abs(gpr.astype(np.float64))).max()
            g.check(f"c{i}_ddx_del_kmat", gkm, rtol=0, atol=1e-5 * b1)
            g.check(f"c{i}_ddx_del_kern", dkern, rtol=0, atol=1e-5 * b1)
            b2 = (np.abs(gpr.astype(np.float64)) @ np.abs(km.astype(np.float64).T)).max()
            g.check(f"c{i}_ddx_del_col", gim, rtol=0, atol=1e-5 * b2)
            g.check(f"c{i}_ddx_del_x", dx, rtol=0, atol=1e-5 * b2 * k * k)


@gpu
def test_conv_strict_reference_mode(host, ora, monkeypatch):
    """BLA_STRICT_REFERENCE=1 restores conv() as written: product <- stale output, output untouched."""
    monkeypatch.setenv("BLA_STRICT_REFERENCE", "1")
    g = golden("conv")
    h, w, cin, cout, k, s = [int(v) for v in g["cfgs"][0]]
    x = uniform(1000, (cin, h, w), -1, 1, F32); kern = uniform(1001, (cout, cin, k, k), -0.3, 0.3, F32)
    im, km, pr = np.zeros((h * w, k * k * cin), F32), np.zeros((k * k * cin, cout), F32), np.zeros((h * w, cout), F32)
    out = np.full((cout, h, w), -777, F32)
    mim, mkm, mpr, mout = mat(im), mat(km), mat(pr), mats(out)
    cd = ConvData(C.pointer(mim), C.pointer(mkm), C.pointer(mpr), C.cast(mout, PM))
    kp, _keep = kernel_ptrs(kern)
    host.conv(mats(x), kp, C.byref(cd), cin, cout, s)
    assert np.array_equal(pr, g["aswritten_conv_product"].astype(F32)) and np.array_equal(out, g["aswritten_conv_output"].astype(F32))
    g.check("c0_im2col", im, exact=True)


@gpu
def test_norm_h_against_golden(host):
    g = golden("norm")
    for i, (c, gs, h, w) in enumerate(g["cfgs"]):
        c, gs, h, w = int(c), int(gs), int(h), int(w)
        x = uniform(2000 + i, (c, h, w), -1, 3, F32); ng = (c + gs - 1) // gs
        out = np.zeros_like(x); sd = np.zeros(ng, F32); mu = np.zeros(ng, F32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        host.group_norm(C.cast(mats(x), PM), C.cast(mats(out), PM), fp(sd), fp(mu), c, gs)
        g.check(f"n{i}_out", out, rtol=1e-5, atol=1e-6)
        up = uniform(2100 + i, (c, h, w), -1, 1, F32); dest = np.zeros_like(x)
        host.group_norm_ddx(C.cast(mats(up), PM), C.cast(mats(dest), PM), C.cast(mats(x), PM), fp(mu), fp(sd), c, gs)
        ref = g[f"n{i}_ddx"]
        assert (np.abs(dest - ref) <= 2e-5 * np.abs(ref) + 2e-5 * np.abs(ref).max()).all()


def _layer_types(host):
    class Layer(C.Structure):
        pass
    ACT = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_int)
    Layer._fields_ = [("num_nodes", C.c_int), ("nodes", PM), ("raw_nodes", PM), ("weights", PM), ("biases", PM),
                      ("previous_layer", C.POINTER(Layer)), ("activation", ACT), ("activation_ddx", ACT),
                      ("has_previous_layer", C.c_char), ("has_nodes", C.c_char)]
    host.make_matrix.restype = PM; host.make_matrix.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float)]
    host.feed_forward.argtypes = [C.POINTER(Layer)]
    host.back_propagate_errors.argtypes = [C.POINTER(Layer), C.POINTER(C.c_float), C.c_float]
    host.free_layer_data.argtypes = [Layer]
    host.load_weights_from_csv.argtypes = [C.POINTER(Layer), C.c_char_p]
    host.load_biases_from_csv.argtypes = [C.POINTER(Layer), C.c_char_p]
    host.read_csv_contents.restype = C.POINTER(C.c_float); host.read_csv_contents.argtypes = [C.c_char_p]
    return Layer, ACT


def _arr(pm):
    return np.ctypeslib.as_array(pm.contents.data, shape=(pm.contents.rows, pm.contents.cols)).copy()


@gpu
def test_layer_h_main_c_known_answer(host, tmp_path):
    """main.c:52-87 through the drop-in layer.h, against what the reference's own lib/layer.c computed (tests/golden/layer.npz, case
    "main"): the 3-2-2 net is loaded from the reference's data files with load_weights_from_csv / load_biases_from_csv (both layers read
    the same files, main.c:60-61,70-71), activation x0.1, derivative 0.1, expectations {0.5, 0.5}, learn rate 0.05; released with
    free_layer_data like main.c:85-87.  SURVEY section 4: output [2.47; 5.39]; then weights [0.91 1.80; 2.78 3.51], biases [0.08; 0.15]."""
    g = golden("layer")
    Layer, ACT = _layer_types(host)
    paths = {}
    for f in ("inputs", "weights", "biases"):
        paths[f] = str(tmp_path / f"{f}.csv")
        with open(paths[f], "wb") as fh:
            fh.write(bytes(g[f"main_{f}_csv"].astype(np.uint8)))

    @ACT
    def act(p, n):
        for i in range(n):
            p[i] = float(p[i]) * 0.1          # data[i] *= 0.1 (main.c:9): double product, stored as float

    @ACT
    def act_ddx(p, n):
        for i in range(n):
            p[i] = 0.1
    inp = Layer(3, host.make_matrix(3, 1, host.read_csv_contents(paths["inputs"].encode())), None, None, None, None, act, act_ddx, b"\x00", b"\x01")
    hid = Layer(2, None, None, None, None, C.pointer(inp), act, act_ddx, b"\x01", b"\x00")
    outl = Layer(2, None, None, None, None, C.pointer(hid), act, act_ddx, b"\x01", b"\x00")
    for l in (hid, outl):
        host.load_weights_from_csv(C.byref(l), paths["weights"].encode())
        host.load_biases_from_csv(C.byref(l), paths["biases"].encode())
    host.load_weights_from_csv(C.byref(inp), paths["weights"].encode())       # input layer: a no-op (lib/layer.c:35-37)
    assert not inp.weights
    assert np.array_equal(_arr(hid.weights), g["main_w0"]) and np.array_equal(_arr(outl.weights), g["main_w1"])
    assert np.array_equal(_arr(outl.biases), g["main_b1"].astype(F32))
    host.feed_forward(C.byref(inp))                                            # no-op as well (:7-9)
    host.feed_forward(C.byref(hid)); host.feed_forward(C.byref(outl))
    tol = dict(rtol=2e-6, atol=0)
    for i, l in enumerate((hid, outl)):
        np.testing.assert_allclose(_arr(l.nodes), g[f"main_nodes{i}"], **tol)
        np.testing.assert_allclose(_arr(l.raw_nodes), g[f"main_raw{i}"], **tol)
    assert np.allclose(_arr(outl.nodes).ravel(), [2.47, 5.39], atol=5e-3)
    e = g["main_expect"].astype(F32)
    host.back_propagate_errors(C.byref(outl), e.ctypes.data_as(C.POINTER(C.c_float)), float(g["main_lr"]))
    for i, l in enumerate((hid, outl)):
        np.testing.assert_allclose(_arr(l.weights), g[f"main_w{i}_new"], **tol)
        np.testing.assert_allclose(_arr(l.biases), g[f"main_b{i}_new"], **tol)
    assert np.allclose(_arr(outl.weights).ravel(), [0.91, 1.80, 2.78, 3.51], atol=5e-3) and np.allclose(_arr(outl.biases).ravel(), [0.08, 0.15], atol=5e-3)
    host.feed_forward(C.byref(hid))        # a second pass takes the has_nodes branch (frees only the structs, :12-15)
    host.free_layer_data(outl); host.free_layer_data(hid); host.free_matrix(inp.nodes)


@gpu
def test_layer_h_mlp_against_reference_run(host):
    """12-7-5-3 net, leaky-ReLU callbacks: feed_forward / back_propagate_errors / do_back_propagate_errors (lib/layer.c:6-107) against
    the reference's own run (layer.npz case "mlp"), 1e-5 relative with an absolute floor of 1e-6 of the tensor's scale."""
    g = golden("layer")
    Layer, ACT = _layer_types(host)
    malloc = C.CDLL(None).malloc; malloc.restype = C.c_void_p; malloc.argtypes = [C.c_size_t]

    def heap(a):   # the library frees these with free(): they must come from malloc
        a = np.ascontiguousarray(a, F32)
        p = malloc(a.nbytes); C.memmove(p, a.ctypes.data, a.nbytes)
        return host.make_matrix(a.shape[0], a.shape[1], C.cast(p, C.POINTER(C.c_float)))

    @ACT
    def act(p, n):
        for i in range(n):
            if p[i] < 0:
                p[i] = p[i] * 0.25

    @ACT
    def act_ddx(p, n):
        for i in range(n):
            p[i] = 1.0 if p[i] > 0 else 0.25
    sizes = [int(v) for v in g["mlp_sizes"]]
    layers = [Layer(sizes[0], heap(g["mlp_x"]), None, None, None, None, act, act_ddx, b"\x00", b"\x01")]
    for i in range(1, 4):
        layers.append(Layer(sizes[i], None, None, heap(g[f"mlp_w{i-1}"]), heap(g[f"mlp_b{i-1}"]), C.pointer(layers[i - 1]), act, act_ddx, b"\x01", b"\x00"))
    for l in layers[1:]:
        host.feed_forward(C.byref(l))

    def close(got, ref):
        return (np.abs(got - ref) <= 1e-5 * np.abs(ref) + 1e-6 * np.abs(ref).max()).all()
    for i, l in enumerate(layers[1:]):
        assert close(_arr(l.nodes), g[f"mlp_nodes{i}"]) and close(_arr(l.raw_nodes), g[f"mlp_raw{i}"]), i
    e = g["mlp_expect"].astype(F32)
    host.back_propagate_errors(C.byref(layers[-1]), e.ctypes.data_as(C.POINTER(C.c_float)), float(g["mlp_lr"]))
    for i, l in enumerate(layers[1:]):
        assert close(_arr(l.weights), g[f"mlp_w{i}_new"]) and close(_arr(l.biases), g[f"mlp_b{i}_new"]), i
    for l in reversed(layers[1:]):
        host.free_layer_data(l)
    host.free_matrix(layers[0].nodes)


@gpu
def test_library_leaves_the_callers_rand_stream_alone(tmp_path, pkg):
    """tests/c/rand_stream.c: srand(42) ... rand() around bla_init, allocations, copies, a launch, a second context, the exchange object
    and an RCCL communicator -- the draws the reference's programs make between library calls (sampler, dropout) must be the seeded ones."""
    pkg.build_native()
    exe = str(tmp_path / "rand_stream")
    r = _cc(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "rand_stream.c"), "-o", exe,
             "-L", CSRC, "-l:libbla_hip.so", f"-Wl,-rpath,{CSRC}"])
    assert r.returncode == 0, r.stderr
    r = _cc([exe])
    assert r.returncode == 0 and r.stdout.count(": ok") == 8, r.stdout + r.stderr

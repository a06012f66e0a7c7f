#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (oracle/_ref/libref.so).

Run in the build container only (needs /root/reference); the .npz files are
committed, this script is committed, the reference never travels.

    make -C oracle && python oracle/gen_golden.py

What is recorded (SURVEY.md 8(c) list): GEMM known answers (main.c:20-41 and the
data/a.csv x data/b.csv plumbing case) and random shapes; every lib/matrix.c
elementwise/reduction op; every lib/conv.c stage, conv()/conv_ddx() as written
(sentinels document quirk Q1) and in the intended composition; group_norm fwd/bwd;
relu/softmax; one MNIST-NN SGD step (model/mnist_nn.c:218-315 driven call by call
through the reference's matrix.h functions).

Big outputs are stored as a digest (sum, abs-sum and a 4096-point strided sample)
so the fixtures stay small; small ones in full.  All values are fp64, exactly as
the reference computed them.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
REFROOT = "/root/reference"
L = ref.lib()
PD = C.POINTER(C.c_double)


sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests", "golden"))
from inputs import uniform, randint, sample_idx  # noqa: E402  (shared with the tests)


def put(d, name, arr, full_limit=16384):
    """Store arr in full when small, else as a digest."""
    arr = np.ascontiguousarray(arr, np.float64)
    if arr.size <= full_limit:
        d[name] = arr
    else:
        flat = arr.ravel()
        d[name + "__shape"] = np.array(arr.shape, np.int64)
        d[name + "__sum"] = np.array(flat.sum(dtype=np.float64))
        d[name + "__asum"] = np.array(np.abs(flat).sum(dtype=np.float64))
        d[name + "__sample"] = flat[sample_idx(flat.size)]


def read_csv(path, n):
    """The reference's own CSV reader (lib/csv.c:18-57) -> float32[n]."""
    L.read_csv_contents.restype = C.POINTER(C.c_float)
    p = L.read_csv_contents(path.encode())
    out = np.ctypeslib.as_array(p, shape=(n,)).copy()
    return out


# ---- GEMM ---------------------------------------------------------------------
def gen_gemm():
    d = {}
    # main.c:20-41 known answer (printed "1.40 8.50 / 5.00 19.00" by the fp32-typedef build)
    a = np.array([[1, 2, 3], [4, 5, 6]], np.float64)
    b = np.array([[1, 0.5], [0.2, 1], [0, 2]], np.float64)
    d["kat_main_a"], d["kat_main_b"], d["kat_main_c"] = a, b, ref.matmul(a, b)
    # config 1 plumbing: data/a.csv (3x3) x data/b.csv (6 values as 3x2), read by lib/csv.c
    a = read_csv(f"{REFROOT}/data/a.csv", 9).astype(np.float64).reshape(3, 3)
    b = read_csv(f"{REFROOT}/data/b.csv", 6).astype(np.float64).reshape(3, 2)
    d["csv_a"], d["csv_b"], d["csv_c"] = a, b, ref.matmul(a, b)
    shapes = [(1, 1, 1), (2, 3, 2), (7, 5, 3), (33, 17, 65), (65, 129, 33), (129, 65, 257), (257, 33, 129),
              (128, 128, 128), (64, 256, 64), (10, 128, 64), (1, 512, 128), (300, 1, 40)]
    d["shapes"] = np.array(shapes, np.int64)
    for i, (m, k, n) in enumerate(shapes):
        a = uniform(100 + i, (m, k)); b = uniform(200 + i, (k, n))
        c = ref.matmul_inplace(a, b)
        assert np.array_equal(c, ref.matmul(a, b))
        d[f"rand{i}_c"] = c  # inputs are regenerated from the seed by tests (tests/golden/inputs.py)
    np.savez_compressed(os.path.join(GOLD, "gemm.npz"), **d)


# ---- lib/matrix.c ops ------------------------------------------------------------
def gen_matrix_ops():
    d = {}
    shapes = [(3, 5), (16, 16), (37, 64), (64, 37)]
    d["shapes"] = np.array(shapes, np.int64)
    for i, (r, c) in enumerate(shapes):
        a = uniform(300 + i, (r, c), -2, 2); b = uniform(400 + i, (r, c), -2, 2)
        d[f"s{i}_scale"] = ref.inplace1("matrix_scale", a, C.c_double(-0.37))
        d[f"s{i}_add"] = ref.inplace2("matrix_add", a, b)
        d[f"s{i}_hadamard"] = ref.inplace2("matrix_multiply_elementwise", a, b)
        d[f"s{i}_transpose"] = ref.inplace1("matrix_transpose", a)
        d[f"s{i}_row_sum"] = ref.take(L.matrix_row_sum(ref.mat(a)))
        if r <= c:  # defined behaviour only (Q2); rows > cols is a heap over-read in the reference
            d[f"s{i}_col_sum"] = ref.take(L.matrix_col_sum(ref.mat(a)))
        d[f"s{i}_frobenius"] = np.array(L.frobenius_norm(ref.mat(a)))
        d[f"s{i}_max"] = np.array(L.max_value(ref.mat(a)))
        d[f"s{i}_zscore"] = ref.inplace1("matrix_z_score_normalize", a)
        bias_c = uniform(500 + i, (r, 1)); bias_r = uniform(600 + i, (1, c))
        d[f"s{i}_tile_cols"] = ref.inplace2("matrix_add_tile_columns", a, bias_c)
        d[f"s{i}_tile_rows"] = ref.inplace2("matrix_add_tile_rows", a, bias_r)
        # tiling a wider b (i % b.cols), lib/matrix.c:192
        if c % 2 == 0:
            d[f"s{i}_tile_cols2"] = ref.inplace2("matrix_add_tile_columns", a, uniform(700 + i, (r, 2)))
        d[f"s{i}_relu"] = ref.data_fn("relu", a, a.size)
        d[f"s{i}_softmax_cols"] = ref.data_fn("softmax", a * 4, r, c)
        d[f"s{i}_softmax_rows"] = ref.data_fn("softmax_row_wise", a * 4, r, c)
    # the SURVEY's documented example: [1 2 3; 4 5 6] -> 6, 12 (true 6, 15)
    ex = np.array([[1, 2, 3], [4, 5, 6]], np.float64)
    d["colsum_example"] = ref.take(L.matrix_col_sum(ref.mat(ex)))
    np.savez_compressed(os.path.join(GOLD, "matrix_ops.npz"), **d)


# ---- lib/conv.c -----------------------------------------------------------------
def run_conv_stages(d, tag, h, w, cin, cout, k, s, seed, full_limit=16384):
    ho = int(np.ceil(np.float32(h) / s)); wo = int(np.ceil(np.float32(w) / s))
    x = uniform(seed, (cin, h, w), -1, 1)
    kern = uniform(seed + 1, (cout, cin, k, k), -0.3, 0.3)
    xm = ref.mats(x); kp, _keep = ref.kernel_ptrs(kern)
    im = np.zeros((ho * wo, k * k * cin)); imm = ref.mat(im)
    L._im2col(xm, C.byref(imm), k, cin, s)
    km = np.zeros((k * k * cin, cout)); kmm = ref.mat(km)
    L._reshape_kernels_matrix(kp, C.byref(kmm))
    prod = ref.matmul_inplace(im, km)
    # intended final step: as written reshape_channels_matrix(channels, matrix) does channels <- matrix (Q1)
    out = np.zeros((cout, ho, wo)); om = ref.mats(out); pm = ref.mat(prod)
    L.reshape_channels_matrix(om, C.byref(pm))
    back = np.zeros_like(prod); bm = ref.mat(back)
    L.reshape_matrix_channels(C.byref(bm), om)
    assert np.array_equal(back, prod)
    k2 = np.zeros_like(kern); k2p, _keep2 = ref.kernel_ptrs(k2)
    L._reshape_matrix_kernels(C.byref(kmm), k2p)
    assert np.array_equal(k2, kern)
    put(d, f"{tag}_im2col", im, full_limit); put(d, f"{tag}_kmat", km, full_limit)
    put(d, f"{tag}_product", prod, full_limit); put(d, f"{tag}_output", out, full_limit)
    if s == 1:
        # col2im on a fresh random column matrix, and the intended conv_ddx chain
        cols = uniform(seed + 2, (h * w, k * k * cin), -1, 1)
        dx = np.zeros((cin, h, w)); dxm = ref.mats(dx); cm = ref.mat(cols)
        L._col2im(C.byref(cm), dxm, k, cin, 1)
        put(d, f"{tag}_col2im", dx, full_limit)
        del_y = uniform(seed + 3, (cout, h, w), -1, 1)
        dq = np.zeros((h * w, cout)); dqm = ref.mat(dq)
        L.reshape_matrix_channels(C.byref(dqm), ref.mats(del_y))          # del_Q <- del_Y (intended direction)
        imt = ref.inplace1("matrix_transpose", im)
        dkm = ref.matmul_inplace(imt, dq)                                 # conv.c:221-222
        dkern = np.zeros_like(kern); dkp, _k3 = ref.kernel_ptrs(dkern); dkmm = ref.mat(dkm)
        L._reshape_matrix_kernels(C.byref(dkmm), dkp)                     # :223
        kmt = ref.inplace1("matrix_transpose", km)
        dcol = ref.matmul_inplace(dq, kmt)                                # :225-226
        dxi = np.zeros((cin, h, w)); dcm = ref.mat(dcol)
        L._col2im(C.byref(dcm), ref.mats(dxi), k, cin, 1)                 # :228
        put(d, f"{tag}_ddx_del_q", dq, full_limit); put(d, f"{tag}_ddx_del_kmat", dkm, full_limit)
        put(d, f"{tag}_ddx_del_kern", dkern, full_limit); put(d, f"{tag}_ddx_del_col", dcol, full_limit)
        put(d, f"{tag}_ddx_del_x", dxi, full_limit)
    return x, kern, im, km, prod


def gen_conv():
    d = {}
    cfgs = [(8, 8, 3, 4, 3, 1), (8, 8, 3, 4, 3, 2), (7, 9, 2, 3, 3, 2), (6, 6, 2, 5, 1, 1), (32, 32, 3, 8, 3, 2),
            (5, 7, 3, 2, 3, 1), (32, 32, 128, 128, 3, 1)]
    d["cfgs"] = np.array(cfgs, np.int64)
    for i, (h, w, cin, cout, k, s) in enumerate(cfgs):
        run_conv_stages(d, f"c{i}", h, w, cin, cout, k, s, 1000 + 10 * i)
    # conv()/conv_ddx() AS WRITTEN with sentinels (documents Q1): cfg 0
    h, w, cin, cout, k, s = cfgs[0]
    x = uniform(1000, (cin, h, w), -1, 1); kern = uniform(1001, (cout, cin, k, k), -0.3, 0.3)
    im = np.zeros((h * w, k * k * cin)); km = np.zeros((k * k * cin, cout)); prod = np.zeros((h * w, cout))
    out = np.full((cout, h, w), -777.0)
    imm, kmm, pm = ref.mat(im), ref.mat(km), ref.mat(prod)
    om = ref.mats(out)

    class ConvData(C.Structure):
        _fields_ = [("im2col", ref.PM), ("kernel_matrix", ref.PM), ("product", ref.PM), ("output", ref.PM)]
    cd = ConvData(C.pointer(imm), C.pointer(kmm), C.pointer(pm), C.cast(om, ref.PM))
    kp, _keep = ref.kernel_ptrs(kern)
    L.conv(ref.mats(x), kp, C.byref(cd), cin, cout, s)
    d["aswritten_conv_product"] = prod.copy()      # == -777 everywhere: product <- stale output
    d["aswritten_conv_output"] = out.copy()        # == -777 everywhere: never written
    # conv_ddx as written: del_Y is overwritten from the stale del_Q, gradients come from the stale del_Q
    del_y = uniform(1003, (cout, h, w), -1, 1)
    g_im = np.zeros_like(im); g_km = np.zeros_like(km); g_prod = uniform(1004, (h * w, cout), -1, 1)
    g_prod0 = g_prod.copy()
    gimm, gkmm, gpm = ref.mat(g_im), ref.mat(g_km), ref.mat(g_prod)
    gcd = ConvData(C.pointer(gimm), C.pointer(gkmm), C.pointer(gpm), None)
    # restore forward workspaces (im2col/kernel_matrix are valid; they were computed before the last step)
    dkern = np.zeros_like(kern); dkp, _k2 = ref.kernel_ptrs(dkern)
    dx = np.zeros((cin, h, w))
    L.conv_ddx(ref.mats(del_y), C.byref(cd), C.byref(gcd), dkp, ref.mats(dx), cin, 1)
    d["aswritten_ddx_stale_del_q"] = g_prod0
    d["aswritten_ddx_del_y_after"] = del_y.copy()
    d["aswritten_ddx_del_kern"] = dkern
    d["aswritten_ddx_del_x"] = dx
    np.savez_compressed(os.path.join(GOLD, "conv.npz"), **d)


# ---- lib/norm.c -------------------------------------------------------------------
def gen_norm():
    d = {}
    cfgs = [(3, 32, 8, 8), (128, 32, 8, 8), (40, 32, 4, 6), (64, 16, 2, 2)]
    d["cfgs"] = np.array(cfgs, np.int64)
    L.group_norm.argtypes = [ref.PM, ref.PM, PD, PD, C.c_int, C.c_int]
    L.group_norm_ddx.argtypes = [ref.PM, ref.PM, ref.PM, PD, PD, C.c_int, C.c_int]
    for i, (c, g, h, w) in enumerate(cfgs):
        x = uniform(2000 + i, (c, h, w), -1, 3)
        ng = (c + g - 1) // g
        out = np.zeros_like(x); sd = np.zeros(ng); mu = np.zeros(ng)
        L.group_norm(C.cast(ref.mats(x), ref.PM), C.cast(ref.mats(out), ref.PM),
                     sd.ctypes.data_as(PD), mu.ctypes.data_as(PD), c, g)
        up = uniform(2100 + i, (c, h, w), -1, 1)
        dest = np.zeros_like(x)
        L.group_norm_ddx(C.cast(ref.mats(up), ref.PM), C.cast(ref.mats(dest), ref.PM), C.cast(ref.mats(x), ref.PM),
                         mu.ctypes.data_as(PD), sd.ctypes.data_as(PD), c, g)
        d[f"n{i}_out"], d[f"n{i}_stdevs"], d[f"n{i}_means"], d[f"n{i}_ddx"] = out, sd, mu, dest
    # documented probe: group {1..8} -> mean 4.5, "stdev" 5.25, out[0] = -0.6667
    x = np.arange(1, 9, dtype=np.float64).reshape(2, 2, 2)
    out = np.zeros_like(x); sd = np.zeros(1); mu = np.zeros(1)
    L.group_norm(C.cast(ref.mats(x), ref.PM), C.cast(ref.mats(out), ref.PM), sd.ctypes.data_as(PD), mu.ctypes.data_as(PD), 2, 32)
    d["probe_out"], d["probe_sd"], d["probe_mu"] = out, sd, mu
    np.savez_compressed(os.path.join(GOLD, "norm.npz"), **d)


# ---- model/mnist_nn.c:218-315 driven through the reference's matrix.h: ref.mnist_step (oracle/ref.py; bench.py's cpu_baseline leg times the same driver)
ref_mnist_step = ref.mnist_step


def ref_mnist_metrics(a3, y):
    """The bookkeeping of model/mnist_nn.c:237-257 for one batch: num_correct by the loop of :240-250, batch_loss by the reference's own
    cross_entropy_loss (:83-91, from oracle/_ref/libref_mnist.so) on the flat chunks [10k, 10k+10) of both matrices exactly as :252-254
    takes them, added in k order."""
    M = C.CDLL(os.path.join(HERE, "_ref", "libref_mnist.so"))
    M.cross_entropy_loss.restype = C.c_double
    M.cross_entropy_loss.argtypes = [PD, PD, C.c_int]
    a3 = np.ascontiguousarray(a3, np.float64); y = np.ascontiguousarray(y, np.float64)
    n3, B = a3.shape
    correct, loss = 0, 0.0
    for k in range(B):
        pred, best = 0, 0.0
        for p in range(n3):
            if a3[p, k] > best:
                best, pred = a3[p, k], p
        if y.ravel()[k + pred * B] == 1:
            correct += 1
        off = k * n3 * 8
        loss += M.cross_entropy_loss(C.cast(a3.ctypes.data + off, PD), C.cast(y.ctypes.data + off, PD), n3)
    return loss, correct


def load_mnist_params():
    shapes = [(256, 784), (256, 1), (128, 256), (128, 1), (10, 128), (10, 1)]
    files = ["weights_1", "biases_1", "weights_2", "biases_2", "weights_3", "biases_3"]
    return [read_csv(f"{REFROOT}/data/mnist_nn/{f}.csv", r * c).reshape(r, c) for f, (r, c) in zip(files, shapes)]


def gen_mnist():
    params32 = load_mnist_params()   # trained reference weights (fp32 as lib/csv.c returns them)
    np.savez_compressed(os.path.join(GOLD, "mnist_nn_params.npz"),
                        **{n: p for n, p in zip(["w1", "b1", "w2", "b2", "w3", "b3"], params32)})
    d = {}
    params = [p.astype(np.float64) for p in params32]
    for tag, B, intended in [("b256_aswritten", 256, False), ("b64_intended", 64, True), ("b256_intended", 256, True)]:
        x_raw = randint(3000 + B, (784, B), 256).astype(np.float64)
        lab = randint(3100 + B, (B,), 10)
        y = np.zeros((10, B)); y[lab, np.arange(B)] = 1
        new, acts, grads = ref_mnist_step(params, x_raw, y, intended)
        for n, v in zip(["w1", "b1", "w2", "b2", "w3", "b3"], new):
            put(d, f"{tag}_new_{n}", v, 4096)
        for n, v in zip(["dw1", "db1", "dw2", "db2", "dw3", "db3"], grads):
            put(d, f"{tag}_{n}", v, 4096)
        for n, v in acts.items():
            put(d, f"{tag}_{n}", v, 4096)
        loss, correct = ref_mnist_metrics(acts["a3"], y)
        d[f"{tag}_batch_loss"] = np.array(loss); d[f"{tag}_num_correct"] = np.array(correct, np.int64)
    # tiny architecture stored in full: 12 -> 8 -> 6 -> 4, B = 16 (all col_sum windows in bounds)
    sizes = (12, 8, 6, 4); B = 16
    tp = [uniform(3200, (8, 12)), uniform(3201, (8, 1)), uniform(3202, (6, 8)), uniform(3203, (6, 1)),
          uniform(3204, (4, 6)), uniform(3205, (4, 1))]
    x_raw = randint(3206, (12, B), 256).astype(np.float64)
    lab = randint(3207, (B,), 4); y = np.zeros((4, B)); y[lab, np.arange(B)] = 1
    for tag, intended in [("tiny_aswritten", False), ("tiny_intended", True)]:
        new, acts, grads = ref_mnist_step(tp, x_raw, y, intended)
        for n, v in zip(["w1", "b1", "w2", "b2", "w3", "b3"], new):
            d[f"{tag}_new_{n}"] = v
        for n, v in zip(["dw1", "db1", "dw2", "db2", "dw3", "db3"], grads):
            d[f"{tag}_{n}"] = v
        for n, v in acts.items():
            d[f"{tag}_{n}"] = v
        loss, correct = ref_mnist_metrics(acts["a3"], y)
        d[f"{tag}_batch_loss"] = np.array(loss); d[f"{tag}_num_correct"] = np.array(correct, np.int64)
    np.savez_compressed(os.path.join(GOLD, "mnist_step.npz"), **d)


# ---- model/cifar_unet.c glue ops, called in the reference itself (oracle/_ref/libref_unet.so) ---------------
def gen_unet_glue():
    U = C.CDLL(os.path.join(HERE, "_ref", "libref_unet.so"))
    libc = C.CDLL(None)
    d = {}
    c, h, w = 6, 8, 8
    x = uniform(4000, (c, h, w), -1, 1); src = uniform(4001, (c, h, w), -1, 1)
    rr = ref.inplace1("matrix_scale", x.reshape(c, h * w), C.c_double(1.0)).reshape(c, h, w).copy()
    np.maximum(rr, 0, out=rr)                                     # a relu result (zeros where x < 0)
    dest = np.zeros_like(x)
    U.multi_channel_relu_ddx(ref.mats(src), ref.mats(dest), ref.mats(rr), c)
    d["relu_mask"] = dest
    t = uniform(4002, (c, 1)); xa = x.copy(); tm = ref.mat(t)
    U._add_time_embedding(ref.mats(xa), C.byref(tm), c)
    d["add_time_embedding"] = xa
    # dropout: the draws come from libc rand(); replay them to record which elements were dropped
    rate = C.c_float.in_dll(U, "DROPOUT_RATE").value if hasattr(U, "DROPOUT_RATE") else None
    libc.srand(1234); y = np.zeros_like(x)
    U._dropout(ref.mats(x), ref.mats(y), c)
    libc.srand(1234); libc.rand.restype = C.c_int
    RAND_MAX = 2147483647
    draws = np.array([np.float32(libc.rand()) / np.float32(RAND_MAX) for _ in range(x.size)], np.float32)
    d["dropout_draws"] = draws
    d["dropout_y"] = y
    d["dropout_dropped"] = (y == 0).astype(np.uint8).ravel()
    g = uniform(4003, (c, h, w), -1, 1); gm = g.copy()
    U._dropout_mask(ref.mats(gm), ref.mats(y), c)
    d["dropout_mask"] = gm
    for tag, (ih, iw, oh, ow, sc) in {"nn2": (4, 4, 8, 8, 2), "nn3": (3, 2, 7, 5, 3)}.items():
        xi = uniform(4010 + sc, (c, ih, iw), -1, 1); out = np.zeros((c, oh, ow))
        U._nearest_neighbours(ref.mats(xi), ref.mats(out), c, sc)
        d[tag + "_up"] = out
        gs = uniform(4020 + sc, (c, oh, ow), -1, 1); dst = np.full((c, ih, iw), 7.0)
        U._nearest_neighbours_ddx(ref.mats(gs), ref.mats(dst), c, sc)
        d[tag + "_ddx"] = dst
    s_ = ref.data_fn("softmax_row_wise", uniform(4030, (16, 16), -3, 3), 16, 16); gr = uniform(4031, (16, 16), -1, 1)
    o = np.zeros((16, 16)); ms, mg, mo = ref.mat(s_), ref.mat(gr), ref.mat(o)
    U._softmax_ddx(C.byref(ms), C.byref(mg), C.byref(mo))
    d["softmax_ddx_s"], d["softmax_ddx"] = s_, o
    a = uniform(4040, (3, 4, 4)); b = uniform(4041, (3, 4, 4)); cat = np.zeros((6, 4, 4))
    U._concat_skip(ref.mats(a), ref.mats(b), ref.mats(cat), 3)
    d["concat"] = cat
    half = np.zeros((3, 4, 4)); U._split_concat(ref.mats(cat), ref.mats(half), 3, 1)
    d["split_second"] = half
    np.savez_compressed(os.path.join(GOLD, "unet_glue.npz"), **d)


# ---- self-attention block: the reference's own call sequence (model/cifar_unet.c:999-1022,1261-1337) driven step by
# step through its matrix.h / util.h functions and its _softmax_ddx, with the channel reshapes in the INTENDED direction
def ref_attention(U, x, wq, wk, wv, w, bias, del_y, jac_from_raw):
    c, hh, ww = x.shape; s = hh * ww; d = wq.shape[1]
    T_ = lambda m: ref.inplace1("matrix_transpose", m)
    z = np.zeros((s, c)); zm = ref.mat(z); L.reshape_matrix_channels(C.byref(zm), ref.mats(x))       # input <- X
    q, k, v = ref.matmul_inplace(z, wq), ref.matmul_inplace(z, wk), ref.matmul_inplace(z, wv)
    wts = ref.matmul_inplace(q, T_(k))
    wts = ref.inplace1("matrix_scale", wts, C.c_double(1.0 / np.sqrt(float(d))))
    raw = wts.copy()
    wts = ref.data_fn("softmax_row_wise", wts, s, s)
    att = ref.matmul_inplace(wts, v)
    dense = ref.inplace2("matrix_add_tile_rows", ref.matmul_inplace(att, w), bias)
    out = np.zeros((c, hh, ww)); dm = ref.mat(dense); L.reshape_channels_matrix(ref.mats(out), C.byref(dm))   # output <- dense
    fwd = dict(q=q, k=k, v=v, raw=raw, wts=wts, att=att, out=out)
    dyp = np.zeros((s, c)); dym = ref.mat(dyp); L.reshape_matrix_channels(C.byref(dym), ref.mats(del_y))
    del_w = ref.matmul_inplace(T_(att), dyp)
    del_p = ref.matmul_inplace(dyp, T_(w))
    del_v = ref.matmul_inplace(T_(wts), del_p)
    del_s = ref.matmul_inplace(del_p, T_(v))
    jac = np.ascontiguousarray(raw if jac_from_raw else wts); del_i = np.zeros((s, s))
    mj, ms_, mi = ref.mat(jac), ref.mat(del_s), ref.mat(del_i)
    U._softmax_ddx(C.byref(mj), C.byref(ms_), C.byref(mi))
    del_i = ref.inplace1("matrix_scale", del_i, C.c_double(1.0 / np.sqrt(float(d))))
    del_q = ref.matmul_inplace(del_i, k)
    del_k = ref.matmul_inplace(T_(del_i), q)
    zt = T_(z)
    del_wk, del_wq, del_wv = ref.matmul_inplace(zt, del_k), ref.matmul_inplace(zt, del_q), ref.matmul_inplace(zt, del_v)
    del_z = ref.matmul_inplace(del_q, T_(wq))
    del_z = ref.inplace2("matrix_add", del_z, ref.matmul_inplace(del_k, T_(wk)))
    del_z = ref.inplace2("matrix_add", del_z, ref.matmul_inplace(del_v, T_(wv)))
    del_x = np.zeros((c, hh, ww)); dzm = ref.mat(del_z); L.reshape_channels_matrix(ref.mats(del_x), C.byref(dzm))
    return fwd, dict(del_wq=del_wq, del_wk=del_wk, del_wv=del_wv, del_w=del_w, del_x=del_x)


def attention_inputs(i, c, hh, d):
    sd = 5000 + 20 * i
    return (uniform(sd, (c, hh, hh), -1, 1), uniform(sd + 1, (c, d), -0.2, 0.2), uniform(sd + 2, (c, d), -0.2, 0.2),
            uniform(sd + 3, (c, d), -0.2, 0.2), uniform(sd + 4, (d, c), -0.2, 0.2), uniform(sd + 5, (1, c), -0.1, 0.1),
            uniform(sd + 6, (c, hh, hh), -1, 1))


def gen_attention():
    U = C.CDLL(os.path.join(HERE, "_ref", "libref_unet.so"))
    d = {}
    cfgs = [(8, 3, 4), (32, 4, 16), (256, 16, 16)]   # (C, H=W, key dim); the last is the U-Net's 16x16 block (S = 256)
    d["cfgs"] = np.array(cfgs, np.int64)
    for i, (c, hh, kd) in enumerate(cfgs):
        x, wq, wk, wv, w, b, dy = attention_inputs(i, c, hh, kd)
        for tag, jr in (("intended", False), ("rawjac", True)):
            fwd, bwd = ref_attention(U, x, wq, wk, wv, w, b, dy, jr)
            if not jr:
                for n, v_ in fwd.items():
                    put(d, f"a{i}_{n}", v_, 8192)
            for n, v_ in bwd.items():
                put(d, f"a{i}_{tag}_{n}", v_, 8192)
    np.savez_compressed(os.path.join(GOLD, "attention.npz"), **d)


# ---- ResNet block: the reference's call sequence (model/cifar_unet.c:1044-1072, 1180-1227) through its own functions,
# in the INTENDED composition (conv output delivered, gradients into gradient structs)
def ref_conv_fwd(x, kern, s=1):
    cin, h, w = x.shape; cout, _, k, _ = kern.shape
    ho = int(np.ceil(np.float32(h) / s)); wo = int(np.ceil(np.float32(w) / s))
    im = np.zeros((ho * wo, k * k * cin)); imm = ref.mat(im)
    L._im2col(ref.mats(x), C.byref(imm), k, cin, s)
    km = np.zeros((k * k * cin, cout)); kmm = ref.mat(km); kp, _keep = ref.kernel_ptrs(kern)
    L._reshape_kernels_matrix(kp, C.byref(kmm))
    prod = ref.matmul_inplace(im, km)
    out = np.zeros((cout, ho, wo)); pm = ref.mat(prod)
    L.reshape_channels_matrix(ref.mats(out), C.byref(pm))
    return out, im, km


def ref_conv_bwd(del_y, im, km, cin, k):
    cout, h, w = del_y.shape
    dq = np.zeros((h * w, cout)); dqm = ref.mat(dq)
    L.reshape_matrix_channels(C.byref(dqm), ref.mats(del_y))
    dkm = ref.matmul_inplace(ref.inplace1("matrix_transpose", im), dq)
    dkern = np.zeros((cout, cin, k, k)); dkp, _keep = ref.kernel_ptrs(dkern); dkmm = ref.mat(dkm)
    L._reshape_matrix_kernels(C.byref(dkmm), dkp)
    dcol = ref.matmul_inplace(dq, ref.inplace1("matrix_transpose", km))
    dx = np.zeros((cin, h, w)); dcm = ref.mat(dcol)
    L._col2im(C.byref(dcm), ref.mats(dx), k, cin, 1)
    return dkern, dx


def ref_group_norm(x, gs):
    c = x.shape[0]; ng = (c + gs - 1) // gs
    out = np.zeros_like(x); sd = np.zeros(ng); mu = np.zeros(ng)
    L.group_norm(C.cast(ref.mats(x), ref.PM), C.cast(ref.mats(out), ref.PM), sd.ctypes.data_as(PD), mu.ctypes.data_as(PD), c, gs)
    return out, sd, mu


def ref_group_norm_ddx(src, data, mu, sd, gs):
    dest = np.zeros_like(src)
    L.group_norm_ddx(C.cast(ref.mats(src), ref.PM), C.cast(ref.mats(dest), ref.PM), C.cast(ref.mats(data), ref.PM), mu.ctypes.data_as(PD),
                     sd.ctypes.data_as(PD), src.shape[0], gs)
    return dest


def resnet_inputs(i, cin, cout, hh, tdim):
    sd = 6000 + 30 * i
    return dict(x=uniform(sd, (cin, hh, hh), -1, 1), temb=uniform(sd + 1, (1, tdim), 0, 1), k1=uniform(sd + 2, (cout, cin, 3, 3), -0.2, 0.2),
                k2=uniform(sd + 3, (cout, cout, 3, 3), -0.1, 0.1), tw=uniform(sd + 4, (tdim, cout), -0.1, 0.1), tb=uniform(sd + 5, (1, cout), -0.1, 0.1),
                kres=uniform(sd + 6, (cout, cin, 1, 1), -0.3, 0.3) if cin != cout else None, del_out=uniform(sd + 7, (cout, hh, hh), -1, 1))


def gen_resnet():
    U = C.CDLL(os.path.join(HERE, "_ref", "libref_unet.so")); libc = C.CDLL(None)
    L.group_norm.argtypes = [ref.PM, ref.PM, PD, PD, C.c_int, C.c_int]
    L.group_norm_ddx.argtypes = [ref.PM, ref.PM, ref.PM, PD, PD, C.c_int, C.c_int]
    d = {}
    cfgs = [(3, 8, 6, 8, 32), (8, 8, 5, 16, 4), (64, 32, 8, 32, 32)]      # (Cin, Cout, H=W, time dim, group size)
    d["cfgs"] = np.array(cfgs, np.int64)
    for i, (cin, cout, hh, tdim, gs) in enumerate(cfgs):
        I = resnet_inputs(i, cin, cout, hh, tdim)
        x = I["x"]
        relu1, sd1, mu1 = ref_group_norm(x, gs); relu1 = ref.data_fn("relu", relu1, relu1.size)
        c1, im1, km1 = ref_conv_fwd(relu1, I["k1"])
        tdense = ref.inplace2("matrix_add", ref.matmul_inplace(I["temb"], I["tw"]), I["tb"])
        U._add_time_embedding(ref.mats(c1), C.byref(ref.mat(tdense)), cout)
        relu2, sd2, mu2 = ref_group_norm(c1, gs); relu2 = ref.data_fn("relu", relu2, relu2.size)
        libc.srand(77 + i); dp = np.zeros_like(relu2)
        U._dropout(ref.mats(relu2), ref.mats(dp), cout)
        libc.srand(77 + i); libc.rand.restype = C.c_int
        draws = np.array([np.float32(libc.rand()) / np.float32(2147483647) for _ in range(relu2.size)], np.float32)
        c2, im2, km2 = ref_conv_fwd(dp, I["k2"])
        res = x
        if cin != cout:
            res, imr, kmr = ref_conv_fwd(x, I["kres"])
        result = c2 + res                                                           # :1067-1071 (one add per element)
        fwd = dict(mu1=mu1, sd1=sd1, relu1=relu1, c1=c1, tdense=tdense.ravel(), mu2=mu2, sd2=sd2, relu2=relu2, dp=dp, c2=c2, result=result)
        if cin != cout:
            fwd["res"] = res
        d[f"r{i}_dropped"] = (draws < np.float32(0.1)).astype(np.uint8)
        for n, v_ in fwd.items():
            put(d, f"r{i}_{n}", v_, 8192)
        # backward
        dk2, g_dp = ref_conv_bwd(I["del_out"], im2, km2, cout, 3)
        U._dropout_mask(ref.mats(g_dp), ref.mats(dp), cout)
        g_relu2 = np.zeros_like(g_dp); U.multi_channel_relu_ddx(ref.mats(g_dp), ref.mats(g_relu2), ref.mats(relu2), cout)
        g_c1 = ref_group_norm_ddx(g_relu2, c1, mu2, sd2, gs)
        dtb = np.array([[sum_seq(g_c1[c].ravel()) for c in range(cout)]])
        dtw = ref.matmul_inplace(ref.inplace1("matrix_transpose", I["temb"]), dtb)
        dk1, g_relu1 = ref_conv_bwd(g_c1, im1, km1, cin, 3)
        U.multi_channel_relu_ddx(ref.mats(g_relu1), ref.mats(g_relu1), ref.mats(relu1), cin)
        del_x = ref_group_norm_ddx(g_relu1, x, mu1, sd1, gs)
        bwd = dict(dk1=dk1, dk2=dk2, dtw=dtw, dtb=dtb.ravel())
        if cin != cout:
            dkres, g_res = ref_conv_bwd(I["del_out"], imr, kmr, cin, 1)
            bwd["dkres"] = dkres
            del_x = np.stack([ref.inplace2("matrix_add", del_x[c], g_res[c]) for c in range(cin)])
        else:
            del_x = np.stack([ref.inplace2("matrix_add", del_x[c], I["del_out"][c]) for c in range(cin)])
        bwd["del_x"] = del_x
        for n, v_ in bwd.items():
            put(d, f"r{i}_{n}", v_, 8192)
    np.savez_compressed(os.path.join(GOLD, "resnet.npz"), **d)


# ---- lib/layer.c through the reference itself -----------------------------------------------------------------
class RefLayer(C.Structure):
    """struct Layer, lib/layer.h:4-15 (reference build: Matrix.data is double*)."""


RefLayer._fields_ = [("num_nodes", C.c_int), ("nodes", ref.PM), ("raw_nodes", ref.PM), ("weights", ref.PM), ("biases", ref.PM),
                     ("previous_layer", C.POINTER(RefLayer)), ("activation", C.c_void_p), ("activation_ddx", C.c_void_p),
                     ("has_previous_layer", C.c_char), ("has_nodes", C.c_char)]
# the reference declares the callbacks void(*)(float*, int) and hands them Matrix.data, which is double* in its own build (Q4):
# the harness's callbacks take the pointer for what it is
ACT64 = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int)


def ref_layer_net(sizes, weights, biases, x, act, act_ddx, expectations, lr):
    """Builds the chain input -> ... -> output the way main.c:52-73 does, runs feed_forward on every non-input layer and then
    back_propagate_errors on the output layer; returns per layer (nodes, raw_nodes) after the forward pass and (weights, biases)
    after the update.  Matrices are made with the reference's make_matrix on malloc'd buffers (free_layer_data releases them)."""
    libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]
    L.make_matrix.restype = ref.PM; L.make_matrix.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.feed_forward.argtypes = [C.POINTER(RefLayer)]
    L.back_propagate_errors.argtypes = [C.POINTER(RefLayer), C.POINTER(C.c_float), C.c_float]
    L.free_layer_data.argtypes = [RefLayer]

    def heap(a):
        a = np.ascontiguousarray(a, np.float64)
        p = libc.malloc(a.nbytes); C.memmove(p, a.ctypes.data, a.nbytes)
        return L.make_matrix(a.shape[0], a.shape[1], C.cast(p, C.POINTER(C.c_double)))
    fa, fd = ACT64(act), ACT64(act_ddx)
    layers = [RefLayer(sizes[0], heap(x), None, None, None, None, None, None, b"\x00", b"\x01")]
    for i in range(1, len(sizes)):
        layers.append(RefLayer(sizes[i], None, None, heap(weights[i - 1]), heap(biases[i - 1]), C.pointer(layers[i - 1]),
                               C.cast(fa, C.c_void_p), C.cast(fd, C.c_void_p), b"\x01", b"\x00"))
    for l in layers[1:]:
        L.feed_forward(C.byref(l))
    arr = lambda pm: np.ctypeslib.as_array(pm.contents.data, shape=(pm.contents.rows, pm.contents.cols)).copy()
    fwd = [(arr(l.nodes), arr(l.raw_nodes)) for l in layers[1:]]
    e = np.ascontiguousarray(expectations, np.float32)
    L.back_propagate_errors(C.byref(layers[-1]), e.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(lr))
    upd = [(arr(l.weights), arr(l.biases)) for l in layers[1:]]
    for l in reversed(layers[1:]):
        L.free_layer_data(l)
    L.free_matrix(layers[0].nodes)
    return fwd, upd


def gen_layer():
    """lib/layer.c:6-107.  Case "main": main.c:52-87 verbatim (3-2-2 net from data/inputs.csv, weights.csv, biases.csv -- the output
    layer loads the same files, i.e. the first 4 weights; activation x0.1, derivative 0.1, expectations {0.5, 0.5}, lr 0.05).
    Case "mlp": 12-7-5-3 net, leaky-ReLU-like callbacks, random values."""
    d = {}

    def act_main(p, n):
        for i in range(n):
            p[i] *= 0.1

    def ddx_main(p, n):
        for i in range(n):
            p[i] = 0.1
    x = read_csv(f"{REFROOT}/data/inputs.csv", 3).astype(np.float64).reshape(3, 1)
    wv = read_csv(f"{REFROOT}/data/weights.csv", 6).astype(np.float64)
    bv = read_csv(f"{REFROOT}/data/biases.csv", 2).astype(np.float64)
    ws = [wv.reshape(2, 3), wv[:4].reshape(2, 2)]; bs = [bv.reshape(2, 1), bv.reshape(2, 1)]
    fwd, upd = ref_layer_net([3, 2, 2], ws, bs, x, act_main, ddx_main, [0.5, 0.5], 0.05)
    for f in ("inputs", "weights", "biases"):      # the data files themselves (data, not source): the host-layer test loads them through load_*_from_csv
        d[f"main_{f}_csv"] = np.frombuffer(open(f"{REFROOT}/data/{f}.csv", "rb").read(), np.uint8)
    d["main_x"] = x; d["main_lr"] = np.array(np.float32(0.05), np.float64); d["main_expect"] = np.array([0.5, 0.5])
    for i in range(2):
        d[f"main_w{i}"], d[f"main_b{i}"] = ws[i], bs[i]
        d[f"main_nodes{i}"], d[f"main_raw{i}"] = fwd[i]
        d[f"main_w{i}_new"], d[f"main_b{i}_new"] = upd[i]

    def act_leaky(p, n):
        for i in range(n):
            if p[i] < 0:
                p[i] *= 0.25

    def ddx_leaky(p, n):
        for i in range(n):
            p[i] = 1.0 if p[i] > 0 else 0.25
    sizes = [12, 7, 5, 3]
    f32 = lambda a: a.astype(np.float32).astype(np.float64)          # fp32-representable inputs: the device build holds exactly these
    ws = [f32(uniform(9100 + i, (sizes[i + 1], sizes[i]), -0.7, 0.7)) for i in range(3)]
    bs = [f32(uniform(9200 + i, (sizes[i + 1], 1), -0.3, 0.3)) for i in range(3)]
    x = f32(uniform(9300, (12, 1), -1, 1)); e = f32(uniform(9301, (3,), 0, 1))
    fwd, upd = ref_layer_net(sizes, ws, bs, x, act_leaky, ddx_leaky, e, 0.125)
    d["mlp_sizes"] = np.array(sizes, np.int64); d["mlp_x"] = x; d["mlp_expect"] = e; d["mlp_lr"] = np.array(0.125)
    for i in range(3):
        d[f"mlp_w{i}"], d[f"mlp_b{i}"] = ws[i], bs[i]
        d[f"mlp_nodes{i}"], d[f"mlp_raw{i}"] = fwd[i]
        d[f"mlp_w{i}_new"], d[f"mlp_b{i}_new"] = upd[i]
    np.savez_compressed(os.path.join(GOLD, "layer.npz"), **d)


def capture_stdout(fn):
    """Runs fn() with file descriptor 1 pointing at a temporary file; returns the bytes the C library wrote."""
    import tempfile
    libc = C.CDLL(None)
    sys.stdout.flush(); libc.fflush(None)
    with tempfile.TemporaryFile() as tmp:
        saved = os.dup(1)
        os.dup2(tmp.fileno(), 1)
        try:
            fn(); libc.fflush(None)
        finally:
            os.dup2(saved, 1); os.close(saved)
        tmp.seek(0)
        return tmp.read()


def gen_print():
    """print_matrix / print_matrix_dim, lib/matrix.c:71-93: the reference's own stdout for matrices that hit every branch -- exact zero,
    negative values (all of which fall into the `< 0.01` branch and print as %.2e), small positives, values >= 0.01 -- and the
    two error messages that precede exit(1) are pinned in tests/c/host_errors.c.  Values are fp32-representable, so the float build
    prints the same digits."""
    L.print_matrix.argtypes = [ref.Matrix]; L.print_matrix_dim.argtypes = [ref.Matrix]
    mats_ = {
        "zeros": np.zeros((2, 3)),
        "mixed": np.array([[0.0, -3.25, 0.0078125, 0.5], [1234.5, -0.001953125, 0.015625, 100.0], [0.009765625, 0.25, -1e3, 2.0 ** -20]]),
        "main_kat": ref.matmul(np.array([[1, 2, 3], [4, 5, 6]], np.float64), np.array([[1, 0.5], [0.25, 1], [0, 2]], np.float64)),
        "column": np.array([[2.5], [-0.125], [0.0]]),
        "row": np.array([[0.01171875, 7.0, 65536.0, -65536.0]]),
    }
    d = {}
    for name, a in mats_.items():
        a = np.ascontiguousarray(a, np.float64)
        assert np.array_equal(a.astype(np.float32).astype(np.float64), a), name
        m = ref.mat(a)
        d[name] = a
        d[name + "__stdout"] = np.frombuffer(capture_stdout(lambda: (L.print_matrix(m), L.print_matrix_dim(m))), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "print_matrix.npz"), **d)


def sum_seq(v):
    """left-to-right fp64 sum, the order of the loop at model/cifar_unet.c:1191-1196"""
    s_ = 0.0
    for t in v:
        s_ += float(t)
    return s_


if __name__ == "__main__":
    assert ref.available(), "build oracle/_ref first: make -C oracle"
    os.makedirs(GOLD, exist_ok=True)
    only = sys.argv[1:]          # e.g. `gen_golden.py layer print` regenerates just those files
    gens = {"gemm": gen_gemm, "matrix_ops": gen_matrix_ops, "conv": gen_conv, "norm": gen_norm, "mnist": gen_mnist, "unet_glue": gen_unet_glue,
            "attention": gen_attention, "resnet": gen_resnet, "layer": gen_layer, "print": gen_print}
    for name, fn in gens.items():
        if not only or name in only:
            fn()
    tot = sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD))
    print("golden vectors written to", GOLD, f"({tot/1e6:.2f} MB)")

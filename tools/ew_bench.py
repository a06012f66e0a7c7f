#!/usr/bin/env python3
"""HBM roofline of the elementwise / reduction / reshape kernels: GB/s of ALGORITHMIC bytes
against the 8 TB/s HBM3E spec, timed with HIP events on the launch stream."""
import ctypes as C, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import time
import oracle   # CPU restatement of the reference loops: the baseline timed beside each kernel (checker, never the product)
from __graft_entry__ import load_pkg
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
st = L.bla_default_stream()
e0, e1 = C.c_void_p(), C.c_void_p(); chk(L.bla_event_create(C.byref(e0))); chk(L.bla_event_create(C.byref(e1)))
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
Cc = R
n = R * Cc
rng = np.random.default_rng(0)
a = bla.to_device(rng.uniform(-1, 1, (R, Cc)).astype(np.float32)); b = bla.to_device(rng.uniform(-1, 1, (R, Cc)).astype(np.float32))
o = bla.empty((R, Cc)); small = bla.empty((max(R, Cc),))

def timeit(fn, iters=10):
    fn(); fn()
    chk(L.bla_event_record(e0, st))
    for _ in range(iters): fn()
    chk(L.bla_event_record(e1, st))
    ms = C.c_float(); chk(L.bla_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value / iters * 1e-3

ha = rng.uniform(-1, 1, (R, Cc)); hb = rng.uniform(-1, 1, (R, Cc)); hbias_c = rng.uniform(-1, 1, (R, 1)); hbias_r = rng.uniform(-1, 1, (1, Cc))
cpu = {   # the reference's own loop (fp64, 1 core, gcc -O2) on the same shape
    "matrix_scale": lambda: oracle.scale(ha, 1.0001), "matrix_add": lambda: oracle.add(ha, hb), "hadamard": lambda: oracle.hadamard(ha, hb),
    "sgd_axpy": lambda: oracle.add(ha, oracle.scale(hb, 1e-9)), "relu": lambda: oracle.relu(ha), "transpose": lambda: oracle.transpose(ha),
    "add_tile_columns": lambda: oracle.add_tile_columns(ha, hbias_c), "add_tile_rows": lambda: oracle.add_tile_rows(ha, hbias_r),
    "frobenius_norm": lambda: oracle.frobenius(ha), "row_sum": lambda: oracle.row_sum(ha), "col_sum_intended": lambda: oracle.col_sum_intended(ha),
    "softmax_cols": lambda: oracle.softmax_cols(ha), "softmax_rows": lambda: oracle.softmax_rows(ha),
}
cases = [
    ("matrix_scale", 8, lambda: chk(L.bla_scale_f32(st, a.ptr, n, 1.0001))),
    ("matrix_add", 12, lambda: chk(L.bla_add_f32(st, a.ptr, b.ptr, n))),
    ("hadamard", 12, lambda: chk(L.bla_hadamard_f32(st, a.ptr, b.ptr, n))),
    ("sgd_axpy", 12, lambda: chk(L.bla_axpy_f32(st, a.ptr, b.ptr, 1e-9, n))),
    ("relu", 8, lambda: chk(L.bla_relu_f32(st, a.ptr, n))),
    ("transpose", 8, lambda: chk(L.bla_transpose_f32(st, a.ptr, o.ptr, R, Cc))),
    ("add_tile_columns", 8, lambda: chk(L.bla_add_tile_columns_f32(st, a.ptr, R, Cc, small.ptr, 1))),
    ("add_tile_rows", 8, lambda: chk(L.bla_add_tile_rows_f32(st, a.ptr, R, Cc, small.ptr))),
    ("frobenius_norm", 4, lambda: chk(L.bla_frobenius_f32(st, a.ptr, n, small.ptr))),
    ("row_sum", 4, lambda: chk(L.bla_row_sum_f32(st, a.ptr, R, Cc, small.ptr))),
    ("col_sum_intended", 4, lambda: chk(L.bla_col_sum_f32(st, a.ptr, R, Cc, small.ptr, 1))),
    ("softmax_cols", 8, lambda: chk(L.bla_softmax_cols_f32(st, b.ptr, R, Cc))),
    ("softmax_rows", 8, lambda: chk(L.bla_softmax_rows_f32(st, b.ptr, R, Cc))),
]
out = {}
for name, bpe, fn in cases:
    t = timeit(fn)
    gbs = bpe * n / t / 1e9
    t0 = time.perf_counter(); cpu[name](); tc = time.perf_counter() - t0      # includes the wrapper's input copy for in-place ops
    out[name] = {"GB/s": round(gbs, 1), "frac_of_8TB/s": round(gbs / 8000, 3), "us": round(t * 1e6, 1), "bytes_per_element": bpe,
                 "cpu_ref_ms_1core_fp64": round(tc * 1e3, 1)}
    print(f"{name:<20} {R}x{Cc}  {t*1e6:9.1f} us  {gbs:8.1f} GB/s algorithmic  {gbs/8000*100:5.1f}% of 8 TB/s   | CPU reference loop {tc*1e3:8.1f} ms (1 core, fp64)", flush=True)
# conv stages at the U-Net's largest layer, batched 64 images worth of work is out of the single-image API: single image numbers
h = w = 32; cin = 128; k = 3
x = bla.to_device(rng.uniform(-1, 1, (cin, h, w)).astype(np.float32)); im = bla.empty((h * w, k * k * cin))
t = timeit(lambda: chk(L.bla_im2col_f32(st, x.ptr, im.ptr, h, w, k, cin, 1)), 50)
by = 4 * (cin * h * w + h * w * k * k * cin)
print(f"{'im2col 128x32x32 k3':<20} {t*1e6:9.1f} us  {by/t/1e9:8.1f} GB/s algorithmic (single image: {by/1e6:.1f} MB, latency-bound)")
out["im2col_128x32x32"] = {"GB/s": round(by / t / 1e9, 1), "us": round(t * 1e6, 1)}
gn_o = bla.empty((cin, h, w)); sd = bla.empty((4,)); mu = bla.empty((4,))
t = timeit(lambda: chk(L.bla_group_norm_f32(st, x.ptr, gn_o.ptr, sd.ptr, mu.ptr, cin, 32, h * w)), 50)
by = 8 * cin * h * w
print(f"{'group_norm 128x32x32':<20} {t*1e6:9.1f} us  {by/t/1e9:8.1f} GB/s algorithmic (4 groups = 4 workgroups: latency-bound)")
out["group_norm_128x32x32"] = {"GB/s": round(by / t / 1e9, 1), "us": round(t * 1e6, 1)}
print(json.dumps(out))

#!/usr/bin/env python3
"""Config 5: the U-Net's convolution shapes, one image, forward + both gradients:
staged path (bla_conv_forward/backward: im2col + transposes + GEMMs + col2im, fills every ConvData workspace) versus
the implicit-GEMM path (bla_conv2d_*: nothing materialised).  FLOPs = 2*M*K*N per product (3 products)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import time
import oracle   # CPU restatement (checker): the reference's conv()/conv_ddx() timed beside the kernels
from __graft_entry__ import load_pkg
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
st = L.bla_default_stream()
e0, e1 = C.c_void_p(), C.c_void_p(); chk(L.bla_event_create(C.byref(e0))); chk(L.bla_event_create(C.byref(e1)))
rng = np.random.default_rng(0)

def timeit(fn, iters=30):
    fn(); fn()
    chk(L.bla_event_record(e0, st))
    for _ in range(iters): fn()
    chk(L.bla_event_record(e1, st))
    ms = C.c_float(); chk(L.bla_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value / iters * 1e-3

single = [] if os.environ.get("CONV_BENCH_BATCH_ONLY") == "1" else [(32, 128, 128, 3, 1), (16, 256, 256, 3, 1), (8, 256, 256, 3, 1), (4, 256, 256, 3, 1), (32, 3, 128, 3, 1), (32, 256, 128, 3, 1), (32, 128, 256, 3, 2)]
for (h, cin, cout, k, s) in single:
    w = h; ho = -(-h // s); hw = ho * ho; kkc = k * k * cin
    x = bla.to_device(rng.uniform(-1, 1, (cin, h, w)).astype(np.float32)); kern = bla.to_device(rng.uniform(-.1, .1, (cout, cin, k, k)).astype(np.float32))
    dy = bla.to_device(rng.uniform(-1, 1, (cout, ho, ho)).astype(np.float32))
    im, km, pr, out = bla.empty((hw, kkc)), bla.empty((kkc, cout)), bla.empty((hw, cout)), bla.empty((cout, ho, ho))
    dq, dkm, dk, dcol, dx, scr = bla.empty((hw, cout)), bla.empty((kkc, cout)), bla.empty((cout, cin, k, k)), bla.empty((hw, kkc)), bla.empty((cin, h, w)), bla.empty((cout * kkc,))
    fl = 2.0 * hw * kkc * cout
    t_sf = timeit(lambda: chk(L.bla_conv_forward_f32(st, x.ptr, kern.ptr, im.ptr, km.ptr, pr.ptr, out.ptr, h, w, k, cin, cout, s)))
    t_if = timeit(lambda: chk(L.bla_conv2d_forward_f32(st, x.ptr, kern.ptr, out.ptr, h, w, k, cin, cout, s)))
    line = f"{cin:>3}->{cout:<3} {h}x{h} k{k} s{s} (M,K,N)=({hw},{kkc},{cout})  fwd staged {t_sf*1e6:7.1f} us  implicit {t_if*1e6:7.1f} us ({fl/t_if/1e12:5.2f} TF/s)"
    if s == 1:
        t_sb = timeit(lambda: chk(L.bla_conv_backward_f32(st, dy.ptr, im.ptr, km.ptr, dq.ptr, dkm.ptr, dk.ptr, dcol.ptr, dx.ptr, h, w, k, cin, cout, 1)))
        t_ib = timeit(lambda: chk(L.bla_conv2d_backward_f32(st, dy.ptr, x.ptr, kern.ptr, dk.ptr, dx.ptr, scr.ptr, h, w, k, cin, cout, 1)))
        line += f"   bwd staged {t_sb*1e6:7.1f} us  implicit {t_ib*1e6:7.1f} us ({2*fl/t_ib/1e12:5.2f} TF/s)"
    if cin * cout * hw <= 128 * 128 * 1024:      # CPU reference (fp64, 1 core): forward and, at stride 1, backward
        hx = rng.uniform(-1, 1, (cin, h, w)); hk = rng.uniform(-.1, .1, (cout, cin, k, k))
        t0 = time.perf_counter(); fw = oracle.conv_intended(hx, hk, s); tcf = time.perf_counter() - t0
        line += f"   | CPU ref fwd {tcf*1e3:7.1f} ms"
        if s == 1:
            t0 = time.perf_counter(); oracle.conv_ddx_intended(rng.uniform(-1, 1, (cout, ho, ho)), fw["im2col"], fw["kmat"], cin, k); tcb = time.perf_counter() - t0
            line += f" bwd {tcb*1e3:7.1f} ms"
    print(line, flush=True)

# batch of 64 images through the same kernels (SURVEY 8(d) cfg 5: "batch-of-64 images (M x 64)"): one launch per pass
print("\nbatch of 64 images, implicit-GEMM path (one launch per pass; FLOPs = 64 x the single-image figure)", flush=True)
B = 64
for (h, cin, cout, k, s) in [(32, 128, 128, 3, 1), (16, 256, 256, 3, 1), (8, 256, 256, 3, 1), (4, 256, 256, 3, 1), (32, 128, 256, 3, 2), (16, 256, 256, 3, 2)]:
    w = h; ho = -(-h // s); hw = ho * ho; kkc = k * k * cin
    x = bla.to_device(rng.uniform(-1, 1, (B, cin, h, w)).astype(np.float32)); kern = bla.to_device(rng.uniform(-.1, .1, (cout, cin, k, k)).astype(np.float32))
    dy = bla.to_device(rng.uniform(-1, 1, (B, cout, ho, ho)).astype(np.float32))
    out, dk, dx, scr = bla.empty((B, cout, ho, ho)), bla.empty((cout, cin, k, k)), bla.empty((B, cin, h, w)), bla.empty((cout * kkc,))
    fl = 2.0 * hw * kkc * cout * B
    t_f = timeit(lambda: chk(L.bla_conv2d_forward_batched_f32(st, x.ptr, kern.ptr, out.ptr, B, h, w, k, cin, cout, s)), iters=10)
    line = f"{cin:>3}->{cout:<3} {h}x{h} k{k} s{s} x{B}  fwd {t_f*1e6:8.1f} us ({fl/t_f/1e12:6.2f} TF/s, {fl/t_f/1e12/157.3*100:4.1f}% of fp32 MFMA peak)"
    t_b = timeit(lambda: chk(L.bla_conv2d_backward_batched_f32(st, dy.ptr, x.ptr, kern.ptr, dk.ptr, dx.ptr, scr.ptr, B, h, w, k, cin, cout, s)), iters=10)
    line += f"   bwd (dkern + dx) {t_b*1e6:8.1f} us ({2*fl/t_b/1e12:6.2f} TF/s, {2*fl/t_b/1e12/157.3*100:4.1f}%)"
    print(line, flush=True)

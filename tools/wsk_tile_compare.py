#!/usr/bin/env python3
"""The wave-split-K kernel on 32x32 tiles (config 6) against 16x16 tiles (config 16) on the eight products of one MNIST-NN step
(model/mnist_nn.c:221-292 at batch 256) and a few conv / attention shapes: HIP-event time per launch, back to back."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_pkg
from inputs import uniform
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
st = L.bla_default_stream()
e0, e1 = C.c_void_p(), C.c_void_p(); chk(L.bla_event_create(C.byref(e0))); chk(L.bla_event_create(C.byref(e1)))

def timeit(fn, iters=200):
    for _ in range(20): fn()
    chk(L.bla_event_record(e0, st))
    for _ in range(iters): fn()
    chk(L.bla_event_record(e1, st))
    ms = C.c_float(); chk(L.bla_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value / iters * 1e3

B = 256
cases = [("fwd1 W1.X", 256, B, 784, 0, 0), ("fwd2 W2.A1", 128, B, 256, 0, 0), ("fwd3 W3.A2", 10, B, 128, 0, 0), ("dZ2 W3^T.dZ3", 128, B, 10, 1, 0),
         ("dW3 dZ3.A2^T", 10, 128, B, 0, 1), ("dZ1 W2^T.dZ2", 256, B, 128, 1, 0), ("dW2 dZ2.A1^T", 128, 256, B, 0, 1), ("dW1 dZ1.X^T", 256, 784, B, 0, 1),
         ("attention QK^T", 256, 256, 16, 0, 1), ("conv 8x8 product", 64, 256, 1152, 0, 0)]
for name, m, n, k, ta, tb in cases:
    k_eff = k
    a = bla.to_device(uniform(1, (k, m) if ta else (m, k), dtype=np.float32)); b = bla.to_device(uniform(2, (n, k) if tb else (k, n), dtype=np.float32))
    c = bla.empty((m, n))
    out = []
    for cfg in (6, 16):
        L.bla_gemm_set_config(cfg, 0)
        try:
            t = timeit(lambda: bla.gemm(a, b, c, transa=bool(ta), transb=bool(tb), stream=st))
            out.append(f"{t:6.2f} us ({L.bla_gemm_last_kernel().decode().split('_')[2]})")
        except bla.BlaError as ex:
            out.append("   n/a")
    L.bla_gemm_set_config(-1, 0)
    print(f"{name:<18} {m:>4}x{k:<4}x{n:<4} {'T' if ta else 'N'}{'T' if tb else 'N'}   32x32 {out[0]}   16x16 {out[1]}", flush=True)

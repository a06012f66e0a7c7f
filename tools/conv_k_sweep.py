#!/usr/bin/env python3
"""Fixed cost and slope of the batched convolution kernels: forward of C_in -> C_out, k3 s1, batch 64, for growing C_in (K = 9 C_in) at a fixed
output size.  Run under `rocprofv3 --kernel-trace --stats` to read the gather kernel's own duration per K (tools/measure_r03.sh); the printed
figures are event timings of the whole call (padding / reordering kernels included)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
st = L.bla_default_stream()
e0, e1 = C.c_void_p(), C.c_void_p(); chk(L.bla_event_create(C.byref(e0))); chk(L.bla_event_create(C.byref(e1)))
rng = np.random.default_rng(0)
B = 64
shapes = [(int(a), int(b)) for a, b in (s.split("x") for s in (sys.argv[1:] or ["32x128", "16x256"]))]      # side x C_out
for h, cout in shapes:
    for cin in (16, 32, 64, 128, 256, 512):
        x = bla.to_device(rng.uniform(-1, 1, (B, cin, h, h)).astype(np.float32)); kern = bla.to_device(rng.uniform(-.1, .1, (cout, cin, 3, 3)).astype(np.float32))
        out = bla.empty((B, cout, h, h))
        fn = lambda: chk(L.bla_conv2d_forward_batched_f32(st, x.ptr, kern.ptr, out.ptr, B, h, h, 3, cin, cout, 1))
        fn(); fn()
        chk(L.bla_event_record(e0, st))
        for _ in range(10): fn()
        chk(L.bla_event_record(e1, st))
        ms = C.c_float(); chk(L.bla_event_elapsed_ms(e0, e1, C.byref(ms)))
        t = ms.value / 10 * 1e-3; fl = 2.0 * h * h * 9 * cin * cout * B
        print(f"{cin:>3}->{cout:<3} {h}x{h} x{B}: K {9 * cin:5d}  {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TF/s", flush=True)

#!/usr/bin/env python3
"""Do the weight gradient and the data gradient of one convolution gain from running CONCURRENTLY (two contexts = two streams, two workspaces) instead of one
after the other?  Both read del_y and neither reads what the other writes.  Wall time of N back-to-back (wgrad, dgrad) pairs, batch 64."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
rng = np.random.default_rng(0)
B = 64
ctx2 = C.c_void_p(); chk(L.bla_context_create(C.byref(ctx2), 0))
for (h, cin, cout) in [(32, 128, 128), (16, 256, 256), (8, 256, 256)]:
    kkc = 9 * cin
    x = bla.to_device(rng.uniform(-1, 1, (B, cin, h, h)).astype(np.float32)); kern = bla.to_device(rng.uniform(-.1, .1, (cout, cin, 3, 3)).astype(np.float32))
    dy = bla.to_device(rng.uniform(-1, 1, (B, cout, h, h)).astype(np.float32))
    dk, dx, scr = bla.empty((cout, cin, 3, 3)), bla.empty((B, cin, h, h)), bla.empty((cout * kkc,))
    def wgrad(): chk(L.bla_conv2d_backward_batched_f32(None, dy.ptr, x.ptr, kern.ptr, dk.ptr, None, scr.ptr, B, h, h, 3, cin, cout, 1))
    def dgrad(): chk(L.bla_conv2d_backward_batched_f32(None, dy.ptr, x.ptr, kern.ptr, None, dx.ptr, scr.ptr, B, h, h, 3, cin, cout, 1))
    def both(): chk(L.bla_conv2d_backward_batched_f32(None, dy.ptr, x.ptr, kern.ptr, dk.ptr, dx.ptr, scr.ptr, B, h, h, 3, cin, cout, 1))
    def sync_all():
        chk(L.bla_context_set_current(None)); chk(L.bla_stream_sync(None))
        chk(L.bla_context_set_current(ctx2)); chk(L.bla_stream_sync(None)); chk(L.bla_context_set_current(None))
    N = 40
    for _ in range(3): both()
    sync_all(); t0 = time.perf_counter()
    for _ in range(N): both()
    sync_all(); t_seq = (time.perf_counter() - t0) / N
    def pair():
        chk(L.bla_context_set_current(None)); wgrad()
        chk(L.bla_context_set_current(ctx2)); dgrad()
    for _ in range(3): pair()
    sync_all(); t0 = time.perf_counter()
    for _ in range(N): pair()
    sync_all(); t_par = (time.perf_counter() - t0) / N
    chk(L.bla_context_set_current(None))
    print(f"{cin}->{cout} {h}x{h} x{B}: one stream {t_seq * 1e6:7.1f} us   two streams {t_par * 1e6:7.1f} us", flush=True)

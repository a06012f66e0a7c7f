#!/usr/bin/env python3
"""Would a backward pass gain from running its weight gradients as a chain of their own beside everything else?  Chain A = what stays on the main stream of a
ResNet block (data gradient + the norm gradient, HBM-bound, behind it), chain B = the weight gradients; each alone, one after the other, and side by side on
two contexts (two streams, two workspaces), N links each."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
rng = np.random.default_rng(0)
B, N = 64, 30
ctx2 = C.c_void_p(); chk(L.bla_context_create(C.byref(ctx2), 0))
def sync_all():
    chk(L.bla_context_set_current(None)); chk(L.bla_stream_sync(None))
    chk(L.bla_context_set_current(ctx2)); chk(L.bla_stream_sync(None)); chk(L.bla_context_set_current(None))
for (h, c) in [(32, 128), (16, 256), (8, 256)]:
    kkc = 9 * c; hw = h * h; groups = B * c // 32
    x = bla.to_device(rng.uniform(0, 1, (B, c, h, h)).astype(np.float32)); kern = bla.to_device(rng.uniform(-.1, .1, (c, c, 3, 3)).astype(np.float32))
    dy = bla.to_device(rng.uniform(-1, 1, (B, c, h, h)).astype(np.float32))
    dk, dx, dx2, scr = bla.empty((c, c, 3, 3)), bla.empty((B, c, h, h)), bla.empty((B, c, h, h)), bla.empty((c * kkc,))
    mu, sd = bla.to_device(rng.uniform(-.1, .1, (groups,)).astype(np.float32)), bla.to_device(rng.uniform(.5, 1, (groups,)).astype(np.float32))
    def link_a():
        chk(L.bla_conv2d_backward_batched_f32(None, dy.ptr, x.ptr, kern.ptr, None, dx.ptr, scr.ptr, B, h, h, 3, c, c, 1))
        chk(L.bla_group_norm_ddx_gated_batched_f32(None, B, dx.ptr, dx2.ptr, x.ptr, mu.ptr, sd.ptr, c, 32, hw, x.ptr, None))
    def link_b(): chk(L.bla_conv2d_backward_batched_f32(None, dy.ptr, x.ptr, kern.ptr, dk.ptr, None, scr.ptr, B, h, h, 3, c, c, 1))
    def timed(fn):
        fn(); sync_all(); t0 = time.perf_counter(); fn(); sync_all(); return (time.perf_counter() - t0) / N * 1e6
    def only_a():
        for _ in range(N): link_a()
    def only_b():
        for _ in range(N): link_b()
    def seq():
        for _ in range(N): link_b(); link_a()
    def par():
        for _ in range(N):
            chk(L.bla_context_set_current(ctx2)); link_b()
            chk(L.bla_context_set_current(None)); link_a()
    ta, tb, ts, tp = timed(only_a), timed(only_b), timed(seq), timed(par)
    print(f"{c} ch {h}x{h} x{B}: data gradient + norm gradient {ta:7.1f} us | weight gradient {tb:7.1f} us | one stream {ts:7.1f} us | two streams {tp:7.1f} us", flush=True)

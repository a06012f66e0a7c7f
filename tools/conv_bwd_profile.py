#!/usr/bin/env python3
"""One convolution's gradients (bla_conv2d_backward_f32) in a loop, for rocprofv3 --kernel-trace --stats.
usage: conv_bwd_profile.py H CIN COUT [ITERS]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
h, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50
k = 3; st = L.bla_default_stream(); rng = np.random.default_rng(0)
x = bla.to_device(rng.uniform(-1, 1, (cin, h, h)).astype(np.float32)); kern = bla.to_device(rng.uniform(-.1, .1, (cout, cin, k, k)).astype(np.float32))
dy = bla.to_device(rng.uniform(-1, 1, (cout, h, h)).astype(np.float32))
dk, dx, scr = bla.empty((cout, cin, k, k)), bla.empty((cin, h, h)), bla.empty((cout * cin * k * k,))
for _ in range(iters):
    chk(L.bla_conv2d_backward_f32(st, dy.ptr, x.ptr, kern.ptr, dk.ptr, dx.ptr, scr.ptr, h, h, k, cin, cout, 1))
bla.sync(st)

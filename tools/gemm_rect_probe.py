#!/usr/bin/env python3
"""Plain fp32 products of the shapes the convolutions turn into (128 x 65536 x 1152, 256 x 16384 x 2304, 128 x 65536 x 2304) on the tile families that take them:
the ceiling of a 128-row tile without any gather (0.79-0.87 of peak), and what the dispatcher picks by itself.  Round 3: 168.6 / 154.9 / 153.3 us on configs 3 / 14 / 15
for the first shape -- the image-window convolution kernel does the same product in 169 us."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_pkg
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
rng = np.random.default_rng(0)
for (m, n, k) in [(128, 65536, 1152), (256, 16384, 2304), (128, 65536, 2304)]:
    a = bla.to_device(rng.uniform(-1, 1, (m, k)).astype(np.float32)); b = bla.to_device(rng.uniform(-1, 1, (k, n)).astype(np.float32)); c = bla.empty((m, n))
    for cfg in (3, 14, 15, 13, -1):      # the automatic choice last: the first measurement of a shape runs on cold clocks
        try:
            chk(L.bla_gemm_set_config(cfg, 0))
            for _ in range(5): chk(L.bla_gemm_f32(None, 0, 0, m, n, k, a.ptr, k, b.ptr, n, c.ptr, n, None))
            bla.sync(); t = time.perf_counter(); it = 40
            for _ in range(it): chk(L.bla_gemm_f32(None, 0, 0, m, n, k, a.ptr, k, b.ptr, n, c.ptr, n, None))
            bla.sync(); dt = (time.perf_counter() - t) / it
            print(f"{m}x{n}x{k} cfg {cfg:3d} {L.bla_gemm_last_kernel().decode():50s} {dt*1e6:8.1f} us {2*m*n*k/dt/1e12:7.1f} TF/s ({2*m*n*k/dt/157.3e12:.3f})", flush=True)
        except Exception as e:
            print(m, n, k, cfg, "n/a", str(e)[:80])
    chk(L.bla_gemm_set_config(-1, 0))

#!/usr/bin/env python3
"""One rectangular fp32 product in a loop (m n k [iters]) -- for counter passes on the headline kernel at a contraction short enough that
SQ_VALU_MFMA_BUSY_CYCLES (summed over 1,024 SIMDs) stays below the 2^31 at which the profiler's per-dispatch value saturates (at 4096^3 it does)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg
m, n, k = (int(v) for v in sys.argv[1:4]); iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
rng = np.random.default_rng(0)
a = bla.to_device(rng.uniform(-1, 1, (m, k)).astype(np.float32)); b = bla.to_device(rng.uniform(-1, 1, (k, n)).astype(np.float32)); c = bla.empty((m, n))
for _ in range(iters + 200):
    chk(L.bla_gemm_f32(None, 0, 0, m, n, k, a.ptr, k, b.ptr, n, c.ptr, n, None))
bla.sync()
print(L.bla_gemm_last_kernel().decode())

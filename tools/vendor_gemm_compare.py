#!/usr/bin/env python3
"""Context for the GEMM numbers: the vendor library's fp32 GEMM (rocBLAS / hipBLASLt behind torch.matmul, TF32-like modes off) timed on
the same box, same shapes, same HIP-event method, beside bla_gemm_f32.  Not part of the product path -- a yardstick only."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from __graft_entry__ import load_pkg
from inputs import uniform
torch.backends.cuda.matmul.allow_tf32 = False
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
st = L.bla_default_stream()
e0, e1 = C.c_void_p(), C.c_void_p(); chk(L.bla_event_create(C.byref(e0))); chk(L.bla_event_create(C.byref(e1)))
for n in [int(x) for x in (sys.argv[1:] or ["1024", "2048", "4096", "8192"])]:
    a = uniform(1, (n, n), dtype=np.float32); b = uniform(2, (n, n), dtype=np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    tc = torch.empty((n, n), device="cuda")
    for _ in range(10): torch.matmul(ta, tb, out=tc)
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20 if n <= 4096 else 5
    s0.record()
    for _ in range(iters): torch.matmul(ta, tb, out=tc)
    s1.record(); torch.cuda.synchronize()
    tv = s0.elapsed_time(s1) / iters * 1e-3
    da, db, dc = bla.to_device(a), bla.to_device(b), bla.empty((n, n))
    for _ in range(10): bla.gemm(da, db, dc, stream=st)
    chk(L.bla_event_record(e0, st))
    for _ in range(iters): bla.gemm(da, db, dc, stream=st)
    chk(L.bla_event_record(e1, st))
    ms = C.c_float(); chk(L.bla_event_elapsed_ms(e0, e1, C.byref(ms)))
    tm = ms.value / iters * 1e-3
    diff = float(np.abs(dc.numpy()[:8] - tc[:8].cpu().numpy()).max())
    fl = 2.0 * n ** 3
    print(f"N={n:5d}  vendor library (torch.matmul fp32) {tv*1e6:9.1f} us = {fl/tv/1e12:7.2f} TFLOP/s   |   bla_gemm_f32 {tm*1e6:9.1f} us = {fl/tm/1e12:7.2f} TFLOP/s"
          f"   (ratio {tv/tm:4.2f}x, max |difference| over 8 rows {diff:.2e})", flush=True)

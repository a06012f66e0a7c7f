#!/usr/bin/env python3
"""Practical ceiling of the fp32 matrix pipe: workgroups that issue nothing but independent v_mfma_f32_32x32x2_f32 (no loads, no LDS,
no barriers), for 1 / 2 / 4 waves per SIMD.  Whatever this reaches, a GEMM cannot exceed: peak FLOP/clk x the clock the chip holds
under an all-MFMA load, minus the pipe's own issue gaps."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
st = L.bla_default_stream()
e0, e1 = C.c_void_p(), C.c_void_p(); chk(L.bla_event_create(C.byref(e0))); chk(L.bla_event_create(C.byref(e1)))
sink = bla.empty((16,))
CUS = 256
for shape, sign in (("32x32x2", 1), ("16x16x4", -1)):
  for waves_per_simd in (1, 2, 4):
    waves = 4 * waves_per_simd            # one workgroup per CU
    iters = 20000 // waves_per_simd
    for rep in range(3):
        chk(L.bla_diag_mfma_rate(st, CUS, sign * waves, iters, sink.ptr))
    chk(L.bla_event_record(e0, st))
    for rep in range(5):
        chk(L.bla_diag_mfma_rate(st, CUS, sign * waves, iters, sink.ptr))
    chk(L.bla_event_record(e1, st))
    ms = C.c_float(); chk(L.bla_event_elapsed_ms(e0, e1, C.byref(ms)))
    t = ms.value / 5 * 1e-3
    flops = CUS * waves * iters * 8 * (2.0 * 32 * 32 * 2)   # both kernels do the FLOPs of 8 MFMAs 32x32x2 per trip
    print(f"v_mfma_f32_{shape}_f32, {waves_per_simd} wave(s) per SIMD: {flops/t/1e12:7.2f} TFLOP/s = {flops/t/1e12/157.3*100:5.1f}% of 157.3  ({t*1e3:.2f} ms per launch)", flush=True)

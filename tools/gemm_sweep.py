#!/usr/bin/env python3
"""Kernel-level GEMM timing sweep (HIP events), interleaved A/B rounds in one process.
usage: gemm_sweep.py [--sizes 1024,2048,4096] [--configs -1,0,2] [--layouts nn,nt,tn] [--rounds 5] [--iters 10]"""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_pkg
from inputs import uniform

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="1024,2048,4096,8192")
ap.add_argument("--configs", default="-1,0,2")
ap.add_argument("--layouts", default="nn")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--split", type=int, default=0)
ap.add_argument("--warm", type=float, default=1.0, help="seconds of 4096^3 products before the first measurement (the clocks ramp for about that long; 0 = cold)")
a = ap.parse_args()
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
st = L.bla_default_stream()
e0, e1 = C.c_void_p(), C.c_void_p(); chk(L.bla_event_create(C.byref(e0))); chk(L.bla_event_create(C.byref(e1)))
if a.warm > 0:
    import time
    wa = bla.to_device(uniform(1, (4096, 4096), dtype=np.float32)); wc = bla.empty((4096, 4096))
    t0 = time.time()
    while time.time() - t0 < a.warm:
        for _ in range(20):
            bla.gemm(wa, wa, wc, stream=st)
        bla.sync()
    del wa, wc
    print(f"# warmed up for {a.warm:.1f} s of 4096^3 products", flush=True)
for spec in a.sizes.split(","):
    dims = [int(x) for x in spec.split("x")]
    m, k, n = (dims * 3)[:3] if len(dims) == 1 else dims
    for lay in a.layouts.split(","):
        ta, tb = lay[0] == "t", lay[1] == "t"
        A = bla.to_device(uniform(1, (k, m) if ta else (m, k), dtype=np.float32))
        B = bla.to_device(uniform(2, (n, k) if tb else (k, n), dtype=np.float32))
        Cc = bla.empty((m, n))
        cfgs = [int(c) for c in a.configs.split(",")]
        res = {c: [] for c in cfgs}
        for r in range(a.rounds + 1):
            for c in cfgs:
                chk(L.bla_gemm_set_config(c, a.split))
                bla.gemm(A, B, Cc, transa=ta, transb=tb, stream=st)
                chk(L.bla_event_record(e0, st))
                for _ in range(a.iters):
                    bla.gemm(A, B, Cc, transa=ta, transb=tb, stream=st)
                chk(L.bla_event_record(e1, st))
                ms = C.c_float(); chk(L.bla_event_elapsed_ms(e0, e1, C.byref(ms)))
                if r > 0:
                    res[c].append(ms.value / a.iters)
                name = L.bla_gemm_last_kernel().decode()
                res.setdefault(("name", c), name)
        for c in cfgs:
            t = np.array(res[c]); fl = 2.0 * m * n * k
            print(f"{m}x{k}x{n} {lay} cfg{c:>2} {res[('name', c)]:<44} median {np.median(t)*1e3:9.1f} us  min {t.min()*1e3:9.1f} us  "
                  f"{fl/np.median(t)/1e9:8.2f} TF/s median  {fl/t.min()/1e9:8.2f} best  ({fl/np.median(t)/1e9/157.3*100:5.1f}% of 157.3)", flush=True)

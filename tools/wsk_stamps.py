#!/usr/bin/env python3
"""Diagnostics (never part of the product): builds libbla_hip with -DBLA_WSK_DIAG into gpurun_out/diag/, runs one
latency-bound GEMM and prints where workgroup 0 / wave 0 spends its cycles (s_memtime stamps).  The kernel currently carries three
stamps -- start, K loop done, after the barrier in front of the fold; per-chunk stamps (slots 1..19) print only if a build adds them."""
import ctypes as C, glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
out = os.path.join(ROOT, "gpurun_out", "diag"); os.makedirs(out, exist_ok=True)
csrc = os.path.join(ROOT, "big-linear-algebra_amd", "csrc")
objs = []
for src in sorted(glob.glob(os.path.join(csrc, "*.hip"))):
    o = os.path.join(out, os.path.basename(src)[:-4] + ".o"); objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DBLA_BUILDING", "-DBLA_WSK_DIAG",
                           "-I", os.path.join(ROOT, "include"), "-c", src, "-o", o])
so = os.path.join(out, "libbla_hip_diag.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
from __graft_entry__ import load_pkg
bla = load_pkg(); bla.native._SO = so
bla.init(0); L = bla.lib()
from inputs import uniform
m, k, n = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256x784x256").split("x")]
layouts = sys.argv[2] if len(sys.argv) > 2 else "nn"
ta, tb = layouts[0] == "t", layouts[1] == "t"
L.bla_gemm_set_config(6, 1)
a = bla.to_device(uniform(1, (k, m) if ta else (m, k), dtype=np.float32)); b = bla.to_device(uniform(2, (n, k) if tb else (k, n), dtype=np.float32)); c = bla.empty((m, n))
st = bla.zeros((64,), np.uint64)
L.bla_diag_set_stamps.argtypes = [C.c_void_p]; L.bla_diag_set_stamps(st.ptr)
for _ in range(5):
    bla.gemm(a, b, c, transa=ta, transb=tb)
bla.sync()
t = st.numpy().astype(np.int64)
print("kernel:", L.bla_gemm_last_kernel().decode())
t0 = t[0]
names = {0: "start"}
nch = 0
for i in range(1, 20):
    if t[i]:
        names[i] = ("chunk %d operands arrived" if i % 2 else "chunk %d MFMAs issued+done") % ((i - 1) // 2); nch = max(nch, (i + 1) // 2)
names[20] = "K loop done"; names[21] = "after barrier"
prev = t0
for i in sorted(names):
    if t[i]:
        print(f"  {names[i]:<34} +{t[i]-prev:7d} cycles   (t = {t[i]-t0:7d})"); prev = t[i]

#!/usr/bin/env python3
"""How well conditioned is the reference's U-Net at its own constants (model/cifar_unet.c:26-37)?  CPU only.

The oracle composition (oracle.unet: every block is the restatement pinned to the reference's functions) is run twice on the same
fp32-representable parameters and inputs: in fp64 (the reference's matrix_float_t) and in fp32 (the same loops, the same order of additions,
float arithmetic).  The normwise distance between the two, per gradient tensor, is what ANY fp32 evaluation of this network can be expected to
sit from the fp64 result -- group norm divides by the variance with epsilon 0 (lib/norm.c:3,36-44), so rounding of a small variance
is amplified through 36 norm layers.  tests/test_unet_model.py::test_reference_constants_against_the_oracle holds the device to a small
multiple of these figures.  Writes tests/golden/unet_refconst.npz: the fp64 prediction (the fixture bench.py checks its batch-64 pass against)
and the per-tensor fp32-vs-fp64 distances."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle
from inputs import uniform
from unet_refconst import CFG, tensor_list, make_params, make_inputs

oracle.build()
names = tensor_list(CFG)
P32 = make_params(CFG)
x, temb, noise = make_inputs(0)
t0 = time.perf_counter()
out64, G64 = oracle.unet(CFG, {k: v.astype(np.float64) for k, v in P32.items()}, x.astype(np.float64), temb.astype(np.float64), noise.astype(np.float64), None)
t1 = time.perf_counter()
out32, G32 = oracle.unet(CFG, P32, x, temb, noise, None)
t2 = time.perf_counter()
print(f"fp64 oracle {t1 - t0:.1f} s, fp32 oracle {t2 - t1:.1f} s")
e_out = np.linalg.norm(out32 - out64) / np.linalg.norm(out64)
print(f"prediction: fp32 vs fp64 {e_out:.3e}")
dist = {}
for n, shp in names:
    w = G64[n].ravel(); s = np.linalg.norm(w)
    dist[n] = 0.0 if s == 0 else float(np.linalg.norm(G32[n].ravel() - w) / s)
worst = sorted(dist.items(), key=lambda t: -t[1])[:8]
for n, e in worst:
    print(f"  {n:44s} {e:.3e}")
print(f"median over tensors {np.median([v for v in dist.values() if v > 0]):.3e}")
if "--write" in sys.argv:
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "unet_refconst.npz"), prediction=out64, prediction_fp32_distance=e_out,
                        names=np.array([n for n, _ in names]), grad_fp32_distance=np.array([dist[n] for n, _ in names]),
                        grad_norm=np.array([float(np.linalg.norm(G64[n])) for n, _ in names]))
    print("wrote tests/golden/unet_refconst.npz")

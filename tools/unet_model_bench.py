#!/usr/bin/env python3
"""The reference's U-Net at its own constants (model/cifar_unet.c:26-37: 32 x 32 x 3 image, widths 128 / 256 / 256 / 256, time embedding 512,
key dimension 16, groups of 32): one image forward + backward on the device-resident composition (bla_unet_*), HIP-event time per pass,
directly issued and replayed as one recorded graph.  The reference's own `cifar_unet train 1` (one forward + one backward on one image) took
29.3 s on one CPU core.  `--oracle` also times the fp64 oracle composition of the same network on this host."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from __graft_entry__ import load_pkg
from inputs import uniform
import test_unet_model as T
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
cfg = dict(image_h=32, image_w=32, in_channels=3, dims=[128, 256, 256, 256], time_dim=512, kernel=3, group_size=32, key_dim=16)
h, tensors = T.build(bla, cfg)
shapes = T.shapes_for(tensors, cfg)
total = L.bla_unet_param_count(h)
flat = np.zeros(total, np.float32); P = {}
for i, (name, off, cnt) in enumerate(tensors):
    shp = shapes[name]; fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
    scale = 0.05 if name.endswith("biases") else float(np.sqrt(3.0 / fan_in))
    v = uniform(7000 + i, shp, -scale, scale, np.float32); flat[off:off + cnt] = v.ravel(); P[name] = v
chk(L.bla_memcpy_h2d(L.bla_unet_params(h), flat.ctypes.data, flat.nbytes, None)); bla.sync()
x = bla.to_device(uniform(1, (3, 32, 32), -1, 1, np.float32)); temb = bla.to_device(uniform(2, (512,), -1, 1, np.float32)); noise = bla.to_device(uniform(3, (3, 32, 32), -1, 1, np.float32))
st = L.bla_default_stream()
e = [C.c_void_p() for _ in range(3)]
for ev in e: chk(L.bla_event_create(C.byref(ev)))
def one():
    chk(L.bla_unet_forward_f32(h, st, x.ptr, temb.ptr, None)); chk(L.bla_unet_backward_f32(h, st, noise.ptr))
for _ in range(3): one()
bla.sync()
tf = tb = 0.0; iters = 20
for _ in range(iters):
    chk(L.bla_event_record(e[0], st)); chk(L.bla_unet_forward_f32(h, st, x.ptr, temb.ptr, None)); chk(L.bla_event_record(e[1], st))
    chk(L.bla_unet_backward_f32(h, st, noise.ptr)); chk(L.bla_event_record(e[2], st))
    ms = C.c_float(); chk(L.bla_event_elapsed_ms(e[0], e[1], C.byref(ms))); tf += ms.value
    chk(L.bla_event_elapsed_ms(e[1], e[2], C.byref(ms))); tb += ms.value
print(f"U-Net {total} parameters ({total * 4 / 1e6:.1f} MB), one image: forward {tf / iters:.3f} ms, backward {tb / iters:.3f} ms (direct launches; reference CPU: 29.3 s for both)", flush=True)
g = C.c_void_p()
chk(L.bla_graph_begin(st)); one(); chk(L.bla_graph_end(st, C.byref(g)))
for _ in range(3): chk(L.bla_graph_launch(g, st))
chk(L.bla_event_record(e[0], st))
for _ in range(iters): chk(L.bla_graph_launch(g, st))
chk(L.bla_event_record(e[1], st))
ms = C.c_float(); chk(L.bla_event_elapsed_ms(e[0], e[1], C.byref(ms)))
print(f"replayed as one recorded graph: {ms.value / iters:.3f} ms per forward + backward", flush=True)
if "--oracle" in sys.argv:
    import oracle
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    t0 = time.perf_counter()
    out, G = oracle.unet(cfg, P64, x.numpy().astype(np.float64), temb.numpy().astype(np.float64), noise.numpy().astype(np.float64), None)
    dt = time.perf_counter() - t0
    got = np.empty((3, 32, 32), np.float32); chk(L.bla_memcpy_d2h(got.ctypes.data, L.bla_unet_output(h), got.nbytes, None)); bla.sync()
    print(f"fp64 oracle composition on this host: {dt:.1f} s (1 core); prediction normwise difference {np.linalg.norm(got - out) / np.linalg.norm(out):.2e}", flush=True)

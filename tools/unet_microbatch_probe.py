#!/usr/bin/env python3
"""One batch-64 model against two batch-32 models run side by side on two contexts (two streams, two workspaces, two side lanes): would splitting a pass into
two independent half-batch chains beat the one-chain-plus-side-lane form?  Wall time per 64 images, forward + backward."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_pkg
from inputs import uniform
import test_unet_model as T
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
cfg = dict(image_h=32, image_w=32, in_channels=3, dims=[128, 256, 256, 256], time_dim=512, kernel=3, group_size=32, key_dim=16)
ctx2 = C.c_void_p(); chk(L.bla_context_create(C.byref(ctx2), 0))
def setup(B):
    h, tensors = T.build(bla, cfg, B); T.load_params(bla, h, tensors, cfg)
    x = bla.to_device(uniform(1, (B, 3, 32, 32), -1, 1, np.float32)); temb = bla.to_device(uniform(2, (B, 512), -1, 1, np.float32)); noise = bla.to_device(uniform(3, (B, 3, 32, 32), -1, 1, np.float32))
    return h, x, temb, noise
def step(m):
    h, x, temb, noise = m
    chk(L.bla_unet_forward_f32(h, None, x.ptr, temb.ptr, None)); chk(L.bla_unet_backward_f32(h, None, noise.ptr))
def sync_all():
    chk(L.bla_context_set_current(None)); chk(L.bla_stream_sync(None)); chk(L.bla_context_set_current(ctx2)); chk(L.bla_stream_sync(None)); chk(L.bla_context_set_current(None))
N = 10
one = setup(64)
for _ in range(3): step(one)
sync_all(); t0 = time.perf_counter()
for _ in range(N): step(one)
sync_all(); t1 = (time.perf_counter() - t0) / N
chk(L.bla_unet_destroy(one[0]))
a = setup(32)
chk(L.bla_context_set_current(ctx2)); b = setup(32); chk(L.bla_context_set_current(None))
def pair():
    chk(L.bla_context_set_current(None)); step(a)
    chk(L.bla_context_set_current(ctx2)); step(b); chk(L.bla_context_set_current(None))
for _ in range(3): pair()
sync_all(); t0 = time.perf_counter()
for _ in range(N): pair()
sync_all(); t2 = (time.perf_counter() - t0) / N
print(f"one batch-64 model: {t1 * 1e3:7.3f} ms per 64 images ({64 / t1:7.1f} images/s)   two batch-32 models side by side: {t2 * 1e3:7.3f} ms ({64 / t2:7.1f} images/s)")

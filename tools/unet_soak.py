# Soak of the batch-64 U-Net step: 300 forward+backward passes (dropout on, weight gradients on the side lane), the gradient arena compared BITWISE every 15 passes.
# Run on the GPU box: python tools/unet_soak.py  -> "passes 300, snapshots compared 20 mismatches 0 finite True" (round 3, final tree).
import ctypes as C, os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_pkg
from inputs import uniform
import test_unet_model as T
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
cfg = dict(image_h=32, image_w=32, in_channels=3, dims=[128, 256, 256, 256], time_dim=512, kernel=3, group_size=32, key_dim=16)
B = 64
h, tensors = T.build(bla, cfg, B); _, total = T.load_params(bla, h, tensors, cfg)
x = bla.to_device(uniform(1, (B, 3, 32, 32), -1, 1, np.float32)); temb = bla.to_device(uniform(2, (B, 512), -1, 1, np.float32)); noise = bla.to_device(uniform(3, (B, 3, 32, 32), -1, 1, np.float32))
drop = (uniform(4, (L.bla_unet_dropout_count(h),), 0, 1, np.float32) < 0.1).astype(np.uint8)
dd = bla.DeviceArray((drop.size,), np.uint8).copy_from(drop)
ref = None; bad = 0
snap = bla.empty((total,))
for it in range(300):
    chk(L.bla_unet_forward_f32(h, None, x.ptr, temb.ptr, dd.ptr)); chk(L.bla_unet_backward_f32(h, None, noise.ptr))
    if it % 15 == 0:
        chk(L.bla_memcpy_d2d(snap.ptr, L.bla_unet_grads(h), total * 4, None)); bla.sync()
        g = snap.numpy().copy()
        if ref is None: ref = g
        elif not np.array_equal(ref, g): bad += 1
print("passes 300, snapshots compared", 300 // 15, "mismatches", bad, "finite", bool(np.isfinite(ref).all()))

#!/usr/bin/env python3
"""The reference's U-Net at its own constants (model/cifar_unet.c:26-37) for a batch of CIFAR-shaped images per pass (bla_unet_create_batched):
forward + backward time per batch, images/s, and the convolution FLOPs of the pass against the fp32 MFMA peak.
usage: unet_batch_bench.py [batch ...]   (default 1 16 64)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_pkg
from inputs import uniform
import test_unet_model as T
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
cfg = dict(image_h=32, image_w=32, in_channels=3, dims=[128, 256, 256, 256], time_dim=512, kernel=3, group_size=32, key_dim=16)
st = L.bla_default_stream()
e = [C.c_void_p() for _ in range(3)]
for ev in e: chk(L.bla_event_create(C.byref(ev)))


def conv_flops(cfg):
    """2 * MACs of every convolution of one forward pass of one image (3x3 and 1x1), from the layer plan"""
    D = cfg["dims"]; k2 = cfg["kernel"] ** 2; c0 = cfg["in_channels"]
    hw = [cfg["image_h"] * cfg["image_w"] // 4 ** i for i in range(4)]
    res = [(c0, D[0], 0), (D[0], D[0], 0), (D[1], D[1], 1), (D[1], D[1], 1), (D[2], D[2], 2), (D[2], D[2], 2), (D[3], D[3], 3), (D[3], D[3], 3), (D[3], D[3], 3), (D[3], D[3], 3),
           (2 * D[3], D[3], 3), (D[3], D[3], 3), (2 * D[2], D[2], 2), (D[2], D[2], 2), (2 * D[1], D[1], 1), (D[1], D[1], 1), (2 * D[0], D[0], 0), (D[0], D[0], 0)]
    f = 0
    for cin, cout, l in res:
        f += 2 * hw[l] * cout * (cin * k2 + cout * k2 + (cin if cin != cout else 0))
    f += 2 * k2 * (hw[1] * D[0] * D[1] + hw[2] * D[1] * D[2] + hw[3] * D[2] * D[3])                  # stride-2 convolutions
    for i, (a, b, l) in enumerate([(D[3], D[2], 2), (D[2], D[1], 1), (D[1], D[0], 0)]):
        if a != b: f += 2 * k2 * hw[l] * a * b                                                     # channel-changing convolutions after the resize
    return f + 2 * k2 * hw[0] * D[0] * c0


fl = conv_flops(cfg)
for B in [int(a) for a in sys.argv[1:]] or [1, 16, 64]:
    h, tensors = T.build(bla, cfg, B)
    T.load_params(bla, h, tensors, cfg)
    x = bla.to_device(uniform(1, (B, 3, 32, 32), -1, 1, np.float32)); temb = bla.to_device(uniform(2, (B, 512), -1, 1, np.float32))
    noise = bla.to_device(uniform(3, (B, 3, 32, 32), -1, 1, np.float32))
    for _ in range(3):
        chk(L.bla_unet_forward_f32(h, st, x.ptr, temb.ptr, None)); chk(L.bla_unet_backward_f32(h, st, noise.ptr))
    bla.sync()
    tf = tb = 0.0; iters = 10
    for _ in range(iters):
        chk(L.bla_event_record(e[0], st)); chk(L.bla_unet_forward_f32(h, st, x.ptr, temb.ptr, None)); chk(L.bla_event_record(e[1], st))
        chk(L.bla_unet_backward_f32(h, st, noise.ptr)); chk(L.bla_event_record(e[2], st))
        ms = C.c_float(); chk(L.bla_event_elapsed_ms(e[0], e[1], C.byref(ms))); tf += ms.value
        chk(L.bla_event_elapsed_ms(e[1], e[2], C.byref(ms))); tb += ms.value
    tf /= iters; tb /= iters
    print(f"U-Net batch {B:3d}: forward {tf:8.3f} ms, backward {tb:8.3f} ms -> {B / (tf + tb) * 1e3:8.1f} images/s;  convolution FLOPs {fl * B / 1e9:7.1f} G forward: "
          f"{fl * B / tf / 1e9:6.1f} TFLOP/s ({fl * B / tf / 1e9 / 157.3 * 100:4.1f}% of fp32 MFMA peak), backward (2x) {2 * fl * B / tb / 1e9:6.1f} TFLOP/s "
          f"({2 * fl * B / tb / 1e9 / 157.3 * 100:4.1f}%)", flush=True)
    chk(L.bla_unet_destroy(h))

#!/usr/bin/env python3
"""Forward pass of the batch-64 U-Net under `rocprofv3 --kernel-trace`: where its wall time goes -- kernels, or the gaps between them.
  run:      rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/unet_forward_gaps.py run
  summary:  python3 tools/unet_forward_gaps.py summary DIR      (per pass: wall from the first kernel's start to the last one's end, busy time, gap
            histogram, and the kernels by time)"""
import csv, glob, os, sys, time


def run():
    import ctypes as C
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from __graft_entry__ import load_pkg
    from inputs import uniform
    import test_unet_model as T
    bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
    cfg = dict(image_h=32, image_w=32, in_channels=3, dims=[128, 256, 256, 256], time_dim=512, kernel=3, group_size=32, key_dim=16)
    B = 64
    h, tensors = T.build(bla, cfg, B); T.load_params(bla, h, tensors, cfg)
    x = bla.to_device(uniform(1, (B, 3, 32, 32), -1, 1, np.float32)); temb = bla.to_device(uniform(2, (B, 512), -1, 1, np.float32))
    noise = bla.to_device(uniform(3, (B, 3, 32, 32), -1, 1, np.float32))
    for _ in range(2):
        chk(L.bla_unet_forward_f32(h, None, x.ptr, temb.ptr, None)); chk(L.bla_unet_backward_f32(h, None, noise.ptr))
    bla.sync(); time.sleep(0.02)
    for _ in range(6):      # forward passes alone, a synchronisation and 20 ms of sleep between them mark the pass boundaries in the trace
        chk(L.bla_unet_forward_f32(h, None, x.ptr, temb.ptr, None)); bla.sync(); time.sleep(0.02)


def summary(d):
    f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    # passes: split at gaps > 5 ms (the host's sleep)
    passes, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - max(x[1] for x in cur[-8:]) > 5_000_000: passes.append(cur); cur = [r]
        else: cur.append(r)
    passes.append(cur)
    fw = [p for p in passes[-6:]]
    p = fw[-1]
    wall = (max(r[1] for r in p) - p[0][0]) / 1e3
    busy = 0; end = p[0][0]; gaps = []
    for s, e, _ in p:
        if s > end: gaps.append((s - end) / 1e3)
        busy += max(0, e - max(s, end)); end = max(end, e)
    print(f"forward pass: {len(p)} kernels, wall {wall:.1f} us, busy (union of kernels) {busy / 1e3:.1f} us, gaps {sum(gaps):.1f} us in {len(gaps)} ({sum(gaps) / max(1, len(gaps)):.2f} us each)")
    import collections
    hist = collections.Counter(min(int(g), 10) for g in gaps)
    print("gap histogram (us, 10 = 10+):", sorted(hist.items()))
    by = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n in p: by[n.split("(")[0][:100]][0] += 1; by[n.split("(")[0][:100]][1] += (e - s) / 1e3
    for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:22]: print(f"  {t:8.1f} us {c:4d} x {t / c:7.1f}  {n}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else summary(sys.argv[2])

#!/usr/bin/env python3
"""Backward pass of the batch-64 U-Net under `rocprofv3 --kernel-trace`: per hardware queue (main chain / weight-gradient lane) the busy time, the idle time
and the kernels by time; run once as is and once with BLA_UNET_SIDE=0 (everything on the main chain) to see what sharing the chip costs each kernel.
  run:      rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/unet_backward_lanes.py run
  summary:  python3 tools/unet_backward_lanes.py summary DIR"""
import collections, csv, glob, os, sys, time


def run():
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
    for _ in range(5):      # a sleep between forward and backward marks the pass boundaries in the trace
        chk(L.bla_unet_forward_f32(h, None, x.ptr, temb.ptr, None)); bla.sync(); time.sleep(0.02)
        chk(L.bla_unet_backward_f32(h, None, noise.ptr)); bla.sync(); time.sleep(0.02)


def summary(d):
    f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f))]
    rows.sort()
    passes, cur, end = [], [rows[0]], rows[0][1]
    for r in rows[1:]:
        if r[0] - end > 5_000_000: passes.append(cur); cur = [r]; end = r[1]
        else: cur.append(r); end = max(end, r[1])
    passes.append(cur)
    p = passes[-1]      # the last backward pass
    t0 = p[0][0]; wall = (max(r[1] for r in p) - t0) / 1e3
    print(f"backward pass: {len(p)} kernels, wall {wall:.1f} us")
    queues = collections.defaultdict(list)
    for r in p: queues[r[3]].append(r)
    for q, rs in sorted(queues.items(), key=lambda kv: -len(kv[1])):
        busy = 0; end = rs[0][0]; idle = 0
        for s, e, _, _ in rs:
            if s > end: idle += s - end
            busy += max(0, e - max(s, end)); end = max(end, e)
        print(f" queue {q}: {len(rs)} kernels, first start +{(rs[0][0] - t0) / 1e3:.1f} us, last end +{(end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle between its kernels {idle / 1e3:.1f} us")
        by = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n, _ in rs: by[n.split("(")[0][:96]][0] += 1; by[n.split("(")[0][:96]][1] += (e - s) / 1e3
        for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]: print(f"    {t:8.1f} us {c:4d} x {t / c:7.1f}  {n}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else summary(sys.argv[2])

#!/usr/bin/env python3
"""One named workload, run N times with nothing else in the process, for rocprofv3 (kernel trace / stats, or one --pmc pass).
Prints a JSON line with the algorithmic bytes and FLOPs of ONE iteration, which tools/profile_summary.py joins with the counters.

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/profile_targets.py conv128 20
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/profile_targets.py conv128 5

targets: conv128 / conv256 (batch of 64, forward + backward), conv8 (256->256 @8x8 x64), convs2 (128->256 @32x32 stride 2 x64),
mnist (fused-update step), mnist_dp (world-1 data-parallel step), softmax_cols, transpose, add, colsum, rowsum (8192 x 8192; softmax_cols4096 etc.: 4096 x 4096)"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_pkg
from inputs import uniform, randint
bla = load_pkg(); bla.init(0); L = bla.lib(); chk = bla.native.check
st = L.bla_default_stream()
target = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
info = {"target": target, "iters": iters}

if target.startswith("conv"):
    B, k = 64, 3
    h, cin, cout, s = {"conv128": (32, 128, 128, 1), "conv256": (16, 256, 256, 1), "conv8": (8, 256, 256, 1), "convs2": (32, 128, 256, 2)}[target]
    ho = -(-h // s); hw = ho * ho; kkc = k * k * cin
    x = bla.to_device(uniform(31, (B, cin, h, h), -1, 1, np.float32)); kern = bla.to_device(uniform(32, (cout, cin, k, k), -0.1, 0.1, np.float32))
    dy = bla.to_device(uniform(33, (B, cout, ho, ho), -1, 1, np.float32))
    out, dk, dx, scr = bla.empty((B, cout, ho, ho)), bla.empty((cout, cin, k, k)), bla.empty((B, cin, h, h)), bla.empty((cout * kkc,))
    fl = 2.0 * hw * kkc * cout * B
    info.update(flops_forward=fl, flops_backward=2 * fl,
                bytes_forward=4.0 * (B * cin * h * h + cout * kkc + B * cout * hw),                       # input + kernels + output (not the im2col)
                bytes_backward=4.0 * (B * cout * hw + B * cin * h * h + cout * kkc + cout * kkc + B * cin * h * h))
    def run():
        chk(L.bla_conv2d_forward_batched_f32(st, x.ptr, kern.ptr, out.ptr, B, h, h, k, cin, cout, s))
        chk(L.bla_conv2d_backward_batched_f32(st, dy.ptr, x.ptr, kern.ptr, dk.ptr, dx.ptr, scr.ptr, B, h, h, k, cin, cout, s))
elif target in ("mnist", "mnist_dp"):
    mn = bla.mnist_nn
    nn = mn.MnistNN(256, colsum_mode=mn.COLSUM_INTENDED)
    z = np.load(os.path.join(ROOT, "tests", "golden", "mnist_nn_params.npz"))
    nn.set_params([z[n] for n in ["w1", "b1", "w2", "b2", "w3", "b3"]])
    xr = randint(7, (784, 256), 256).astype(np.float32); lab = randint(8, (256,), 10)
    y = np.zeros((10, 256), np.float32); y[lab, np.arange(256)] = 1
    nn.load_batch(xr, y)
    info.update(flops=256 * 1007104.0, bytes=4.0 * (2 * 235146 + 784 * 256))          # weights read + written once, the batch read once
    if target == "mnist":
        run = nn.fused_step
    else:
        ex = mn.Exchange(0, 1, nn.count)
        run = lambda: nn.dp_step(ex, graph=False)
else:
    R = 4096 if target.endswith("4096") else 8192; n = R * R
    target = target.replace("4096", "")
    a = bla.to_device(uniform(1, (R, R), -1, 1, np.float32)); o = bla.empty((R, R)); small = bla.empty((R,))
    if target == "softmax_cols":
        info.update(bytes=8.0 * n); run = lambda: chk(L.bla_softmax_cols_f32(st, a.ptr, R, R))
    elif target == "transpose":
        info.update(bytes=8.0 * n); run = lambda: chk(L.bla_transpose_f32(st, a.ptr, o.ptr, R, R))
    elif target == "add":
        info.update(bytes=12.0 * n); run = lambda: chk(L.bla_add_f32(st, a.ptr, o.ptr, n))
    elif target == "colsum":
        info.update(bytes=4.0 * n); run = lambda: chk(L.bla_col_sum_f32(st, a.ptr, R, R, small.ptr, 1))
    elif target == "rowsum":
        info.update(bytes=4.0 * n); run = lambda: chk(L.bla_row_sum_f32(st, a.ptr, R, R, small.ptr))
    else:
        sys.exit("unknown target " + target)
for _ in range(3):
    run()
bla.sync()
for _ in range(iters):
    run()
bla.sync()
print(json.dumps(info), flush=True)

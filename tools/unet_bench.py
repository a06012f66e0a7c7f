#!/usr/bin/env python3
"""the U-Net's ResNet and self-attention blocks (model/cifar_unet.c:999-1072,1180-1335) at the model's own
shapes, one image: device block (forward / backward) with the CPU restatement of the reference's call sequence timed beside it
(fp64, one core, smaller shapes only -- the 32x32x128 block takes the CPU about a second per pass)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle
from __graft_entry__ import load_pkg
from inputs import uniform
pkg = load_pkg(); pkg.init(0); L = pkg.lib(); chk = pkg.native.check; N = pkg.native
st = L.bla_default_stream()
e0, e1 = C.c_void_p(), C.c_void_p(); chk(L.bla_event_create(C.byref(e0))); chk(L.bla_event_create(C.byref(e1)))
F32 = np.float32


def graphed(fn):
    """fn's launches recorded once (after an eager run has created scratch and tables), replayed with one graph launch"""
    fn(); pkg.native.sync(st)
    g = C.c_void_p()
    chk(L.bla_graph_begin(st)); fn(); chk(L.bla_graph_end(st, C.byref(g)))
    return lambda: chk(L.bla_graph_launch(g, st))


def timeit(fn, iters=20):
    fn(); fn()
    chk(L.bla_event_record(e0, st))
    for _ in range(iters): fn()
    chk(L.bla_event_record(e1, st))
    ms = C.c_float(); chk(L.bla_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value / iters * 1e-3


print("ResNet block (group norm+ReLU, 3x3 conv, time-embedding dense, group norm+ReLU, dropout, 3x3 conv, [1x1 residual conv], add)")
for (cin, cout, hh) in [(3, 128, 32), (128, 128, 32), (256, 128, 32), (256, 256, 16), (512, 256, 16), (256, 256, 8), (256, 256, 4)]:
    tdim, gs = 512, 32
    u = lambda k, shape, lo, hi, dt=F32: uniform(9000 + k, shape, lo, hi, dt)
    I = dict(x=u(0, (cin, hh, hh), -1, 1), temb=u(1, (1, tdim), 0, 1), k1=u(2, (cout, cin, 3, 3), -0.05, 0.05), k2=u(3, (cout, cout, 3, 3), -0.05, 0.05),
             tw=u(4, (tdim, cout), -0.05, 0.05), tb=u(5, (1, cout), -0.1, 0.1), kres=u(6, (cout, cin, 1, 1), -0.1, 0.1) if cin != cout else None,
             del_out=u(7, (cout, hh, hh), -1, 1))
    g1 = (cin + gs - 1) // gs; g2 = (cout + gs - 1) // gs
    D = {n: pkg.to_device(v) for n, v in I.items() if v is not None}
    dropped = (uniform(77, (cout, hh, hh), 0, 1, F32) < 0.1).astype(np.uint8)
    drop = pkg.to_device(dropped, np.uint8)
    W = dict(mu1=pkg.empty((g1,)), sd1=pkg.empty((g1,)), relu1=pkg.empty((cin, hh, hh)), c1=pkg.empty((cout, hh, hh)), tdense=pkg.empty((cout,)),
             mu2=pkg.empty((g2,)), sd2=pkg.empty((g2,)), relu2=pkg.empty((cout, hh, hh)), dp=pkg.empty((cout, hh, hh)), c2=pkg.empty((cout, hh, hh)),
             res=pkg.empty((cout, hh, hh)))
    params = N.ResnetParams(D["k1"].ptr, D["k2"].ptr, D["tw"].ptr, D["tb"].ptr, D["kres"].ptr if cin != cout else None)
    ws = N.ResnetWs(*[W[n].ptr for n in ("mu1", "sd1", "relu1", "c1", "tdense", "mu2", "sd2", "relu2", "dp", "c2", "res")])
    result = pkg.empty((cout, hh, hh))
    fwd = lambda: chk(L.bla_resnet_forward_f32(st, D["x"].ptr, D["temb"].ptr, C.byref(params), drop.ptr, C.byref(ws), result.ptr, hh, hh, cin, cout, 3, tdim, gs))
    G = dict(k1=pkg.empty((cout, cin, 3, 3)), k2=pkg.empty((cout, cout, 3, 3)), tw=pkg.empty((tdim, cout)), tb=pkg.empty((cout,)), kres=pkg.empty((cout, cin, 1, 1)))
    grads = N.ResnetGrads(G["k1"].ptr, G["k2"].ptr, G["tw"].ptr, G["tb"].ptr, G["kres"].ptr if cin != cout else None)
    S = dict(a=pkg.empty((cout, hh, hh)), b=pkg.empty((cout, hh, hh)), c=pkg.empty((cin, hh, hh)), f=pkg.empty((cout * max(cin, cout) * 9,)))
    scratch = N.ResnetScratch(S["a"].ptr, S["b"].ptr, S["c"].ptr, S["f"].ptr)
    del_x = pkg.empty((cin, hh, hh))
    bwd = lambda: chk(L.bla_resnet_backward_f32(st, D["del_out"].ptr, D["x"].ptr, D["temb"].ptr, C.byref(params), C.byref(ws), C.byref(grads), C.byref(scratch),
                                                del_x.ptr, hh, hh, cin, cout, 3, tdim, gs))
    tf, tb = timeit(fwd), timeit(bwd)
    tfg, tbg = timeit(graphed(fwd)), timeit(graphed(bwd))
    fl = 2.0 * hh * hh * 9 * (cin * cout + cout * cout) + (2.0 * hh * hh * cin * cout if cin != cout else 0)   # conv products, forward
    line = (f"{cin:>3}->{cout:<3} {hh:>2}x{hh:<2}  fwd {tf*1e6:7.1f} us, as one graph {tfg*1e6:7.1f} us ({fl/tfg/1e12:5.2f} TF/s)  "
            f"bwd {tb*1e6:7.1f} us, as one graph {tbg*1e6:7.1f} us ({2*fl/tbg/1e12:5.2f} TF/s)")
    if cin * cout * hh * hh <= 256 * 256 * 64:
        J = {k: (v.astype(np.float64) if v is not None else None) for k, v in I.items()}
        t0 = time.perf_counter(); f = oracle.resnet_forward(J["x"], J["temb"], J["k1"], J["k2"], J["tw"], J["tb"], J["kres"], dropped.reshape(-1).astype(np.uint8), gs); tcf = time.perf_counter() - t0
        t0 = time.perf_counter(); oracle.resnet_backward(J["del_out"], J["x"], J["temb"], J["k1"], J["k2"], J["kres"], f, gs); tcb = time.perf_counter() - t0
        err = np.abs(result.numpy() - f["result"]).max() / (np.abs(f["result"]).mean() + 1e-30)
        line += f"   | CPU restatement fwd {tcf*1e3:7.1f} ms  bwd {tcb*1e3:7.1f} ms   (max |diff| / mean |ref| of the block output {err:.1e})"
    print(line, flush=True)

print("\nSelf-attention block (Q,K,V dense, softmax(QK^T) rows, AV, output dense + bias + residual), d = 16")
for (c, hh) in [(256, 16), (256, 4)]:
    d = 16; s = hh * hh
    u = lambda k, shape, lo, hi, dt=F32: uniform(9500 + k, shape, lo, hi, dt)
    A = dict(x=u(0, (c, hh, hh), -1, 1), wq=u(1, (c, d), -0.1, 0.1), wk=u(2, (c, d), -0.1, 0.1), wv=u(3, (c, d), -0.1, 0.1), w=u(4, (d, c), -0.1, 0.1),
             b=u(5, (1, c), -0.1, 0.1), dy=u(6, (c, hh, hh), -1, 1))
    dev = {n: pkg.to_device(a) for n, a in A.items()}

    def mk():
        bufs = dict(q=pkg.empty((s, d)), k=pkg.empty((s, d)), v=pkg.empty((s, d)), scores_raw=pkg.empty((s, s)), weights=pkg.empty((s, s)), attention=pkg.empty((s, d)))
        return bufs, N.AttentionWs(*[bufs[n].ptr for n in ("q", "k", "v", "scores_raw", "weights", "attention")])
    fb, fws = mk(); gb, gws = mk()
    out = pkg.empty((c, hh, hh))
    outs = dict(del_wq=pkg.empty((c, d)), del_wk=pkg.empty((c, d)), del_wv=pkg.empty((c, d)), del_w=pkg.empty((d, c)), del_x=pkg.empty((c, hh, hh)))
    fwd = lambda: chk(L.bla_attention_forward_f32(st, dev["x"].ptr, dev["wq"].ptr, dev["wk"].ptr, dev["wv"].ptr, dev["w"].ptr, dev["b"].ptr, C.byref(fws), out.ptr, c, s, d))
    bwd = lambda: chk(L.bla_attention_backward_f32(st, dev["dy"].ptr, dev["x"].ptr, dev["wq"].ptr, dev["wk"].ptr, dev["wv"].ptr, dev["w"].ptr, C.byref(fws), C.byref(gws),
                                                   outs["del_wq"].ptr, outs["del_wk"].ptr, outs["del_wv"].ptr, outs["del_w"].ptr, outs["del_x"].ptr, c, s, d, 0))
    tf, tb = timeit(fwd), timeit(bwd)
    tfg, tbg = timeit(graphed(fwd)), timeit(graphed(bwd))
    print(f"C={c} S={s:<3}  fwd {tf*1e6:7.1f} us, as one graph {tfg*1e6:7.1f} us   bwd {tb*1e6:7.1f} us, as one graph {tbg*1e6:7.1f} us", flush=True)

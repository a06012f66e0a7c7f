"""ResNet block of the U-Net (model/cifar_unet.c:1044-1072 forward, :1180-1227 backward).
CPU: oracle == the reference's own call sequence driven in the intended composition (tests/golden/resnet.npz, bit-exact).
GPU: device block (group norm + ReLU, implicit-GEMM convs, time-embedding dense, dropout with the host's rand() draws,
residual 1x1 conv) within fp32 tolerances."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden
from inputs import uniform

F32 = np.float32
FWD = ["mu1", "sd1", "relu1", "c1", "tdense", "mu2", "sd2", "relu2", "dp", "c2", "result"]


def resnet_inputs(i, cin, cout, hh, tdim, dtype=np.float64):
    sd = 6000 + 30 * i
    u = lambda k, shape, lo, hi: uniform(sd + k, shape, lo, hi, dtype)
    return dict(x=u(0, (cin, hh, hh), -1, 1), temb=u(1, (1, tdim), 0, 1), k1=u(2, (cout, cin, 3, 3), -0.2, 0.2), k2=u(3, (cout, cout, 3, 3), -0.1, 0.1),
                tw=u(4, (tdim, cout), -0.1, 0.1), tb=u(5, (1, cout), -0.1, 0.1), kres=u(6, (cout, cin, 1, 1), -0.3, 0.3) if cin != cout else None,
                del_out=u(7, (cout, hh, hh), -1, 1))


def test_oracle_matches_reference_sequence(ora):
    g = golden("resnet")
    for i, (cin, cout, hh, tdim, gs) in enumerate(g["cfgs"]):
        cin, cout, hh, tdim, gs = int(cin), int(cout), int(hh), int(tdim), int(gs)
        I = resnet_inputs(i, cin, cout, hh, tdim)
        fwd = ora.resnet_forward(I["x"], I["temb"], I["k1"], I["k2"], I["tw"], I["tb"], I["kres"], g[f"r{i}_dropped"], gs)
        for n in FWD + (["res"] if cin != cout else []):
            g.check(f"r{i}_{n}", fwd[n], exact=True)
        bwd = ora.resnet_backward(I["del_out"], I["x"], I["temb"], I["k1"], I["k2"], I["kres"], fwd, gs)
        for n in ["dk1", "dk2", "dtw", "dtb", "del_x"] + (["dkres"] if cin != cout else []):
            g.check(f"r{i}_{n}", bwd[n], exact=True)


@pytest.mark.gpu
def test_device_resnet_block(pkg, ora):
    pkg.init(0)
    L, chk, N = pkg.lib(), pkg.native.check, pkg.native
    g = golden("resnet")
    for i, (cin, cout, hh, tdim, gs) in enumerate(g["cfgs"]):
        cin, cout, hh, tdim, gs = int(cin), int(cout), int(hh), int(tdim), int(gs)
        I = resnet_inputs(i, cin, cout, hh, tdim, F32)
        hw = hh * hh; g1 = (cin + gs - 1) // gs; g2 = (cout + gs - 1) // gs
        D = {n: pkg.to_device(v) for n, v in I.items() if v is not None}
        drop = pkg.to_device(g[f"r{i}_dropped"].reshape(cout, hh, hh), np.uint8)
        W = dict(mu1=pkg.empty((g1,)), sd1=pkg.empty((g1,)), relu1=pkg.empty((cin, hh, hh)), c1=pkg.empty((cout, hh, hh)), tdense=pkg.empty((cout,)),
                 mu2=pkg.empty((g2,)), sd2=pkg.empty((g2,)), relu2=pkg.empty((cout, hh, hh)), dp=pkg.empty((cout, hh, hh)), c2=pkg.empty((cout, hh, hh)),
                 res=pkg.empty((cout, hh, hh)))
        ptr = lambda a: a.ptr if a is not None else None
        params = N.ResnetParams(D["k1"].ptr, D["k2"].ptr, D["tw"].ptr, D["tb"].ptr, ptr(D.get("kres")))
        ws = N.ResnetWs(*[W[n].ptr for n in ("mu1", "sd1", "relu1", "c1", "tdense", "mu2", "sd2", "relu2", "dp", "c2", "res")])
        result = pkg.empty((cout, hh, hh)).fill_bytes(0xFF)
        chk(L.bla_resnet_forward_f32(None, D["x"].ptr, D["temb"].ptr, C.byref(params), drop.ptr, C.byref(ws), result.ptr, hh, hh, cin, cout, 3, tdim, gs))
        got = dict(W); got["result"] = result
        for n in FWD + (["res"] if cin != cout else []):
            # group norm divides by the variance (Q3) and two of them are chained: values reach 1e2..1e3, tolerance relative to scale
            g.check(f"r{i}_{n}", got[n].numpy(), rtol=1e-4, atol=1e-4 * g.mean_abs(f"r{i}_{n}") + 1e-7)
        G = dict(k1=pkg.empty((cout, cin, 3, 3)), k2=pkg.empty((cout, cout, 3, 3)), tw=pkg.empty((tdim, cout)), tb=pkg.empty((cout,)), kres=pkg.empty((cout, cin, 1, 1)))
        grads = N.ResnetGrads(G["k1"].ptr, G["k2"].ptr, G["tw"].ptr, G["tb"].ptr, G["kres"].ptr if cin != cout else None)
        S = dict(a=pkg.empty((cout, hh, hh)), b=pkg.empty((cout, hh, hh)), c=pkg.empty((cin, hh, hh)), f=pkg.empty((cout * max(cin, cout) * 9,)))
        scratch = N.ResnetScratch(S["a"].ptr, S["b"].ptr, S["c"].ptr, S["f"].ptr)
        del_x = pkg.empty((cin, hh, hh)).fill_bytes(0xFF)
        chk(L.bla_resnet_backward_f32(None, D["del_out"].ptr, D["x"].ptr, D["temb"].ptr, C.byref(params), C.byref(ws), C.byref(grads), C.byref(scratch),
                                      del_x.ptr, hh, hh, cin, cout, 3, tdim, gs))
        outs = dict(dk1=G["k1"], dk2=G["k2"], dtw=G["tw"], dtb=G["tb"], del_x=del_x)
        if cin != cout:
            outs["dkres"] = G["kres"]
        for n, a in outs.items():
            g.check(f"r{i}_{n}", a.numpy(), rtol=2e-4, atol=2e-4 * g.mean_abs(f"r{i}_{n}") + 1e-7)

"""Self-attention block of the U-Net (model/cifar_unet.c:999-1022, 1261-1337).
CPU: oracle == the reference's own call sequence (tests/golden/attention.npz, bit-exact).  GPU: device composition
(14 + 8 transposition-free GEMMs, row softmax, softmax Jacobian) within the GEMM tolerance chain."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden
from inputs import uniform

F32 = np.float32
FWD = ["q", "k", "v", "raw", "wts", "att", "out"]
BWD = ["del_wq", "del_wk", "del_wv", "del_w", "del_x"]


def attention_inputs(i, c, hh, d, dtype=np.float64):
    sd = 5000 + 20 * i
    return (uniform(sd, (c, hh, hh), -1, 1, dtype), uniform(sd + 1, (c, d), -0.2, 0.2, dtype), uniform(sd + 2, (c, d), -0.2, 0.2, dtype),
            uniform(sd + 3, (c, d), -0.2, 0.2, dtype), uniform(sd + 4, (d, c), -0.2, 0.2, dtype), uniform(sd + 5, (1, c), -0.1, 0.1, dtype),
            uniform(sd + 6, (c, hh, hh), -1, 1, dtype))


def test_oracle_matches_reference_sequence(ora):
    g = golden("attention")
    for i, (c, hh, d) in enumerate(g["cfgs"]):
        x, wq, wk, wv, w, b, dy = attention_inputs(i, int(c), int(hh), int(d))
        fwd = ora.attention_forward(x, wq, wk, wv, w, b)
        for n in FWD:
            g.check(f"a{i}_{n}", fwd[n], exact=True)
        for tag, jr in (("intended", False), ("rawjac", True)):
            bwd = ora.attention_backward(dy, x, wq, wk, wv, w, fwd, jacobian_from_raw=jr)
            for n in BWD:
                g.check(f"a{i}_{tag}_{n}", bwd[n], exact=True)


@pytest.mark.gpu
def test_device_attention_block(pkg, ora):
    pkg.init(0)
    L, chk = pkg.lib(), pkg.native.check
    g = golden("attention")
    for i, (c, hh, d) in enumerate(g["cfgs"]):
        c, hh, d = int(c), int(hh), int(d); s = hh * hh
        x, wq, wk, wv, w, b, dy = attention_inputs(i, c, hh, d, F32)
        dev = {n: pkg.to_device(a) for n, a in dict(x=x, wq=wq, wk=wk, wv=wv, w=w, b=b, dy=dy).items()}

        def ws():
            bufs = dict(q=pkg.empty((s, d)), k=pkg.empty((s, d)), v=pkg.empty((s, d)), scores_raw=pkg.empty((s, s)), weights=pkg.empty((s, s)),
                        attention=pkg.empty((s, d)))
            return bufs, pkg.native.AttentionWs(*[bufs[n].ptr for n in ("q", "k", "v", "scores_raw", "weights", "attention")])
        fb, fws = ws()
        out = pkg.empty((c, hh, hh)).fill_bytes(0xFF)
        chk(L.bla_attention_forward_f32(None, dev["x"].ptr, dev["wq"].ptr, dev["wk"].ptr, dev["wv"].ptr, dev["w"].ptr, dev["b"].ptr, C.byref(fws),
                                        out.ptr, c, s, d))
        got = dict(q=fb["q"], k=fb["k"], v=fb["v"], raw=fb["scores_raw"], wts=fb["weights"], att=fb["attention"], out=out)
        for n in FWD:
            g.check(f"a{i}_{n}", got[n].numpy(), rtol=2e-5, atol=2e-5 * g.mean_abs(f"a{i}_{n}"))
        for tag, jr in (("intended", 0), ("rawjac", 1)):
            gb, gws = ws()
            outs = dict(del_wq=pkg.empty((c, d)), del_wk=pkg.empty((c, d)), del_wv=pkg.empty((c, d)), del_w=pkg.empty((d, c)), del_x=pkg.empty((c, hh, hh)))
            chk(L.bla_attention_backward_f32(None, dev["dy"].ptr, dev["x"].ptr, dev["wq"].ptr, dev["wk"].ptr, dev["wv"].ptr, dev["w"].ptr, C.byref(fws),
                                             C.byref(gws), outs["del_wq"].ptr, outs["del_wk"].ptr, outs["del_wv"].ptr, outs["del_w"].ptr,
                                             outs["del_x"].ptr, c, s, d, jr))
            for n in BWD:
                # gradients chain 3-5 products: 5e-5 of the tensor's scale
                g.check(f"a{i}_{tag}_{n}", outs[n].numpy(), rtol=5e-5, atol=5e-5 * g.mean_abs(f"a{i}_{tag}_{n}"))

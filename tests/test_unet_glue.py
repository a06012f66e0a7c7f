"""U-Net glue ops (model/cifar_unet.c:235-253,1024-1097,1168-1259).
CPU: the oracle reproduces, bit for bit, what the reference's own functions produced (tests/golden/unet_glue.npz,
generated from oracle/_ref/libref_unet.so).  GPU: the HIP kernels against the same vectors -- index/select ops and the
fixed-order resize gradient bit-exactly, softmax_ddx to fp32 rounding."""
import numpy as np
import pytest

from conftest import golden
from inputs import uniform

F32 = np.float32
C_, H_, W_ = 6, 8, 8


def inputs():
    x = uniform(4000, (C_, H_, W_), -1, 1); src = uniform(4001, (C_, H_, W_), -1, 1)
    return x, src, np.maximum(x, 0)


def test_oracle_matches_reference_functions(ora):
    g = golden("unet_glue")
    x, src, rr = inputs()
    assert np.array_equal(ora.relu_mask(src, rr), g["relu_mask"])
    assert np.array_equal(ora.add_channel_bias(x, uniform(4002, (C_, 1))), g["add_time_embedding"])
    dropped = g["dropout_draws"] < F32(0.1)                     # DROPOUT_RATE = 0.1 (model/cifar_unet.c:37), float compare as there
    assert np.array_equal(ora.dropout_apply(x, dropped.astype(np.uint8)), g["dropout_y"])
    assert 0 < dropped.sum() < x.size // 4
    assert np.array_equal(ora.dropout_mask(uniform(4003, (C_, H_, W_), -1, 1), g["dropout_y"]), g["dropout_mask"])
    for tag, (ih, iw, oh, ow, sc) in {"nn2": (4, 4, 8, 8, 2), "nn3": (3, 2, 7, 5, 3)}.items():
        assert np.array_equal(ora.nearest_neighbours(uniform(4010 + sc, (C_, ih, iw), -1, 1), oh, ow, sc), g[tag + "_up"])
        assert np.array_equal(ora.nearest_neighbours_ddx(uniform(4020 + sc, (C_, oh, ow), -1, 1), ih, iw, sc), g[tag + "_ddx"])
    assert np.array_equal(ora.softmax_ddx(g["softmax_ddx_s"], uniform(4031, (16, 16), -1, 1)), g["softmax_ddx"])
    a, b = uniform(4040, (3, 4, 4)), uniform(4041, (3, 4, 4))
    assert np.array_equal(np.concatenate([a, b]), g["concat"]) and np.array_equal(b, g["split_second"])   # plain channel-range copies


@pytest.mark.gpu
def test_device_kernels_match_reference(pkg, ora):
    pkg.init(0)
    L, chk, dev = pkg.lib(), pkg.native.check, pkg
    g = golden("unet_glue")
    x, src, rr = [a.astype(F32) for a in inputs()]
    n = x.size
    dx, ds, dr = dev.to_device(x), dev.to_device(src), dev.to_device(rr)
    out = dev.empty(x.shape)
    chk(L.bla_relu_mask_f32(None, out.ptr, ds.ptr, dr.ptr, n)); assert np.array_equal(out.numpy(), g["relu_mask"].astype(F32))
    chk(L.bla_relu_mask_f32(None, ds.ptr, ds.ptr, dr.ptr, n)); assert np.array_equal(ds.numpy(), g["relu_mask"].astype(F32))   # in place (:1203)
    t = dev.to_device(uniform(4002, (C_, 1), dtype=F32)); xa = dev.to_device(x)
    chk(L.bla_add_tile_columns_f32(None, xa.ptr, C_, H_ * W_, t.ptr, 1))
    assert np.allclose(xa.numpy(), g["add_time_embedding"], rtol=0, atol=1e-6)
    drop = dev.to_device(g["dropout_dropped"].reshape(x.shape), np.uint8); y = dev.empty(x.shape)
    chk(L.bla_dropout_f32(None, dx.ptr, y.ptr, drop.ptr, n)); assert np.array_equal(y.numpy(), g["dropout_y"].astype(F32))
    gm = dev.to_device(uniform(4003, (C_, H_, W_), -1, 1, F32))
    chk(L.bla_dropout_mask_f32(None, gm.ptr, y.ptr, n)); assert np.array_equal(gm.numpy(), g["dropout_mask"].astype(F32))
    for tag, (ih, iw, oh, ow, sc) in {"nn2": (4, 4, 8, 8, 2), "nn3": (3, 2, 7, 5, 3)}.items():
        xi = uniform(4010 + sc, (C_, ih, iw), -1, 1, F32); up = dev.empty((C_, oh, ow))
        chk(L.bla_nearest_neighbours_f32(None, dev.to_device(xi).ptr if False else (keep := dev.to_device(xi)).ptr, up.ptr, C_, ih, iw, oh, ow, sc))
        assert np.array_equal(up.numpy(), g[tag + "_up"].astype(F32))
        gs = uniform(4020 + sc, (C_, oh, ow), -1, 1, F32); dst = dev.empty((C_, ih, iw)); keep2 = dev.to_device(gs)
        chk(L.bla_nearest_neighbours_ddx_f32(None, keep2.ptr, dst.ptr, C_, oh, ow, ih, iw, sc))
        assert np.array_equal(dst.numpy(), ora.nearest_neighbours_ddx(gs, ih, iw, sc))      # same fp32 adds in the same order
        assert np.allclose(dst.numpy(), g[tag + "_ddx"], rtol=2e-6, atol=2e-6)
        assert L.bla_nearest_neighbours_f32(None, keep.ptr, up.ptr, C_, ih, iw, oh * 2, ow, sc) == 1   # would read past the input
    s_ = dev.to_device(g["softmax_ddx_s"].astype(F32)); gr = dev.to_device(uniform(4031, (16, 16), -1, 1, F32)); o = dev.empty((16, 16))
    chk(L.bla_softmax_ddx_f32(None, s_.ptr, gr.ptr, o.ptr, 16, 16))
    assert np.allclose(o.numpy(), g["softmax_ddx"], rtol=1e-5, atol=1e-7)

"""GPU parity of the lib/conv.c stages and lib/norm.c group norm against the reference's golden
vectors (tests/golden/conv.npz, norm.npz).  Index-only stages (im2col, kernel/channel reshapes) are
bit-exact; col2im is bit-identical to the fp32 oracle (same addition order); GEMM-bearing stages use
the GEMM tolerance 1e-5 * (|A||B|)."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden
from inputs import uniform

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def dev(pkg):
    pkg.init(0)
    return pkg


def call(dev, name, *args):
    """DeviceArray arguments are passed by pointer and kept alive until the call has been issued
    (a temporary's bla_free would otherwise hand its memory to the next allocation)."""
    raw = [a.ptr if isinstance(a, dev.DeviceArray) else a for a in args]
    dev.native.check(getattr(dev.lib(), name)(None, *raw))


def conv_case(i, cfg):
    h, w, cin, cout, k, s = [int(v) for v in cfg]
    seed = 1000 + 10 * i
    return (h, w, cin, cout, k, s, seed, uniform(seed, (cin, h, w), -1, 1, F32), uniform(seed + 1, (cout, cin, k, k), -0.3, 0.3, F32))


def test_conv_forward_stages(dev, ora):
    g = golden("conv")
    for i, cfg in enumerate(g["cfgs"]):
        h, w, cin, cout, k, s, seed, x, kern = conv_case(i, cfg)
        ho, wo = C.c_int(), C.c_int()
        dev.native.check(dev.lib().bla_conv_out_hw(h, w, s, C.byref(ho), C.byref(wo)))
        assert (ho.value, wo.value) == ora.out_hw(h, w, s)
        hw, kkc = ho.value * wo.value, k * k * cin
        dx, dk = dev.to_device(x), dev.to_device(kern)
        im, km, pr, out = dev.empty((hw, kkc)), dev.empty((kkc, cout)), dev.empty((hw, cout)), dev.empty((cout, ho.value, wo.value))
        call(dev, "bla_conv_forward_f32", dx, dk, im, km, pr, out, h, w, k, cin, cout, s)
        g.check(f"c{i}_im2col", im.numpy(), exact=True)          # fp32 inputs are exact in fp64: index-only stage is bit-exact
        g.check(f"c{i}_kmat", km.numpy(), exact=True)
        bound = np.abs(im.numpy().astype(np.float64)) @ np.abs(km.numpy().astype(np.float64))
        ref = ora.conv_intended(x.astype(np.float64), kern.astype(np.float64), s)
        assert (np.abs(pr.numpy() - ref["product"]) <= 1e-5 * bound + 1e-30).all()
        g.check(f"c{i}_product", pr.numpy(), rtol=0, atol=1e-5 * bound.max())
        assert np.array_equal(out.numpy().reshape(cout, hw), pr.numpy().T)      # output is a pure re-indexing of product
        g.check(f"c{i}_output", out.numpy(), rtol=0, atol=1e-5 * bound.max())
        # stand-alone stages
        im2 = dev.empty((hw, kkc)); call(dev, "bla_im2col_f32", dx, im2, h, w, k, cin, s)
        assert np.array_equal(im2.numpy(), im.numpy())
        kb = dev.empty((cout, cin, k, k)); call(dev, "bla_matrix_to_kernels_f32", km, kb, cout, cin, k)
        assert np.array_equal(kb.numpy(), kern)
        back = dev.empty((hw, cout)); call(dev, "bla_reshape_matrix_channels_f32", back, out, cout, hw)
        assert np.array_equal(back.numpy(), pr.numpy())


def test_col2im_and_backward(dev, ora):
    g = golden("conv")
    for i, cfg in enumerate(g["cfgs"]):
        h, w, cin, cout, k, s, seed, x, kern = conv_case(i, cfg)
        hw, kkc = h * w, k * k * cin
        if s != 1:
            # The reference's _col2im / conv_ddx are undefined here.  Default: the intended operation, the adjoint of _im2col
            # (BLA_STRICT_REFERENCE=1 refuses: test_strict_reference_refuses_undefined_strides).  Pinned by the adjoint identity and by the
            # oracle's restatement of that adjoint.
            ho, wo = ora.out_hw(h, w, s); hwo = ho * wo
            cols = uniform(seed + 2, (hwo, kkc), -1, 1, F32)
            o = dev.empty((cin, h, w)).fill_bytes(0xFF); call(dev, "bla_col2im_f32", dev.to_device(cols), o, h, w, k, cin, s)
            want = ora.col2im_adjoint(cols.astype(np.float64), cin, h, w, k, s)
            assert np.abs(o.numpy() - want).max() <= 2e-6 * np.abs(cols).max() * k * k
            xr = uniform(seed + 5, (cin, h, w), -1, 1, F32)
            imx = dev.empty((hwo, kkc)); call(dev, "bla_im2col_f32", dev.to_device(xr), imx, h, w, k, cin, s)
            lhs = float((imx.numpy().astype(np.float64) * cols).sum()); rhs = float((xr.astype(np.float64) * o.numpy()).sum())
            assert abs(lhs - rhs) <= 1e-5 * (np.abs(imx.numpy()).astype(np.float64) * np.abs(cols)).sum()
            # conv_ddx chain with the input's size h x w and del_y of the output's size
            fw = ora.conv_intended(x, kern, s)
            del_y = uniform(seed + 3, (cout, ho, wo), -1, 1, F32)
            dq, dkm, dkern = dev.empty((hwo, cout)), dev.empty((kkc, cout)), dev.empty((cout, cin, k, k))
            dcol, dxx = dev.empty((hwo, kkc)), dev.empty((cin, h, w))
            call(dev, "bla_conv_backward_f32", dev.to_device(del_y), dev.to_device(fw["im2col"]), dev.to_device(fw["kmat"]),
                 dq, dkm, dkern, dcol, dxx, h, w, k, cin, cout, s)
            dq64 = ora.reshape_matrix_channels(del_y.astype(np.float64))
            assert np.array_equal(dq.numpy(), dq64.astype(F32))
            im64, km64 = fw["im2col"].astype(np.float64), fw["kmat"].astype(np.float64)
            assert (np.abs(dkm.numpy() - im64.T @ dq64) <= 1e-5 * (np.abs(im64.T) @ np.abs(dq64))).all()
            dcol64 = dq64 @ km64.T; b2 = (np.abs(dq64) @ np.abs(km64.T)).max()
            assert (np.abs(dcol.numpy() - dcol64) <= 1e-5 * b2).all()
            assert (np.abs(dxx.numpy() - ora.col2im_adjoint(dcol64, cin, h, w, k, s)) <= 1e-5 * b2 * k * k).all()
            continue
        cols = uniform(seed + 2, (hw, kkc), -1, 1, F32)
        o = dev.empty((cin, h, w)); call(dev, "bla_col2im_f32", dev.to_device(cols), o, h, w, k, cin, 1)
        assert np.array_equal(o.numpy(), ora.col2im(cols, cin, h, w, k))       # same fp32 additions in the same order
        g.check(f"c{i}_col2im", o.numpy(), rtol=2e-6, atol=2e-6)
        # intended conv_ddx chain
        fw = ora.conv_intended(x, kern, 1)
        del_y = uniform(seed + 3, (cout, h, w), -1, 1, F32)
        dq, dkm, dkern = dev.empty((hw, cout)), dev.empty((kkc, cout)), dev.empty((cout, cin, k, k))
        dcol, dxx = dev.empty((hw, kkc)), dev.empty((cin, h, w))
        call(dev, "bla_conv_backward_f32", dev.to_device(del_y), dev.to_device(fw["im2col"]), dev.to_device(fw["kmat"]),
             dq, dkm, dkern, dcol, dxx, h, w, k, cin, cout, 1)
        g.check(f"c{i}_ddx_del_q", dq.numpy(), exact=True)
        im64, km64, dq64 = fw["im2col"].astype(np.float64), fw["kmat"].astype(np.float64), dq.numpy().astype(np.float64)
        b1 = np.abs(im64.T) @ np.abs(dq64)
        g.check(f"c{i}_ddx_del_kmat", dkm.numpy(), rtol=0, atol=1e-5 * b1.max())
        assert np.array_equal(dkern.numpy().reshape(cout, kkc), dkm.numpy().T)
        b2 = np.abs(dq64) @ np.abs(km64.T)
        g.check(f"c{i}_ddx_del_col", dcol.numpy(), rtol=0, atol=1e-5 * b2.max())
        g.check(f"c{i}_ddx_del_x", dxx.numpy(), rtol=0, atol=1e-5 * b2.max() * k * k)
        # adjoint identity <im2col(x), y> == <x, col2im(y)> (defines col2im independently of the reference)
        xr = uniform(seed + 5, (cin, h, w), -1, 1, F32)
        imx = dev.empty((hw, kkc)); call(dev, "bla_im2col_f32", dev.to_device(xr), imx, h, w, k, cin, 1)
        lhs = float((imx.numpy().astype(np.float64) * cols).sum()); rhs = float((xr.astype(np.float64) * o.numpy()).sum())
        assert abs(lhs - rhs) <= 1e-5 * (np.abs(imx.numpy()).astype(np.float64) * np.abs(cols)).sum()


def test_group_norm(dev, ora):
    g = golden("norm")
    for i, (c, gs, h, w) in enumerate(g["cfgs"]):
        c, gs, h, w = int(c), int(gs), int(h), int(w)
        x = uniform(2000 + i, (c, h, w), -1, 3, F32)
        ng = (c + gs - 1) // gs
        out, sd, mu = dev.empty((c, h, w)), dev.empty((ng,)), dev.empty((ng,))
        call(dev, "bla_group_norm_f32", dev.to_device(x), out, sd, mu, c, gs, h * w)
        assert np.allclose(mu.numpy(), g[f"n{i}_means"], rtol=2e-6, atol=1e-7)
        assert np.allclose(sd.numpy(), g[f"n{i}_stdevs"], rtol=5e-6)
        g.check(f"n{i}_out", out.numpy(), rtol=1e-5, atol=1e-6)
        up = uniform(2100 + i, (c, h, w), -1, 1, F32)
        dest = dev.empty((c, h, w))
        call(dev, "bla_group_norm_ddx_f32", dev.to_device(up), dest, dev.to_device(x), mu, sd, c, gs, h * w)
        ref = g[f"n{i}_ddx"]
        assert (np.abs(dest.numpy() - ref) <= 2e-5 * np.abs(ref) + 2e-5 * np.abs(ref).max()).all()
    # the documented probe: group {1..8} -> mean 4.5, "stdev" (variance) 5.25, out[0] = -0.6667
    out, sd, mu = dev.empty((2, 2, 2)), dev.empty((1,)), dev.empty((1,))
    call(dev, "bla_group_norm_f32", dev.to_device(np.arange(1, 9, dtype=F32).reshape(2, 2, 2)), out, sd, mu, 2, 32, 4)
    assert mu.numpy()[0] == 4.5 and sd.numpy()[0] == 5.25 and abs(out.numpy().ravel()[0] + 0.6667) < 1e-4
    assert np.allclose(out.numpy(), g["probe_out"], rtol=1e-6)


def test_group_norm_large_groups(dev, ora):
    """Groups beyond the one-pass paths (32 channels x 32x32 = the U-Net's first level; a ragged last group; a group larger than
    the forward kernel's register copy): the gradient is cut into slices over several workgroups there.  Against the fp64 oracle."""
    for i, (c, gs, h, w) in enumerate([(64, 32, 32, 32), (40, 32, 32, 32), (48, 48, 32, 32), (3, 32, 128, 128)]):
        x = uniform(2300 + i, (c, h, w), -1, 3, F32); up = uniform(2400 + i, (c, h, w), -1, 1, F32)
        ng = (c + gs - 1) // gs
        out, sd, mu, dest = dev.empty((c, h, w)), dev.empty((ng,)), dev.empty((ng,)), dev.empty((c, h, w)).fill_bytes(0xFF)
        call(dev, "bla_group_norm_f32", dev.to_device(x), out, sd, mu, c, gs, h * w)
        o64, sd64, mu64 = ora.group_norm(x.astype(np.float64), gs)
        assert np.allclose(mu.numpy(), mu64, rtol=2e-6, atol=1e-7) and np.allclose(sd.numpy(), sd64, rtol=5e-6)
        assert np.allclose(out.numpy(), o64, rtol=1e-5, atol=1e-6)
        call(dev, "bla_group_norm_ddx_f32", dev.to_device(up), dest, dev.to_device(x), mu, sd, c, gs, h * w)
        ref = ora.group_norm_ddx(up.astype(np.float64), x.astype(np.float64), mu64, sd64, gs)
        got = dest.numpy()
        assert np.isfinite(got).all()
        assert (np.abs(got - ref) <= 2e-5 * np.abs(ref) + 2e-5 * np.abs(ref).max()).all()


def test_implicit_gemm_conv_matches_reference(dev, ora):
    """bla_conv2d_forward/backward (im2col gathered inside the MFMA kernel, nothing materialised) against the same
    golden vectors as the staged path: conv()'s output for every stride, conv_ddx()'s del_kernels / del_input at stride 1,
    and the weight gradient against the oracle at stride 2 (where the reference's own conv_ddx is undefined)."""
    g = golden("conv")
    for i, cfg in enumerate(g["cfgs"]):
        h, w, cin, cout, k, s, seed, x, kern = conv_case(i, cfg)
        ho, wo = ora.out_hw(h, w, s)
        out = dev.empty((cout, ho, wo)).fill_bytes(0xFF)
        dx_, dk_ = dev.to_device(x), dev.to_device(kern)
        call(dev, "bla_conv2d_forward_f32", dx_, dk_, out, h, w, k, cin, cout, s)
        fw = ora.conv_intended(x.astype(np.float64), kern.astype(np.float64), s)
        bound = (np.abs(fw["im2col"]) @ np.abs(fw["kmat"])).T.reshape(cout, ho, wo)
        assert (np.abs(out.numpy() - fw["output"]) <= 1e-5 * bound + 1e-30).all(), cfg
        g.check(f"c{i}_output", out.numpy(), rtol=0, atol=1e-5 * bound.max())
        del_y = uniform(seed + 3, (cout, ho, wo), -1, 1, F32)
        dkern = dev.empty((cout, cin, k, k)).fill_bytes(0xFF)
        if s == 1:
            dxx = dev.empty((cin, h, w)).fill_bytes(0xFF); scratch = dev.empty((cout * cin * k * k,))
            call(dev, "bla_conv2d_backward_f32", dev.to_device(del_y), dx_, dk_, dkern, dxx, scratch, h, w, k, cin, cout, 1)
            dd = ora.conv_ddx_intended(del_y.astype(np.float64), fw["im2col"], fw["kmat"], cin, k)
            b1 = (np.abs(fw["im2col"]).T @ np.abs(dd["del_q"])).max()
            g.check(f"c{i}_ddx_del_kern", dkern.numpy(), rtol=0, atol=1e-5 * b1)
            b2 = (np.abs(dd["del_q"]) @ np.abs(fw["kmat"]).T).max() * k * k
            g.check(f"c{i}_ddx_del_x", dxx.numpy(), rtol=0, atol=1e-5 * b2)
        else:
            call(dev, "bla_conv2d_backward_f32", dev.to_device(del_y), dx_, dk_, dkern, None, None, h, w, k, cin, cout, s)
            dq = ora.reshape_matrix_channels(del_y.astype(np.float64))                      # [HoWo][F]
            want = ora.matrix_to_kernels(ora.matmul(ora.transpose(fw["im2col"]), dq), cin, k)   # lib/conv.c:221-223
            b1 = (np.abs(fw["im2col"]).T @ np.abs(dq)).max()
            assert (np.abs(dkern.numpy() - want) <= 1e-5 * b1).all()
            # data gradient at stride s: undefined in the reference (Q5); here the adjoint of the forward map x -> conv(x): the stride-1
            # convolution of the zero-dilated del_y with the flipped kernels.  Against the oracle composite and by <conv(x'), del_y> == <x', del_x>.
            dxx = dev.empty((cin, h, w)).fill_bytes(0xFF); scratch = dev.empty((cout * cin * k * k,))
            call(dev, "bla_conv2d_backward_f32", dev.to_device(del_y), dx_, dk_, dkern, dxx, scratch, h, w, k, cin, cout, s)
            assert (np.abs(dkern.numpy() - want) <= 1e-5 * b1).all()
            dcol64 = dq @ fw["kmat"].T; b2 = (np.abs(dq) @ np.abs(fw["kmat"]).T).max() * k * k
            assert (np.abs(dxx.numpy() - ora.col2im_adjoint(dcol64, cin, h, w, k, s)) <= 1e-5 * b2).all(), cfg
            xp = uniform(seed + 7, (cin, h, w), -1, 1, F32)
            outp = dev.empty((cout, ho, wo)); call(dev, "bla_conv2d_forward_f32", dev.to_device(xp), dk_, outp, h, w, k, cin, cout, s)
            lhs = float((outp.numpy().astype(np.float64) * del_y).sum()); rhs = float((xp.astype(np.float64) * dxx.numpy()).sum())
            assert abs(lhs - rhs) <= 1e-5 * float((np.abs(outp.numpy()).astype(np.float64) * np.abs(del_y)).sum() + np.abs(xp * dxx.numpy()).sum())


def test_strict_reference_refuses_undefined_strides():
    """BLA_STRICT_REFERENCE=1 (read once per process): _col2im / conv_ddx / the implicit data gradient at stride != 1 answer
    BLA_ERR_UNDEFINED (5) instead of the intended adjoint -- the reference indexes out of bounds there (lib/conv.c:80-135)."""
    import os, subprocess, sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from __graft_entry__ import load_pkg\n"
            "d = load_pkg(); d.init(0); L = d.lib(); z = d.zeros((4096,))\n"
            "r = [L.bla_col2im_f32(None, z.ptr, z.ptr, 8, 8, 3, 2, 2), L.bla_conv_backward_f32(None, *([z.ptr] * 9), 8, 8, 3, 2, 2, 2),\n"
            "     L.bla_conv2d_backward_f32(None, z.ptr, z.ptr, z.ptr, None, z.ptr, z.ptr, 8, 8, 3, 2, 2, 2), L.bla_col2im_f32(None, z.ptr, z.ptr, 8, 8, 3, 2, 1)]\n"
            "print(r)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BLA_STRICT_REFERENCE="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "[5, 5, 5, 0]", r.stdout + r.stderr


@pytest.mark.parametrize("batch", [2, 5, 33])
def test_batched_conv_equals_per_image_calls(dev, ora, batch):
    """bla_conv2d_*_batched: every image's output and data gradient bit-identical to the single-image entry point
    (same kernel, same K order; the image index only offsets pointers), the weight gradient = the per-image gradients
    summed, checked against the oracle's per-image conv_ddx."""
    for ci, (h, w, cin, cout, k, s) in enumerate([(8, 8, 3, 4, 3, 1), (7, 9, 2, 3, 3, 2), (16, 16, 32, 24, 3, 1)]):
        ho, wo = ora.out_hw(h, w, s)
        x = uniform(500 + ci, (batch, cin, h, w), -1, 1, F32); kern = uniform(600 + ci, (cout, cin, k, k), -0.3, 0.3, F32)
        del_y = uniform(700 + ci, (batch, cout, ho, wo), -1, 1, F32)
        dx_, dk_, dy_ = dev.to_device(x), dev.to_device(kern), dev.to_device(del_y)
        out = dev.empty((batch, cout, ho, wo)).fill_bytes(0xFF)
        call(dev, "bla_conv2d_forward_batched_f32", dx_, dk_, out, batch, h, w, k, cin, cout, s)
        got = out.numpy()
        dkern = dev.empty((cout, cin, k, k)).fill_bytes(0xFF)
        dxx = dev.empty((batch, cin, h, w)).fill_bytes(0xFF) if s == 1 else None
        scratch = dev.empty((cout * cin * k * k,)) if s == 1 else None
        call(dev, "bla_conv2d_backward_batched_f32", dy_, dx_, dk_, dkern, dxx, scratch, batch, h, w, k, cin, cout, s)
        want_dk = np.zeros((cout, cin, k, k)); bound_dk = 0.0
        for b in range(batch):
            one = dev.empty((cout, ho, wo)); xb = dev.to_device(x[b])
            call(dev, "bla_conv2d_forward_f32", xb, dk_, one, h, w, k, cin, cout, s)
            assert np.array_equal(got[b], one.numpy()), (ci, b)
            fw = ora.conv_intended(x[b].astype(np.float64), kern.astype(np.float64), s)
            dq = ora.reshape_matrix_channels(del_y[b].astype(np.float64))
            want_dk += ora.matrix_to_kernels(ora.matmul(ora.transpose(fw["im2col"]), dq), cin, k)
            bound_dk += (np.abs(fw["im2col"]).T @ np.abs(dq)).max()
            if s == 1:
                d1 = dev.empty((cin, h, w)); dk1 = dev.empty((cout, cin, k, k))
                call(dev, "bla_conv2d_backward_f32", dev.to_device(del_y[b]), xb, dk_, dk1, d1, scratch, h, w, k, cin, cout, 1)
                assert np.array_equal(dxx.numpy()[b], d1.numpy()), (ci, b)
        assert (np.abs(dkern.numpy() - want_dk) <= 1e-5 * bound_dk).all(), ci


@pytest.mark.parametrize("shape", [(64, 16, 16, 16, 64, 3, 1), (256, 16, 16, 16, 64, 3, 2), (120, 12, 12, 16, 72, 3, 1), (40, 20, 20, 32, 128, 1, 1),
                                   (128, 32, 32, 16, 128, 3, 2), (64, 32, 32, 16, 128, 3, 1), (96, 14, 18, 16, 128, 3, 2),
                                   (64, 8, 8, 128, 128, 3, 1), (64, 4, 4, 256, 256, 3, 1),    # small maps: 32 / 16 tiles, the taps cut over 8 / 16 workgroups
                                   (8, 16, 16, 128, 32, 3, 2), (4, 8, 16, 128, 48, 3, 2)])    # stride 2 with 128 input channels: data gradient by output parity
def test_batched_conv_tiled_gather_kernel(dev, ora, shape):
    """Batches big enough to fill the chip with 128x128 tiles run on the LDS-tiled gather kernel (direct-to-LDS 4-byte loads from
    computed addresses, zero padding from a zero word; or, when the output rows are multiples of four pixels, 16-byte loads from the
    zero-padded, stride-split copy -- on the half-slab pipeline for whole 128-row tiles): forward, data gradient (stride 2: the adjoint, via
    the zero-dilated del_y) and batch-summed weight gradient against the oracle's per-image conv()/conv_ddx(), GEMM tolerance
    1e-5 * (|A||B|); ragged M / N tiles, stride 2, 1x1 kernels, odd-sized maps."""
    batch, h, w, cin, cout, k, s = shape
    ho, wo = ora.out_hw(h, w, s)
    x = uniform(810, (batch, cin, h, w), -1, 1, F32); kern = uniform(811, (cout, cin, k, k), -0.3, 0.3, F32)
    del_y = uniform(812, (batch, cout, ho, wo), -1, 1, F32)
    dx_, dk_, dy_ = dev.to_device(x), dev.to_device(kern), dev.to_device(del_y)
    out = dev.empty((batch, cout, ho, wo)).fill_bytes(0xFF)
    call(dev, "bla_conv2d_forward_batched_f32", dx_, dk_, out, batch, h, w, k, cin, cout, s)
    dkern = dev.empty((cout, cin, k, k)).fill_bytes(0xFF)
    dxx = dev.empty((batch, cin, h, w)).fill_bytes(0xFF)
    scratch = dev.empty((cout * cin * k * k,))
    call(dev, "bla_conv2d_backward_batched_f32", dy_, dx_, dk_, dkern, dxx, scratch, batch, h, w, k, cin, cout, s)
    got, got_dx = out.numpy(), dxx.numpy()
    want_dk = np.zeros((cout, cin, k, k)); bound_dk = np.zeros((cout, cin, k, k))
    kmat_abs = None
    for b in range(0, batch, max(1, batch // 16)):          # every 16th image in full detail, all images for the weight gradient below
        fw = ora.conv_intended(x[b].astype(np.float64), kern.astype(np.float64), s)
        bound = (np.abs(fw["im2col"]) @ np.abs(fw["kmat"])).T.reshape(cout, ho, wo)
        assert (np.abs(got[b] - fw["output"]) <= 1e-5 * bound + 1e-30).all(), (shape, b)
        if s == 1:
            dd = ora.conv_ddx_intended(del_y[b].astype(np.float64), fw["im2col"], fw["kmat"], cin, k)
            b2 = (np.abs(dd["del_q"]) @ np.abs(fw["kmat"]).T).max() * k * k
            assert (np.abs(got_dx[b] - dd["del_x"]) <= 1e-5 * b2).all(), (shape, b)
        else:   # the intended data gradient: adjoint of _im2col applied to del_Q . kernel_matrix^T
            dq = ora.reshape_matrix_channels(del_y[b].astype(np.float64))
            b2 = (np.abs(dq) @ np.abs(fw["kmat"]).T).max() * k * k
            assert (np.abs(got_dx[b] - ora.col2im_adjoint(dq @ fw["kmat"].T, cin, h, w, k, s)) <= 1e-5 * b2).all(), (shape, b)
    # weight gradient: sum over all images, im2col via numpy strides would be long-winded -- use the oracle per image
    for b in range(batch):
        cols = ora.im2col(x[b].astype(np.float64), k, s)                                   # [HoWo][K]
        dq = ora.reshape_matrix_channels(del_y[b].astype(np.float64))                    # [HoWo][F]
        want_dk += ora.matrix_to_kernels(cols.T @ dq, cin, k)
        bound_dk += ora.matrix_to_kernels(np.abs(cols).T @ np.abs(dq), cin, k)
    assert (np.abs(dkern.numpy() - want_dk) <= 1e-5 * bound_dk + 1e-30).all(), shape


@pytest.mark.parametrize("shape", [(64, 32, 32, 3, 128, 3, 1), (64, 32, 32, 3, 128, 1, 1),     # the U-Net's first block: 3 -> 128, k 3 and the 1x1 residual (:1102)
                                   (64, 32, 32, 128, 3, 3, 1),                                  # its output convolution: 128 -> 3 (:1165)
                                   (5, 16, 12, 3, 40, 3, 1), (3, 10, 10, 50, 4, 1, 1), (2, 8, 8, 130, 2, 3, 1), (3, 16, 16, 4, 4, 3, 1), (1, 7, 9, 1, 33, 3, 1)])
def test_thin_convolutions(dev, ora, shape):
    """At most four channels on one side (csrc/bla_conv_thin.hip: direct kernels on the vector ALUs, no MFMA tile fits a 3-wide side): forward,
    data gradient (the few-inputs form on del_y with the flipped kernels, and the reverse) and the weight gradient folded over the batch in a fixed
    order, against the oracle like every other batched shape; ragged pixel blocks, channel counts that are no multiple of the quarters / groups."""
    test_batched_conv_tiled_gather_kernel(dev, ora, shape)
    batch, h, w, cin, cout, k, s = shape
    x = uniform(820, (batch, cin, h, w), -1, 1, F32); kern = uniform(821, (cout, cin, k, k), -0.3, 0.3, F32)
    outs = []
    for _ in range(2):                      # deterministic: two runs, the same bits
        out = dev.empty((batch, cout, h, w)).fill_bytes(0xFF)
        call(dev, "bla_conv2d_forward_batched_f32", dev.to_device(x), dev.to_device(kern), out, batch, h, w, k, cin, cout, s)
        outs.append(out.numpy())
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("shape", [(64, 32, 32, 128, 128, 3, 1),     # SURVEY 8(d) cfg 5 headline: (M,K,N) = (1024,1152,128) per image -- what bench.py's tertiary line times
                                   (64, 16, 16, 256, 256, 3, 1),     # the second headline shape: (256,2304,256)
                                   (64, 32, 32, 128, 256, 3, 2)])    # the U-Net's first down-convolution (model/cifar_unet.c:1105): stride 2, data gradient by output parity
def test_headline_shapes_batch_64_against_the_oracle(dev, ora, shape):
    """The sizes bench.py reports, at the size it reports them (batch of 64): forward and data gradient of four sampled images against the oracle's
    conv() / conv_ddx() (lib/conv.c:205-229; stride 2: the adjoint of _im2col), elementwise 1e-5 * (|A||B|); the batch-summed weight gradient against
    the oracle's per-image products over ALL 64 images."""
    batch, h, w, cin, cout, k, s = shape
    ho, wo = ora.out_hw(h, w, s)
    x = uniform(910, (batch, cin, h, w), -1, 1, F32); kern = uniform(911, (cout, cin, k, k), -0.1, 0.1, F32)
    del_y = uniform(912, (batch, cout, ho, wo), -1, 1, F32)
    dx_, dk_, dy_ = dev.to_device(x), dev.to_device(kern), dev.to_device(del_y)
    out = dev.empty((batch, cout, ho, wo)).fill_bytes(0xFF)
    call(dev, "bla_conv2d_forward_batched_f32", dx_, dk_, out, batch, h, w, k, cin, cout, s)
    dkern = dev.empty((cout, cin, k, k)).fill_bytes(0xFF)
    dxx = dev.empty((batch, cin, h, w)).fill_bytes(0xFF)
    scratch = dev.empty((cout * cin * k * k,))
    call(dev, "bla_conv2d_backward_batched_f32", dy_, dx_, dk_, dkern, dxx, scratch, batch, h, w, k, cin, cout, s)
    got, got_dx = out.numpy(), dxx.numpy()
    assert np.isfinite(got).all() and np.isfinite(got_dx).all()
    k64 = kern.astype(np.float64)
    for b in (0, 21, 42, 63):
        fw = ora.conv_intended(x[b].astype(np.float64), k64, s)
        bound = (np.abs(fw["im2col"]) @ np.abs(fw["kmat"])).T.reshape(cout, ho, wo)
        assert (np.abs(got[b] - fw["output"]) <= 1e-5 * bound + 1e-30).all(), (shape, b)
        assert np.linalg.norm(got[b] - fw["output"]) <= 1e-5 * np.linalg.norm(fw["output"]), (shape, b)
        dq = ora.reshape_matrix_channels(del_y[b].astype(np.float64))
        b2 = (np.abs(dq) @ np.abs(fw["kmat"]).T).max() * k * k
        want_dx = ora.conv_ddx_intended(del_y[b].astype(np.float64), fw["im2col"], fw["kmat"], cin, k)["del_x"] if s == 1 else \
            ora.col2im_adjoint(dq @ fw["kmat"].T, cin, h, w, k, s)
        assert (np.abs(got_dx[b] - want_dx) <= 1e-5 * b2).all(), (shape, b)
        assert np.linalg.norm(got_dx[b] - want_dx) <= 1e-5 * np.linalg.norm(want_dx), (shape, b)
    want_dk = np.zeros((cout, cin, k, k)); bound_dk = np.zeros((cout, cin, k, k))
    for b in range(batch):
        cols = ora.im2col(x[b].astype(np.float64), k, s)
        dq = ora.reshape_matrix_channels(del_y[b].astype(np.float64))
        want_dk += ora.matrix_to_kernels(cols.T @ dq, cin, k)
        bound_dk += ora.matrix_to_kernels(np.abs(cols).T @ np.abs(dq), cin, k)
    assert (np.abs(dkern.numpy() - want_dk) <= 1e-5 * bound_dk + 1e-30).all(), shape
    assert np.linalg.norm(dkern.numpy() - want_dk) <= 1e-5 * np.linalg.norm(want_dk), shape


@pytest.mark.parametrize("shape", [(64, 16, 16, 128, 256, 1, 1),     # 1x1 kernels on the tiled kernels (the U-Net's residual convolutions, model/cifar_unet.c:1062-1066): no shifts at all
                                   (64, 8, 8, 128, 128, 2, 1),       # even kernel: the SAME padding sits at the bottom / right only (pt = pl = 0), shifts 0 and +1
                                   (32, 8, 16, 128, 128, 3, 1),      # rows longer than the map is high; few tiles: the taps cut over workgroups
                                   (16, 9, 32, 128, 128, 3, 1),      # 288 pixels per image: 128-pixel tiles straddle two images
                                   (128, 4, 8, 128, 128, 3, 1),      # rows of two chunks: every chunk is a row end
                                   (8, 32, 32, 256, 128, 3, 1)])     # concatenated inputs (256 -> 128, the up-sampling blocks' first convolution) on 64 tiles
def test_tiled_gather_straight_from_the_image(dev, ora, shape):
    """Stride-1 geometries that stress the tiled gather paths' edge handling -- 1x1 and even kernels, tiles that straddle images, maps whose rows are one
    or two 16-byte chunks -- against the oracle like the shapes above (same checks, same tolerances).  (Written for the unpadded tap-major kernels of
    commit 77757e6, which were correct on all of them and slower than the padded copy; the shapes now run the padded / window / checked paths.)"""
    test_batched_conv_tiled_gather_kernel(dev, ora, shape)

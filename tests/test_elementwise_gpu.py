"""GPU parity of the lib/matrix.c / lib/util.c elementwise, broadcast, transpose and reduction kernels
against the reference's golden vectors (tests/golden/matrix_ops.npz) and the oracle at larger sizes.
Index-only ops (transpose) must be bit-exact; fp ops are fp32 on the device vs the fp64 reference:
|got - ref| <= 2e-6 * max(|ref|, scale) where scale is the op's natural magnitude."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden
from inputs import uniform

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(pkg):
    pkg.init(0)
    return pkg


def call(dev, name, *args):
    """DeviceArray arguments are passed by pointer and kept alive until the call has been issued
    (a temporary's bla_free would otherwise hand its memory to the next allocation)."""
    raw = [a.ptr if isinstance(a, dev.DeviceArray) else a for a in args]
    dev.native.check(getattr(dev.lib(), name)(None, *raw))


def close(got, ref, rtol=2e-6, scale=None):
    ref = np.asarray(ref, np.float64); got = np.asarray(got, np.float64).reshape(ref.shape)
    s = np.abs(ref) if scale is None else np.maximum(np.abs(ref), scale)
    bad = np.abs(got - ref) > rtol * s + 1e-30
    assert not bad.any(), f"max rel err {(np.abs(got - ref) / (s + 1e-300)).max():.3e}"


def test_golden_matrix_ops(dev, ora):
    g = golden("matrix_ops")
    for i, (r, c) in enumerate(g["shapes"]):
        r, c = int(r), int(c)
        a = uniform(300 + i, (r, c), -2, 2, np.float32); b = uniform(400 + i, (r, c), -2, 2, np.float32)
        n = r * c
        d = dev.to_device(a); call(dev, "bla_scale_f32", d, n, -0.37); close(d.numpy(), g[f"s{i}_scale"])
        d = dev.to_device(a); call(dev, "bla_add_f32", d, dev.to_device(b), n); close(d.numpy(), g[f"s{i}_add"], scale=2)
        d = dev.to_device(a); call(dev, "bla_hadamard_f32", d, dev.to_device(b), n); close(d.numpy(), g[f"s{i}_hadamard"])
        o = dev.empty((c, r)); call(dev, "bla_transpose_f32", dev.to_device(a), o, r, c)
        assert np.array_equal(o.numpy(), g[f"s{i}_transpose"].astype(np.float32))          # index-only: bit-exact
        o = dev.empty((1, c)); call(dev, "bla_row_sum_f32", dev.to_device(a), r, c, o)
        close(o.numpy(), g[f"s{i}_row_sum"], scale=np.abs(a).sum(0, keepdims=True))
        o = dev.empty((r, 1))
        da_keep = dev.to_device(a)
        st = dev.lib().bla_col_sum_f32(None, da_keep.ptr, r, c, o.ptr, 0)
        if r <= c:
            assert st == 0
            close(o.numpy(), g[f"s{i}_col_sum"], scale=np.abs(a).sum())
        else:
            assert st == 5                                                                 # BLA_ERR_UNDEFINED (Q2)
        call(dev, "bla_col_sum_f32", dev.to_device(a), r, c, o, 1)
        close(o.numpy(), ora.col_sum_intended(a.astype(np.float64)), scale=np.abs(a).sum(1, keepdims=True))
        s = dev.empty((4,)); call(dev, "bla_frobenius_f32", dev.to_device(a), n, s); close(s.numpy()[0], g[f"s{i}_frobenius"])
        call(dev, "bla_max_f32", dev.to_device(a), n, s); assert s.numpy()[0] == np.float32(g[f"s{i}_max"])
        d = dev.to_device(a); call(dev, "bla_zscore_f32", d, n); close(d.numpy(), g[f"s{i}_zscore"], rtol=1e-5, scale=1)
        d = dev.to_device(a); call(dev, "bla_add_tile_columns_f32", d, r, c, dev.to_device(uniform(500 + i, (r, 1), dtype=np.float32)), 1)
        close(d.numpy(), g[f"s{i}_tile_cols"], scale=2)
        d = dev.to_device(a); call(dev, "bla_add_tile_rows_f32", d, r, c, dev.to_device(uniform(600 + i, (1, c), dtype=np.float32)))
        close(d.numpy(), g[f"s{i}_tile_rows"], scale=2)
        if c % 2 == 0:
            d = dev.to_device(a); call(dev, "bla_add_tile_columns_f32", d, r, c, dev.to_device(uniform(700 + i, (r, 2), dtype=np.float32)), 2)
            close(d.numpy(), g[f"s{i}_tile_cols2"], scale=2)
        d = dev.to_device(a); call(dev, "bla_relu_f32", d, n)
        assert np.array_equal(d.numpy(), g[f"s{i}_relu"].astype(np.float32))
        d = dev.to_device(a); call(dev, "bla_relu_ddx_f32", d, n)
        assert np.array_equal(d.numpy(), (a > 0).astype(np.float32))
        d = dev.to_device(a * 4); call(dev, "bla_softmax_cols_f32", d, r, c); close(d.numpy(), g[f"s{i}_softmax_cols"], rtol=1e-5, scale=1e-6)
        d = dev.to_device(a * 4); call(dev, "bla_softmax_rows_f32", d, r, c); close(d.numpy(), g[f"s{i}_softmax_rows"], rtol=1e-5, scale=1e-6)


@pytest.mark.parametrize("shape", [(1, 1), (1, 1000), (1000, 1), (257, 129), (784, 256), (2048, 3000), (300, 9000), (515, 4100), (64, 36000)])
def test_larger_shapes_vs_oracle(dev, ora, shape):
    r, c = shape; n = r * c
    a = uniform(1, shape, -3, 3, np.float32); b = uniform(2, shape, -3, 3, np.float32)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    d = dev.to_device(a); call(dev, "bla_axpy_f32", d, dev.to_device(b), -0.02, n)
    close(d.numpy(), a64 + np.float64(np.float32(-0.02)) * b64, scale=3)
    o = dev.empty((c, r)); call(dev, "bla_transpose_f32", dev.to_device(a), o, r, c)
    assert np.array_equal(o.numpy(), a.T)
    o = dev.empty((1, c)); call(dev, "bla_row_sum_f32", dev.to_device(a), r, c, o)
    close(o.numpy(), ora.row_sum(a64), scale=np.abs(a64).sum(0, keepdims=True))
    o = dev.empty((r, 1)); call(dev, "bla_col_sum_f32", dev.to_device(a), r, c, o, 1)
    close(o.numpy(), ora.col_sum_intended(a64), scale=np.abs(a64).sum(1, keepdims=True))
    s = dev.empty((4,)); call(dev, "bla_frobenius_f32", dev.to_device(a), n, s); close(s.numpy()[0], ora.frobenius(a64))
    call(dev, "bla_max_f32", dev.to_device(a), n, s); assert s.numpy()[0] == a.max()
    d = dev.to_device(a); call(dev, "bla_softmax_cols_f32", d, r, c); close(d.numpy(), ora.softmax_cols(a64), rtol=1e-5, scale=1e-6)
    d = dev.to_device(a); call(dev, "bla_softmax_rows_f32", d, r, c); close(d.numpy(), ora.softmax_rows(a64), rtol=1e-5, scale=1e-6)
    if n > 1:
        d = dev.to_device(a); call(dev, "bla_zscore_f32", d, n); close(d.numpy(), ora.zscore(a64), rtol=2e-5, scale=1)


def test_unaligned_views_and_empty(dev):
    """Odd element offsets force the scalar path; n = 0 is a no-op."""
    a = uniform(1, (1, 1003), dtype=np.float32)
    d = dev.to_device(a)
    call(dev, "bla_scale_f32", d.ptr + 4, 1001, 2.0)       # base 4-byte aligned only
    want = a.copy(); want[0, 1:1002] *= 2
    assert np.array_equal(d.numpy(), want)
    call(dev, "bla_scale_f32", d, 0, 5.0)
    assert np.array_equal(d.numpy(), want)


@pytest.mark.parametrize("rows,cols", [(10, 300), (300, 260)])   # strided kernel / streaming three-launch path
def test_softmax_grad_fused(dev, ora, rows, cols):
    z = uniform(1, (rows, cols), -4, 4, np.float32)
    y = np.zeros((rows, cols), np.float32); y[np.arange(cols) % rows, np.arange(cols)] = 1
    d = dev.to_device(z); g = dev.empty((rows, cols))
    call(dev, "bla_softmax_cols_grad_f32", d, rows, cols, dev.to_device(y), 1 / 784, g)
    p = ora.softmax_cols(z.astype(np.float64))
    close(d.numpy(), p, rtol=1e-5, scale=1e-6)
    close(g.numpy(), ora.scale(ora.add(p, -y.astype(np.float64)), 1 / 784), rtol=1e-5, scale=1e-8)


@pytest.mark.parametrize("rows,cols", [(300, 1024), (1500, 1040), (4096, 2048), (2049, 1024)])
def test_softmax_cols_strip_in_registers(dev, ora, rows, cols):
    """rows <= 4096, cols % 16 == 0, >= 64 strips: a workgroup keeps its 16 columns in registers (one read, one write per element: lib/util.c:15-34 in one
    pass over memory).  Ragged row counts, a strip count that is not a multiple of 8 (plain strip order), every register depth; with and without the fused
    loss-gradient tail of model/mnist_nn.c:263-268."""
    z = uniform(5, (rows, cols), -6, 6, np.float32)
    y = np.zeros((rows, cols), np.float32); y[np.arange(cols) % rows, np.arange(cols)] = 1
    p = ora.softmax_cols(z.astype(np.float64))
    d = dev.to_device(z); call(dev, "bla_softmax_cols_f32", d, rows, cols)
    close(d.numpy(), p, rtol=1e-5, scale=1e-6)
    assert np.abs(d.numpy().sum(0) - 1).max() < 1e-5
    d = dev.to_device(z); g = dev.empty((rows, cols))
    call(dev, "bla_softmax_cols_grad_f32", d, rows, cols, dev.to_device(y), 1 / 784, g)
    close(d.numpy(), p, rtol=1e-5, scale=1e-6)
    close(g.numpy(), ora.scale(ora.add(p, -y.astype(np.float64)), 1 / 784), rtol=1e-5, scale=1e-8)

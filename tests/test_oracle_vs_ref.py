"""Direct pin: oracle (fp64) == the reference itself (oracle/_ref/libref.so, compiled from
/root/reference/lib/*.c by oracle/Makefile) bit for bit on fresh random shapes.
Skipped where no reference build exists (the golden-vector tests still pin the oracle there)."""
import ctypes as C

import numpy as np
import pytest

import ref
from inputs import uniform

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libref.so not built")


@pytest.mark.parametrize("seed", range(12))
def test_matmul_random(ora, seed):
    rng = np.random.default_rng(seed)
    m, k, n = [int(v) for v in rng.integers(1, 90, 3)]
    a = uniform(seed * 3 + 1, (m, k), -3, 3); b = uniform(seed * 3 + 2, (k, n), -3, 3)
    assert np.array_equal(ora.matmul(a, b), ref.matmul(a, b))
    assert np.array_equal(ora.matmul(a, b), ref.matmul_inplace(a, b))


@pytest.mark.parametrize("shape", [(1, 1), (5, 9), (9, 5), (31, 64), (64, 64)])
def test_elementwise_and_reductions(ora, shape):
    L = ref.lib()
    a = uniform(11, shape, -2, 2); b = uniform(12, shape, -2, 2)
    assert np.array_equal(ora.scale(a, 1 / 3), ref.inplace1("matrix_scale", a, C.c_double(1 / 3)))
    assert np.array_equal(ora.add(a, b), ref.inplace2("matrix_add", a, b))
    assert np.array_equal(ora.hadamard(a, b), ref.inplace2("matrix_multiply_elementwise", a, b))
    assert np.array_equal(ora.transpose(a), ref.inplace1("matrix_transpose", a))
    assert np.array_equal(ora.row_sum(a), ref.take(L.matrix_row_sum(ref.mat(a))))
    if shape[0] <= shape[1]:
        assert np.array_equal(ora.col_sum_as_written(a), ref.take(L.matrix_col_sum(ref.mat(a))))
    assert ora.frobenius(a) == L.frobenius_norm(ref.mat(a))
    assert ora.max_value(a) == L.max_value(ref.mat(a))
    assert np.array_equal(ora.zscore(a), ref.inplace1("matrix_z_score_normalize", a), equal_nan=True)  # 1x1 is 0/0 in both
    assert np.array_equal(ora.relu(a), ref.data_fn("relu", a, a.size))
    assert np.array_equal(ora.softmax_cols(a), ref.data_fn("softmax", a, *shape))
    assert np.array_equal(ora.softmax_rows(a), ref.data_fn("softmax_row_wise", a, *shape))


@pytest.mark.parametrize("cfg", [(4, 4, 1, 1, 3, 1), (9, 6, 2, 3, 3, 2), (10, 10, 3, 2, 1, 2), (5, 5, 2, 2, 3, 3)])
def test_im2col_geometry(ora, cfg):
    h, w, cin, cout, k, s = cfg
    x = uniform(21, (cin, h, w))
    ho, wo = ora.out_hw(h, w, s)
    im = np.zeros((ho * wo, k * k * cin)); imm = ref.mat(im)
    ref.lib()._im2col(ref.mats(x), C.byref(imm), k, cin, s)
    assert np.array_equal(ora.im2col(x, k, s), im)

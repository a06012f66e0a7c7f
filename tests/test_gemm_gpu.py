"""GPU parity of bla_gemm_f32 (through the C-ABI) against the golden vectors the reference
produced and against the CPU oracle on seeded inputs.

Tolerance (SURVEY 8c / DESIGN.md): fp32 MFMA vs the fp64 reference,
    normwise   ||C - ref||_F / ||ref||_F <= 1e-5
    elementwise |C - ref|_ij <= 1e-5 * (|A| |B|)_ij   (guards cancellation near zero)
"""
import ctypes as C
import numpy as np
import pytest

from conftest import golden
from inputs import uniform

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def dev(pkg):
    pkg.init(0)
    pkg.lib().bla_gemm_set_config(-1, 0)
    return pkg


def check_gemm(ora, c, a, b, ref=None, tag=""):
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    if ref is None:
        ref = ora.matmul(a64, b64)
    bound = np.abs(a64) @ np.abs(b64)
    err = np.abs(c.astype(np.float64) - ref)
    assert (err <= RTOL * bound + 1e-30).all(), f"{tag}: elementwise excess {(err - RTOL*bound).max():.3e}"
    nr = np.linalg.norm(ref)
    if nr > 0:
        assert np.linalg.norm(err) / nr <= RTOL, f"{tag}: normwise {np.linalg.norm(err)/nr:.3e}"


def run(dev, a, b, transa=False, transb=False, **kw):
    m = a.shape[1] if transa else a.shape[0]
    n = b.shape[0] if transb else b.shape[1]
    c = dev.empty((m, n)).fill_bytes(0xFF)   # NaN pattern: every element must be written
    dev.gemm(dev.to_device(a), dev.to_device(b), c, transa=transa, transb=transb, **kw)
    return c.numpy()


def test_known_answers(dev, ora):
    g = golden("gemm")
    c = run(dev, g["kat_main_a"].astype(np.float32), g["kat_main_b"].astype(np.float32))
    assert np.allclose(c, [[1.4, 8.5], [5.0, 19.0]], rtol=1e-6)          # main.c:20-41
    check_gemm(ora, c, g["kat_main_a"].astype(np.float32), g["kat_main_b"].astype(np.float32), g["kat_main_c"], "kat")
    c = run(dev, g["csv_a"].astype(np.float32), g["csv_b"].astype(np.float32))
    check_gemm(ora, c, g["csv_a"].astype(np.float32), g["csv_b"].astype(np.float32), g["csv_c"], "csv")
    assert np.allclose(c, [[11.5041, 2.3], [2328.603, 9.2], [43.536, 16.1]], rtol=1e-6)


def test_golden_random_shapes(dev, ora):
    g = golden("gemm")
    for i, (m, k, n) in enumerate(g["shapes"]):
        a = uniform(100 + i, (m, k), dtype=np.float32); b = uniform(200 + i, (k, n), dtype=np.float32)
        check_gemm(ora, run(dev, a, b), a, b, g[f"rand{i}_c"], f"rand{i} {m}x{k}x{n}")


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_all_layouts_and_tiles(dev, ora, cfg, ta, tb):
    """Every transpose combination on every tile configuration, sizes straddling tile edges.
    Configs 3-5 and 7-9 are the direct-to-LDS kernels: they need k % BK == 0 and extents % 4 == 0
    (8 and 9 are the three-buffer split-fragment pipeline; odd and even slab counts take different tails).
    Config 6 is the wave-split-K kernel for latency-bound shapes (any shape, k > 0), 16 its 16x16-tile form (16-byte loads:
    contiguous extents % 4 == 0; ragged tiles, K tails shorter than a load group and single-wave K included)."""
    dev.lib().bla_gemm_set_config(cfg, 0)
    shapes = [(1, 1, 1), (5, 3, 7), (64, 16, 64), (65, 17, 63), (130, 40, 129), (128, 128, 128), (257, 100, 31), (200, 260, 136)]
    if cfg == 16:
        shapes = [(4, 4, 4), (16, 16, 16), (20, 12, 36), (64, 64, 64), (68, 20, 60), (132, 100, 128), (128, 784, 48), (256, 256, 128), (260, 136, 36), (12, 1000, 52)]
    if 3 <= cfg <= 5 or cfg >= 7:
        shapes = [(4, 32, 4), (64, 32, 64), (68, 64, 60), (132, 96, 128), (128, 128, 128), (260, 160, 36), (200, 256, 136),
                  (384, 512, 256), (516, 16, 260), (300, 48, 520), (256, 80, 128), (132, 112, 140), (128, 144, 128)]
        if cfg in (5, 10, 18):   # BK = 32, or the persistent kernel's even slab count
            shapes = [s for s in shapes if s[1] % 32 == 0]
        if cfg == 11:        # 256x256 tiles, whole tiles only
            shapes = [(256, 32, 256), (512, 48, 256), (256, 160, 768), (512, 512, 512)]
        if cfg == 15:        # 128x256 tiles, half-slab pipeline, 2x4 blocks per wave
            shapes = [(128, 32, 256), (256, 48, 512), (128, 160, 768), (384, 256, 256)]
        if cfg == 17:        # 192x192 tiles, half-slab pipeline, 3x3 blocks per wave
            shapes = [(192, 32, 192), (384, 48, 192), (192, 160, 576), (384, 256, 384)]
        if cfg == 14:        # 128x128 tiles, half-slab pipeline
            shapes = [(128, 32, 128), (256, 48, 384), (128, 160, 640), (384, 256, 512)]
        if cfg == 13:        # 128x512 tiles
            shapes = [(128, 32, 512), (256, 48, 512), (128, 160, 1024), (384, 256, 512)]
        if cfg == 12:        # the same with 32-deep slabs
            shapes = [(256, 64, 256), (512, 96, 256), (256, 160, 768), (512, 512, 512)]
    try:
        for j, (m, k, n) in enumerate(shapes):
            a = uniform(10 + j, (k, m) if ta else (m, k), dtype=np.float32)
            b = uniform(50 + j, (n, k) if tb else (k, n), dtype=np.float32)
            c = run(dev, a, b, transa=bool(ta), transb=bool(tb))
            check_gemm(ora, c, a.T if ta else a, b.T if tb else b, tag=f"cfg{cfg} ta{ta} tb{tb} {m}x{k}x{n}")
    finally:
        dev.lib().bla_gemm_set_config(-1, 0)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0)])
def test_persistent_kernel_many_tiles_per_workgroup(dev, ta, tb):
    """Config 10 walks several tiles per workgroup once there are more tiles than 2 x CUs: 25 x 25 tiles with ragged
    edges here, so the slab stream crosses tile boundaries (prefetch of the next tile under the current one's tail,
    stores of the finished tile under the next one's MFMAs).  Checked against a float64 product."""
    m, k, n = 3100, 96, 3148   # extents % 4 == 0 (16-byte DMA granules), not multiples of the tile
    a = uniform(31, (k, m) if ta else (m, k), dtype=np.float32)
    b = uniform(32, (n, k) if tb else (k, n), dtype=np.float32)
    dev.lib().bla_gemm_set_config(10, 0)
    try:
        da, db = dev.to_device(a), dev.to_device(b)
        c = dev.zeros((m, n))
        dev.gemm(da, db, c, transa=bool(ta), transb=bool(tb))
        assert "glds128x128x16p" in dev.lib().bla_gemm_last_kernel().decode()
        got = c.numpy()
    finally:
        dev.lib().bla_gemm_set_config(-1, 0)
    A = (a.T if ta else a).astype(np.float64); B = (b.T if tb else b).astype(np.float64)
    ref = A @ B
    bound = np.abs(A) @ np.abs(B)
    assert np.all(np.abs(got - ref) <= 1e-5 * bound + 1e-30)
    assert np.linalg.norm(got - ref) <= 1e-5 * np.linalg.norm(ref)


@pytest.mark.parametrize("split", [2, 3, 7])
def test_split_k(dev, ora, split):
    dev.lib().bla_gemm_set_config(1, split)
    try:
        for (m, k, n) in [(10, 784, 33), (70, 1000, 70), (64, 128, 64)]:
            a = uniform(7, (m, k), dtype=np.float32); b = uniform(8, (k, n), dtype=np.float32)
            check_gemm(ora, run(dev, a, b), a, b, tag=f"split{split} {m}x{k}x{n}")
            assert f"splitk" in dev.lib().bla_gemm_last_kernel().decode()
    finally:
        dev.lib().bla_gemm_set_config(-1, 0)


def test_unaligned_leading_dimensions(dev, ora):
    """ld not a multiple of 4 -> scalar-load variant; sub-matrix views via lda/ldb/ldc."""
    big_a = uniform(1, (70, 91), dtype=np.float32); big_b = uniform(2, (91, 53), dtype=np.float32)
    m, k, n = 60, 80, 45
    da, db = dev.to_device(big_a), dev.to_device(big_b)
    c = dev.zeros((m, 50))
    dev.gemm(da, db, c, m=m, n=n, k=k, lda=91, ldb=53, ldc=50)
    out = c.numpy()
    check_gemm(ora, out[:, :n], big_a[:m, :k], big_b[:k, :n], tag="submatrix")
    assert (out[:, n:] == 0).all()       # nothing outside the m x n window is touched
    name = dev.lib().bla_gemm_last_kernel().decode()
    assert "_scalar" in name or "_ss_" in name          # 4-byte loads on both operands


def test_epilogue(dev, ora):
    m, k, n = 96, 200, 130
    a = uniform(1, (m, k), dtype=np.float32); b = uniform(2, (k, n), dtype=np.float32)
    br = uniform(3, (m, 1), dtype=np.float32); bc = uniform(4, (1, n), dtype=np.float32)
    mask = uniform(5, (m, n), dtype=np.float32); c0 = uniform(6, (m, n), dtype=np.float32)
    ref = ora.matmul(a.astype(np.float64), b.astype(np.float64))
    bound = np.abs(a.astype(np.float64)) @ np.abs(b.astype(np.float64))
    # Z = W X + b (tile_columns), A = relu(Z): model/mnist_nn.c:221-224
    z = dev.empty((m, n)); c = dev.empty((m, n))
    dev.gemm(dev.to_device(a), dev.to_device(b), c, bias_row=dev.to_device(br), pre_act=z, act=dev.ACT_RELU)
    zr = ora.add_tile_columns(ref, br.astype(np.float64))
    assert (np.abs(z.numpy() - zr) <= RTOL * (bound + np.abs(br))).all()
    assert (np.abs(c.numpy() - ora.relu(zr)) <= RTOL * (bound + np.abs(br))).all()
    assert (c.numpy() >= 0).all()
    # bias per column (tile_rows), relu' mask and alpha/beta accumulate
    c = dev.to_device(c0)
    dev.gemm(dev.to_device(a), dev.to_device(b), c, alpha=-0.5, beta=2.0, bias_col=dev.to_device(bc), relu_mask=dev.to_device(mask))
    want = ora.hadamard(ora.add_tile_rows(-0.5 * ref, bc.astype(np.float64)), ora.relu_ddx(mask.astype(np.float64))) + 2.0 * c0
    assert (np.abs(c.numpy() - want) <= RTOL * (bound + np.abs(bc) + 2 * np.abs(c0))).all()


def test_argument_errors(dev):
    a = dev.zeros((4, 4))
    with pytest.raises(dev.BlaError) as e:
        dev.gemm(a, a, a, lda=2)
    assert e.value.status == 1
    with pytest.raises(dev.BlaError):
        dev.gemm(dev.zeros((4, 5)), dev.zeros((4, 4)), a)     # inner dimensions differ
    dev.gemm(dev.zeros((0, 4)), a, dev.zeros((0, 4)))           # empty output: no-op, no error
    # k == 0: empty sum -> zeros
    c = dev.empty((3, 3)).fill_bytes(0xFF)
    dev.gemm(dev.zeros((3, 0)), dev.zeros((0, 3)), c)
    assert (c.numpy() == 0).all()


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192])   # every size of BASELINE configs[1]
def test_large_square_sampled_rows(dev, ora, n):
    """BASELINE config 2 sizes: oracle on a sample of output rows (full oracle would take minutes),
    plus exactness properties that hold at any size: A @ I == A bit-for-bit and linearity in alpha."""
    a = uniform(0xB1A5, (n, n), dtype=np.float32); b = uniform(0xB1A6, (n, n), dtype=np.float32)
    da, db, dc = dev.to_device(a), dev.to_device(b), dev.empty((n, n))
    c = dev.gemm(da, db, dc).numpy()
    rows = [0, 1, n // 2 - 1, n // 2, n - 129, n - 1]
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    ref = np.zeros((n, n))
    for r in rows:
        ora.matmul_rows(a64, b64, ref, r, r + 1)
    bound = np.abs(a64[rows]) @ np.abs(b64)
    err = np.abs(c[rows] - ref[rows])
    assert (err <= RTOL * bound).all(), (err / bound).max()
    assert np.linalg.norm(err) / np.linalg.norm(ref[rows]) <= RTOL
    eye = dev.to_device(np.eye(n, dtype=np.float32))
    assert np.array_equal(dev.gemm(da, eye, dc).numpy(), a)                 # products with 0/1 are exact
    c2 = dev.gemm(da, db, dc, alpha=2.0).numpy()
    assert np.array_equal(c2, 2 * c)                                         # power-of-two scaling is exact


@pytest.mark.parametrize("cfg", [-1, 1, 4, 6, 16])
def test_fused_row_sum_of_a(dev, ora, cfg):
    """row_sum_a[r] = sum_k A[r][k] rides along the product (bias gradient = true row sums of dZ, the intent of
    matrix_col_sum, model/mnist_nn.c:271): fused in the wave-split-K kernels, a separate pass behind the tiled ones."""
    dev.lib().bla_gemm_set_config(cfg, 0)
    try:
        for (m, k, n) in [(10, 256, 128), (128, 256, 256), (70, 96, 40)] if cfg != 16 else [(10, 256, 128), (128, 256, 256), (72, 96, 40), (256, 256, 784)]:
            a = uniform(31, (m, k), dtype=np.float32); b = uniform(32, (n, k), dtype=np.float32)
            rs = dev.empty((m,)).fill_bytes(0xFF); c = dev.empty((m, n))
            dev.gemm(dev.to_device(a), dev.to_device(b), c, transb=True, row_sum_a=rs)
            check_gemm(ora, c.numpy(), a, b.T, tag=f"cfg{cfg} {m}x{k}x{n}")
            want = ora.col_sum_intended(a.astype(np.float64)).ravel()
            assert (np.abs(rs.numpy() - want) <= 2e-6 * np.abs(a).sum(1)).all()
    finally:
        dev.lib().bla_gemm_set_config(-1, 0)


@pytest.mark.parametrize("cfg", [6, 16])
def test_fused_softmax_tail(dev, ora, cfg):
    """Output layer in one launch: Z = W A + b (kept), P = softmax per column, grad = (P - Y) * scale
    (model/mnist_nn.c:231-234,260-268); on 32x32 tiles (m <= 32) and on 16x16 tiles (m <= 16)."""
    dev.lib().bla_gemm_set_config(cfg, 0)
    try:
        _fused_softmax_tail(dev, ora)
    finally:
        dev.lib().bla_gemm_set_config(-1, 0)


def _fused_softmax_tail(dev, ora):
    m, k, n = 10, 128, 300
    w = uniform(41, (m, k), dtype=np.float32); x = uniform(42, (k, n), -2, 2, np.float32); b = uniform(43, (m, 1), dtype=np.float32)
    y = np.zeros((m, n), np.float32); y[np.arange(n) % m, np.arange(n)] = 1
    z, p, g = dev.empty((m, n)), dev.empty((m, n)), dev.empty((m, n))
    dev.gemm(dev.to_device(w), dev.to_device(x), p, bias_row=dev.to_device(b), pre_act=z, softmax_y=dev.to_device(y),
             softmax_scale=1 / 784, softmax_grad=g)
    z64 = ora.add_tile_columns(ora.matmul(w.astype(np.float64), x.astype(np.float64)), b.astype(np.float64))
    bound = np.abs(w.astype(np.float64)) @ np.abs(x.astype(np.float64)) + np.abs(b)
    assert (np.abs(z.numpy() - z64) <= 1e-5 * bound).all()
    p64 = ora.softmax_cols(z64)
    assert np.allclose(p.numpy(), p64, rtol=1e-4, atol=1e-7)
    assert np.allclose(g.numpy(), (p64 - y) / 784, rtol=1e-4, atol=1e-9)
    with pytest.raises(dev.BlaError):     # m > 32 cannot hold whole columns in one tile
        dev.gemm(dev.zeros((40, 8)), dev.zeros((8, 8)), dev.zeros((40, 8)), softmax_y=dev.zeros((40, 8)), softmax_grad=dev.zeros((40, 8)))


@pytest.mark.parametrize("split", [2, 3, 5, 8])
def test_wave_split_k_with_in_launch_fold(dev, ora, split):
    """Latency-bound kernel with K also cut over workgroups: partial tiles are published, the last arriver per tile
    (arrival counter, agent-scope release/acquire) folds them in split order and runs the epilogue in the same launch.
    Repeated launches reuse the counters (the last arriver resets them) and must stay bit-identical (deterministic fold)."""
    dev.lib().bla_gemm_set_config(6, split)
    try:
        for (m, k, n, ta, tb) in [(256, 784, 256, 0, 0), (70, 1000, 45, 0, 1), (33, 640, 200, 1, 0), (10, 2304, 64, 0, 1)]:
            a = uniform(61, (k, m) if ta else (m, k), dtype=np.float32); b = uniform(62, (n, k) if tb else (k, n), dtype=np.float32)
            bias = uniform(63, (m, 1), dtype=np.float32)
            da, db, dbias = dev.to_device(a), dev.to_device(b), dev.to_device(bias)
            z = dev.empty((m, n)); c = dev.empty((m, n)).fill_bytes(0xFF)
            outs = []
            for rep in range(4):
                dev.gemm(da, db, c, transa=bool(ta), transb=bool(tb), bias_row=dbias, pre_act=z, act=dev.ACT_RELU)
                outs.append(c.numpy().copy())
            assert "ksplit" in dev.lib().bla_gemm_last_kernel().decode() and not dev.lib().bla_gemm_last_kernel().decode().endswith("ksplit1")
            for o in outs[1:]:
                assert np.array_equal(o, outs[0])
            a64 = (a.T if ta else a).astype(np.float64); b64 = (b.T if tb else b).astype(np.float64)
            zr = ora.add_tile_columns(ora.matmul(a64, b64), bias.astype(np.float64))
            bound = np.abs(a64) @ np.abs(b64) + np.abs(bias)
            assert (np.abs(z.numpy() - zr) <= RTOL * bound).all()
            assert (np.abs(outs[0] - ora.relu(zr)) <= RTOL * bound).all()
    finally:
        dev.lib().bla_gemm_set_config(-1, 0)


def test_gemm_pair_shares_one_launch(dev, ora):
    """bla_gemm_pair_f32: an NT product beside a TN product (dW_l beside dZ_{l-1} of model/mnist_nn.c:267-289) in one launch,
    both with their epilogues (row sums of A / relu' mask); products that do not fit the latency-bound kernel fall back to two launches.
    Same results either way."""
    nat = dev.native
    for (m1, n1, k1, m2, n2, k2) in [(10, 128, 256, 128, 256, 10), (128, 256, 256, 256, 256, 128), (33, 40, 64, 70, 36, 33)]:
        a1 = uniform(1, (m1, k1), dtype=np.float32); b1 = uniform(2, (n1, k1), dtype=np.float32)      # NT: A [m][k], B [n][k]
        a2 = uniform(3, (k2, m2), dtype=np.float32); b2 = uniform(4, (k2, n2), dtype=np.float32)      # TN: A [k][m], B [k][n]
        z = uniform(5, (m2, n2), dtype=np.float32)
        d = [dev.to_device(x) for x in (a1, b1, a2, b2, z)]
        c1, c2, rs = dev.zeros((m1, n1)), dev.zeros((m2, n2)), dev.zeros((m1,))
        e1 = nat.Epilogue(1.0, 0.0, None, None, None, 0, 0, None, 0, rs.ptr, None, 0.0, None)
        e2 = nat.Epilogue(1.0, 0.0, None, None, None, 0, 0, d[4].ptr, n2, None, None, 0.0, None)
        p = nat.gemm_desc(d[0], d[1], c1, transb=True, epilogue=e1)
        q = nat.gemm_desc(d[2], d[3], c2, transa=True, epilogue=e2)
        nat.gemm_pair(p, q)
        name = dev.lib().bla_gemm_last_kernel().decode()
        check_gemm(ora, c1.numpy(), a1, b1.T, tag=f"pair NT {m1}x{k1}x{n1} ({name})")
        want2 = (a2.T.astype(np.float64) @ b2.astype(np.float64)) * (z > 0)
        bound = np.abs(a2.T).astype(np.float64) @ np.abs(b2).astype(np.float64)
        assert np.all(np.abs(c2.numpy() - want2) <= 1e-5 * bound + 1e-30), name
        np.testing.assert_allclose(rs.numpy(), a1.astype(np.float64).sum(1), rtol=1e-5, atol=1e-5)
    # aligned shapes of the trainer do share the launch
    a1 = dev.to_device(uniform(1, (128, 256), dtype=np.float32)); b1 = dev.to_device(uniform(2, (256, 256), dtype=np.float32))
    a2 = dev.to_device(uniform(3, (128, 256), dtype=np.float32)); b2 = dev.to_device(uniform(4, (128, 256), dtype=np.float32))
    c1, c2 = dev.zeros((128, 256)), dev.zeros((256, 256))
    nat.gemm_pair(nat.gemm_desc(a1, b1, c1, transb=True), nat.gemm_desc(a2, b2, c2, transa=True))
    assert "pair_nt+tn" in dev.lib().bla_gemm_last_kernel().decode()


def test_gemm_pair_every_layout_combination(dev, ora):
    """All 16 (op(A), op(B)) x (op(A), op(B)) combinations of two latency-bound products share one launch (the attention block of
    model/cifar_unet.c:999-1022,1261-1337 pairs TN+TN, TT+TT, NN+TN, NN+NN, NN+NT ...) and give bit-for-bit what two separate calls give."""
    nat = dev.native
    m1, n1, k1, m2, n2, k2 = 64, 96, 128, 160, 32, 64
    for ta1 in (False, True):
        for tb1 in (False, True):
            for ta2 in (False, True):
                for tb2 in (False, True):
                    a1 = uniform(11, (k1, m1) if ta1 else (m1, k1), dtype=np.float32); b1 = uniform(12, (n1, k1) if tb1 else (k1, n1), dtype=np.float32)
                    a2 = uniform(13, (k2, m2) if ta2 else (m2, k2), dtype=np.float32); b2 = uniform(14, (n2, k2) if tb2 else (k2, n2), dtype=np.float32)
                    d = [dev.to_device(x) for x in (a1, b1, a2, b2)]
                    c1, c2, s1, s2 = dev.zeros((m1, n1)), dev.zeros((m2, n2)), dev.zeros((m1, n1)), dev.zeros((m2, n2))
                    nat.gemm_pair(nat.gemm_desc(d[0], d[1], c1, transa=ta1, transb=tb1), nat.gemm_desc(d[2], d[3], c2, transa=ta2, transb=tb2))
                    name = dev.lib().bla_gemm_last_kernel().decode()
                    want = "pair_" + ("t" if ta1 else "n") + ("t" if tb1 else "n") + "+" + ("t" if ta2 else "n") + ("t" if tb2 else "n")
                    assert want in name, (want, name)
                    dev.gemm(d[0], d[1], s1, transa=ta1, transb=tb1); dev.gemm(d[2], d[3], s2, transa=ta2, transb=tb2)
                    assert np.array_equal(c1.numpy(), s1.numpy()) and np.array_equal(c2.numpy(), s2.numpy()), name
                    check_gemm(ora, c1.numpy(), a1.T if ta1 else a1, b1.T if tb1 else b1, tag=name)
                    check_gemm(ora, c2.numpy(), a2.T if ta2 else a2, b2.T if tb2 else b2, tag=name)


def test_recorded_sequence_replays_on_new_data(dev, ora):
    """bla_graph_*: a product with a bias/ReLU epilogue followed by a scale, recorded once and replayed after the inputs changed."""
    L, chk = dev.lib(), dev.native.check
    st = L.bla_default_stream()
    m, k, n = 96, 200, 130
    a = dev.to_device(uniform(1, (m, k), dtype=np.float32)); b = dev.to_device(uniform(2, (k, n), dtype=np.float32))
    bias = dev.to_device(uniform(3, (m,), dtype=np.float32)); c = dev.zeros((m, n))

    def seq():
        dev.gemm(a, b, c, bias_row=bias, act=dev.native.ACT_RELU, stream=st)
        chk(L.bla_scale_f32(st, c.ptr, m * n, 0.5))
    seq(); dev.native.sync(st)
    g = C.c_void_p()
    chk(L.bla_graph_begin(st)); seq(); chk(L.bla_graph_end(st, C.byref(g)))
    for seed in (10, 20):
        a2 = uniform(seed, (m, k), dtype=np.float32); b2 = uniform(seed + 1, (k, n), dtype=np.float32)
        a.copy_from(a2); b.copy_from(b2)
        chk(L.bla_graph_launch(g, st)); dev.native.sync(st)
        want = 0.5 * np.maximum(a2.astype(np.float64) @ b2.astype(np.float64) + bias.numpy().astype(np.float64)[:, None], 0)
        bound = np.abs(a2).astype(np.float64) @ np.abs(b2).astype(np.float64) + 1
        assert np.all(np.abs(c.numpy() - want) <= 1e-5 * bound)
    chk(L.bla_graph_destroy(g))


def test_automatic_dispatch_fuzz(dev):
    """Random shapes, layouts, leading dimensions (sub-matrix views) and epilogues through the AUTOMATIC configuration choice -- whatever
    kernel the heuristics pick (wave-split-K, the three DMA tiles, the 256x256 kernel on whole tiles, split-K slabs, generic path) must
    agree with a float64 product.  Dimensions are drawn so that every family is hit, including multiples of 256 with padded pitches."""
    rng = np.random.default_rng(20250)
    seen = set()
    pools = [[1, 3, 17, 33, 64, 100], [128, 200, 256, 260, 384], [512, 768, 1024, 1280], [2048, 2304, 4096]]
    for case in range(70):
        cls = rng.integers(0, 4)
        m = int(rng.choice(pools[cls])); n = int(rng.choice(pools[rng.integers(0, cls + 1)] if cls else pools[0]))
        if rng.random() < 0.5: m, n = n, m
        k = int(rng.choice([1, 5, 16, 32, 48, 100, 128, 256, 512, 784]))
        if m * n * k > 3e9: k = 64
        ta, tb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        pad_a, pad_b, pad_c = (int(rng.choice([0, 0, 4, 8, 3])) for _ in range(3))
        ar, ac = (k, m) if ta else (m, k); br, bc = (n, k) if tb else (k, n)
        A = uniform(1000 + case, (ar, ac + pad_a), dtype=np.float32); B = uniform(2000 + case, (br, bc + pad_b), dtype=np.float32)
        Cbuf = uniform(3000 + case, (m, n + pad_c), dtype=np.float32)
        dA, dB, dC = dev.to_device(A), dev.to_device(B), dev.to_device(Cbuf)
        kw = {}
        a_eff = A[:, :ac]; b_eff = B[:, :bc]
        opA = (a_eff.T if ta else a_eff).astype(np.float64); opB = (b_eff.T if tb else b_eff).astype(np.float64)
        ref = opA @ opB
        bound = np.abs(opA) @ np.abs(opB)
        mode = rng.integers(0, 3)
        alpha = 1.0
        if mode == 1:      # alpha / beta
            alpha, beta = 0.5, 2.0
            kw = dict(alpha=alpha, beta=beta)
            ref = alpha * ref + beta * Cbuf[:, :n].astype(np.float64); bound = alpha * bound + np.abs(beta * Cbuf[:, :n])
        elif mode == 2:    # bias + relu
            bias = uniform(4000 + case, (m,), dtype=np.float32); dbias = dev.to_device(bias)
            kw = dict(bias_row=dbias, act=dev.native.ACT_RELU)
            ref = np.maximum(ref + bias.astype(np.float64)[:, None], 0); bound = bound + np.abs(bias)[:, None]
        dev.gemm(dA, dB, dC, transa=ta, transb=tb, m=m, n=n, k=k, lda=ac + pad_a, ldb=bc + pad_b, ldc=n + pad_c, **kw)
        name = dev.lib().bla_gemm_last_kernel().decode()
        seen.add(name.split("_")[2])
        got = dC.numpy()
        assert np.all(np.abs(got[:, :n] - ref) <= 1e-5 * bound + 1e-6), (case, m, n, k, ta, tb, mode, name)
        if pad_c:
            assert np.array_equal(got[:, n:], Cbuf[:, n:]), ("wrote outside the view", case, name)
    assert ("wsk32x32" in seen or "wsk16x16" in seen) and len(seen) >= 4, seen
    # the 256x256 kernel needs a chip-filling product on whole tiles with a plain epilogue: all four layouts, padded pitches, alpha
    for ta, tb in [(0, 0), (0, 1), (1, 0), (1, 1)]:
        m, n, k = 4096, 4096, 80
        ar, ac = (k, m) if ta else (m, k); br, bc = (n, k) if tb else (k, n)
        A = uniform(50 + ta, (ar, ac + 8), dtype=np.float32); B = uniform(60 + tb, (br, bc + 4), dtype=np.float32)
        dA, dB, dC = dev.to_device(A), dev.to_device(B), dev.zeros((m, n + 12))
        dev.gemm(dA, dB, dC, transa=bool(ta), transb=bool(tb), m=m, n=n, k=k, lda=ac + 8, ldb=bc + 4, ldc=n + 12, alpha=0.25)
        assert "glds256x256x16" in dev.lib().bla_gemm_last_kernel().decode()
        opA = (A[:, :ac].T if ta else A[:, :ac]).astype(np.float64); opB = (B[:, :bc].T if tb else B[:, :bc]).astype(np.float64)
        got = dC.numpy()
        assert np.all(np.abs(got[:, :n] - 0.25 * (opA @ opB)) <= 1e-5 * (np.abs(opA) @ np.abs(opB))), (ta, tb)
        assert not got[:, n:].any()
    # ... and the 128x512 tile: 128 rows, exactly one tile per CU
    m, n, k = 128, 512 * 256, 48
    A = uniform(70, (m, k), dtype=np.float32); B = uniform(71, (k, n), dtype=np.float32)
    dA, dB, dC = dev.to_device(A), dev.to_device(B), dev.zeros((m, n))
    dev.gemm(dA, dB, dC)
    assert "glds128x512x16" in dev.lib().bla_gemm_last_kernel().decode()
    ref = A.astype(np.float64) @ B.astype(np.float64)
    assert np.all(np.abs(dC.numpy() - ref) <= 1e-5 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)))

"""lib/layer.h on the device, batched (bla_layer_net_*): with one column it is lib/layer.c step by step -- checked against what the reference's
own layer.c computed (tests/golden/layer.npz: main.c:52-87 with activation x0.1, and a 12-7-5-3 net with a leaky ReLU); with several columns
the weight / bias steps are the sums of the single-column steps taken at the same starting weights (oracle.layer_net per column)."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden
from inputs import uniform

pytestmark = pytest.mark.gpu
F32 = np.float32


def make(pkg, sizes, batch, act, p):
    L = pkg.lib(); chk = pkg.native.check
    h = C.c_void_p()
    n = len(sizes) - 1
    chk(L.bla_layer_net_create(C.byref(h), (C.c_int * len(sizes))(*sizes), len(sizes), batch, (C.c_int * n)(*([act] * n)), (C.c_float * n)(*([p] * n))))
    return h


def put(pkg, ptr, a):
    a = np.ascontiguousarray(a, F32)
    pkg.native.check(pkg.lib().bla_memcpy_h2d(ptr, a.ctypes.data, a.nbytes, None)); pkg.sync()


def get(pkg, ptr, shape):
    out = np.empty(shape, F32)
    pkg.native.check(pkg.lib().bla_memcpy_d2h(out.ctypes.data, ptr, out.nbytes, None)); pkg.sync()
    return out


@pytest.mark.parametrize("case,n,act,p", [("main", 2, 1, 0.1), ("mlp", 3, 3, 0.25)])
def test_single_column_is_layer_c(pkg, case, n, act, p):
    pkg.init(0)
    L = pkg.lib(); chk = pkg.native.check
    g = golden("layer")
    ws = [g[f"{case}_w{i}"] for i in range(n)]; bs = [g[f"{case}_b{i}"] for i in range(n)]
    sizes = [ws[0].shape[1]] + [w.shape[0] for w in ws]
    h = make(pkg, sizes, 1, act, p)
    for i in range(n):
        put(pkg, L.bla_layer_net_weights(h, i + 1), ws[i]); put(pkg, L.bla_layer_net_biases(h, i + 1), bs[i])
    x = pkg.to_device(g[f"{case}_x"].astype(F32)); e = pkg.to_device(g[f"{case}_expect"].astype(F32).reshape(-1, 1))
    chk(L.bla_layer_net_forward_f32(h, None, x.ptr))
    close = lambda got, ref: (np.abs(got - ref) <= 1e-5 * np.abs(ref) + 1e-6 * np.abs(ref).max()).all()
    for i in range(n):
        assert close(get(pkg, L.bla_layer_net_nodes(h, i + 1), (sizes[i + 1], 1)), g[f"{case}_nodes{i}"]), i
        assert close(get(pkg, L.bla_layer_net_raw_nodes(h, i + 1), (sizes[i + 1], 1)), g[f"{case}_raw{i}"]), i
    chk(L.bla_layer_net_backward_f32(h, None, e.ptr, float(g[f"{case}_lr"])))
    for i in range(n):
        assert close(get(pkg, L.bla_layer_net_weights(h, i + 1), ws[i].shape), g[f"{case}_w{i}_new"]), i
        assert close(get(pkg, L.bla_layer_net_biases(h, i + 1), bs[i].shape), g[f"{case}_b{i}_new"]), i
    chk(L.bla_layer_net_destroy(h))


def test_batch_sums_the_single_column_steps(pkg, ora):
    pkg.init(0)
    L = pkg.lib(); chk = pkg.native.check
    sizes, B, lr = [20, 16, 12, 8], 24, 0.0625

    def act(v):
        v[v < 0] *= 0.25

    def ddx(v):
        v[:] = np.where(v > 0, 1.0, 0.25)
    ws = [uniform(9500 + i, (sizes[i + 1], sizes[i]), -0.5, 0.5, F32) for i in range(3)]
    bs = [uniform(9600 + i, (sizes[i + 1], 1), -0.2, 0.2, F32) for i in range(3)]
    x = uniform(9700, (sizes[0], B), -1, 1, F32); e = uniform(9701, (sizes[-1], B), 0, 1, F32)
    h = make(pkg, sizes, B, 3, 0.25)
    for i in range(3):
        put(pkg, L.bla_layer_net_weights(h, i + 1), ws[i]); put(pkg, L.bla_layer_net_biases(h, i + 1), bs[i])
    dx, de = pkg.to_device(x), pkg.to_device(e)      # the input has to outlive the backward pass (it is the first layer's a_prev)
    chk(L.bla_layer_net_forward_f32(h, None, dx.ptr))
    chk(L.bla_layer_net_backward_f32(h, None, de.ptr, lr))
    w64 = [w.astype(np.float64) for w in ws]; b64 = [b.astype(np.float64) for b in bs]
    dw = [np.zeros_like(w) for w in w64]; db = [np.zeros_like(b) for b in b64]
    for c in range(B):
        fwd, new = ora.layer_net(sizes, w64, b64, x[:, c].astype(np.float64), act, ddx, e[:, c], lr)
        for i in range(3):
            dw[i] += new[i][0] - w64[i]; db[i] += new[i][1] - b64[i]
        if c == 5:
            assert np.allclose(get(pkg, L.bla_layer_net_nodes(h, 3), (sizes[3], B))[:, c], fwd[2][0].ravel(), rtol=1e-5, atol=1e-6)
    for i in range(3):
        got_w = get(pkg, L.bla_layer_net_weights(h, i + 1), ws[i].shape); got_b = get(pkg, L.bla_layer_net_biases(h, i + 1), bs[i].shape)
        assert np.linalg.norm((got_w - ws[i]) - dw[i]) <= 1e-5 * np.linalg.norm(dw[i]), i
        assert np.linalg.norm((got_b - bs[i]) - db[i]) <= 1e-5 * np.linalg.norm(db[i]), i
    chk(L.bla_layer_net_destroy(h))


def test_wide_layers_take_the_separate_bias_pass(pkg):
    """Layers too large for the latency-bound GEMM (its epilogue carries the accumulated bias step): 512 -> 2048 -> 16 at B = 64 against numpy."""
    pkg.init(0)
    L = pkg.lib(); chk = pkg.native.check
    sizes, B, lr = [512, 2048, 16], 64, 0.01
    ws = [uniform(9800 + i, (sizes[i + 1], sizes[i]), -0.05, 0.05, F32) for i in range(2)]
    bs = [uniform(9810 + i, (sizes[i + 1], 1), -0.1, 0.1, F32) for i in range(2)]
    x = uniform(9820, (sizes[0], B), -1, 1, F32); e = uniform(9821, (sizes[-1], B), 0, 1, F32)
    h = make(pkg, sizes, B, 2, 0.0)        # ReLU
    for i in range(2):
        put(pkg, L.bla_layer_net_weights(h, i + 1), ws[i]); put(pkg, L.bla_layer_net_biases(h, i + 1), bs[i])
    dx, de = pkg.to_device(x), pkg.to_device(e)
    chk(L.bla_layer_net_forward_f32(h, None, dx.ptr)); chk(L.bla_layer_net_backward_f32(h, None, de.ptr, lr))
    w = [a.astype(np.float64) for a in ws]; b = [a.astype(np.float64) for a in bs]
    z1 = w[0] @ x + b[0]; a1 = np.maximum(z1, 0); z2 = w[1] @ a1 + b[1]; a2 = np.maximum(z2, 0)
    h2 = (z2 > 0) * 2 * (a2 - e); d2 = -lr * h2
    h1 = (z1 > 0) * (w[1].T @ h2); d1 = -lr * h1
    want = [(w[0] + d1 @ x.T, b[0] + d1.sum(1, keepdims=True)), (w[1] + d2 @ a1.T, b[1] + d2.sum(1, keepdims=True))]
    for i in range(2):
        gw = get(pkg, L.bla_layer_net_weights(h, i + 1), ws[i].shape); gb = get(pkg, L.bla_layer_net_biases(h, i + 1), bs[i].shape)
        assert np.linalg.norm((gw - ws[i]) - (want[i][0] - w[i])) <= 1e-4 * np.linalg.norm(want[i][0] - w[i]), i
        assert np.linalg.norm((gb - bs[i]) - (want[i][1] - b[i])) <= 1e-4 * np.linalg.norm(want[i][1] - b[i]), i
    chk(L.bla_layer_net_destroy(h))

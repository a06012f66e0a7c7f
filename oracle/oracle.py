"""ctypes/numpy front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- importable from tests/, bench.py's cpu_baseline leg
and __graft_entry__.smoke(); never from the product package.

Every function mirrors one reference function (file:line in oracle_impl.inc) on
flat row-major numpy arrays.  dtype float64 selects the ora64_* instantiation
(the reference's own matrix_float_t, lib/matrix.h:4), float32 the ora32_* one.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile liboracle*.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle_impl.inc", "Makefile")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so", "liboracle_O0.so"])
    if os.path.isdir("/root/reference/lib"):
        ref = os.path.join(_HERE, "_ref", "libref.so")
        if force or not os.path.exists(ref):
            subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def lib(opt="O2"):
    name = "liboracle.so" if opt == "O2" else "liboracle_O0.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        _LIBS[name] = C.CDLL(path)
    return _LIBS[name]


def _pfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "ora64_", C.c_double
    if dtype == np.float32:
        return "ora32_", C.c_float
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype=None):
    return np.ascontiguousarray(a, dtype=dtype)


def _call(name, dtype, *args, restype=None, opt="O2"):
    pfx, _ = _pfx(dtype)
    fn = getattr(lib(opt), pfx + name)
    fn.restype = restype
    return fn(*args)


# ---- lib/matrix.c ---------------------------------------------------------
def matmul(a, b, opt="O2"):
    a, b = _c(a), _c(b)
    assert a.dtype == b.dtype and a.shape[1] == b.shape[0]
    c = np.empty((a.shape[0], b.shape[1]), a.dtype)
    _call("matmul", a.dtype, _p(a), _p(b), _p(c), a.shape[0], a.shape[1], b.shape[1], opt=opt)
    return c


def matmul_rows(a, b, c, row0, row1, opt="O2"):
    _call("matmul_rows", a.dtype, _p(a), _p(b), _p(c), a.shape[1], b.shape[1], row0, row1, opt=opt)


def scale(m, f):
    m = _c(m).copy()
    _, ct = _pfx(m.dtype)
    _call("scale", m.dtype, _p(m), m.size, ct(f))
    return m


def add(a, b):
    a = _c(a).copy(); b = _c(b, a.dtype)
    _call("add", a.dtype, _p(a), _p(b), a.size)
    return a


def hadamard(a, b):
    a = _c(a).copy(); b = _c(b, a.dtype)
    _call("hadamard", a.dtype, _p(a), _p(b), a.size)
    return a


def transpose(m):
    m = _c(m)
    out = np.empty((m.shape[1], m.shape[0]), m.dtype)
    _call("transpose", m.dtype, _p(m), _p(out), m.shape[0], m.shape[1])
    return out


def row_sum(m):
    m = _c(m)
    out = np.empty((1, m.shape[1]), m.dtype)
    _call("row_sum", m.dtype, _p(m), _p(out), m.shape[0], m.shape[1])
    return out


def col_sum_as_written(m):
    """lib/matrix.c:138-148 literally; None where the reference reads out of bounds."""
    m = _c(m)
    out = np.empty((m.shape[0], 1), m.dtype)
    ok = _call("col_sum_as_written", m.dtype, _p(m), _p(out), m.shape[0], m.shape[1], restype=C.c_int)
    return out if ok else None


def col_sum_intended(m):
    m = _c(m)
    out = np.empty((m.shape[0], 1), m.dtype)
    _call("col_sum_intended", m.dtype, _p(m), _p(out), m.shape[0], m.shape[1])
    return out


def frobenius(m):
    m = _c(m)
    _, ct = _pfx(m.dtype)
    return _call("frobenius", m.dtype, _p(m), m.shape[0], m.shape[1], restype=ct)


def max_value(m):
    m = _c(m)
    _, ct = _pfx(m.dtype)
    return _call("max_value", m.dtype, _p(m), m.size, restype=ct)


def zscore(m):
    m = _c(m).copy()
    _call("zscore", m.dtype, _p(m), m.size)
    return m


def add_tile_columns(a, b):
    a = _c(a).copy(); b = _c(b, a.dtype)
    _call("add_tile_columns", a.dtype, _p(a), _p(b), a.shape[0], a.shape[1], b.shape[1])
    return a


def add_tile_rows(a, b):
    a = _c(a).copy(); b = _c(b, a.dtype)
    _call("add_tile_rows", a.dtype, _p(a), _p(b), a.shape[0], a.shape[1])
    return a


# ---- lib/util.c / model/mnist_nn.c ------------------------------------------
def relu(d):
    d = _c(d).copy()
    _call("relu", d.dtype, _p(d), d.size)
    return d


def relu_ddx(d):
    d = _c(d).copy()
    _call("relu_ddx", d.dtype, _p(d), d.size)
    return d


def softmax_cols(d):
    d = _c(d).copy()
    _call("softmax_cols", d.dtype, _p(d), d.shape[0], d.shape[1])
    return d


def softmax_rows(d):
    d = _c(d).copy()
    _call("softmax_rows", d.dtype, _p(d), d.shape[0], d.shape[1])
    return d


# ---- lib/conv.c -------------------------------------------------------------
def out_hw(h, w, s):
    """ceil((float)H/s) as lib/conv.c:55-56."""
    return int(np.ceil(np.float32(h) / s)), int(np.ceil(np.float32(w) / s))


def im2col(x, k, s):
    x = _c(x); c, h, w = x.shape
    ho, wo = out_hw(h, w, s)
    out = np.empty((ho * wo, k * k * c), x.dtype)
    _call("im2col", x.dtype, _p(x), _p(out), h, w, k, c, s)
    return out


def col2im(cols, c, h, w, k, s=1):
    cols = _c(cols)
    out = np.empty((c, h, w), cols.dtype)
    ok = _call("col2im", cols.dtype, _p(cols), _p(out), h, w, k, c, s, restype=C.c_int)
    return out if ok else None


def col2im_adjoint(cols, c, h, w, k, s):
    """The INTENDED _col2im for any stride: the adjoint of _im2col (lib/conv.c:58-74), out[c][i*s+p-pt][j*s+q-pl] += cols[(i,j)][c,p,q].
    The reference's own _col2im (lib/conv.c:80-135) walks the image grid instead of the output grid and is only defined for s == 1, where
    the two coincide; beyond that this restatement is pinned by <im2col(x), v> == <x, col2im_adjoint(v)> (tests).  numpy, small cases."""
    cols = np.asarray(cols)
    ho, wo = out_hw(h, w, s)
    vpad = max(int((np.ceil(np.float32(h) / s) - 1) * s + k - h), 0); hpad = max(int((np.ceil(np.float32(w) / s) - 1) * s + k - w), 0)
    pt, pl = vpad // 2, hpad // 2
    pad = np.zeros((c, h + vpad + k, w + hpad + k), cols.dtype)
    v = cols.reshape(ho, wo, c, k, k)
    for i in range(ho):
        for j in range(wo):
            pad[:, i * s:i * s + k, j * s:j * s + k] += v[i, j]
    return np.ascontiguousarray(pad[:, pt:pt + h, pl:pl + w])


def kernels_to_matrix(kern):
    kern = _c(kern); f, c, k, _ = kern.shape
    mat = np.empty((k * k * c, f), kern.dtype)
    _call("kernels_to_matrix", kern.dtype, _p(kern), _p(mat), f, c, k)
    return mat


def matrix_to_kernels(mat, c, k):
    mat = _c(mat); f = mat.shape[1]
    kern = np.empty((f, c, k, k), mat.dtype)
    _call("matrix_to_kernels", mat.dtype, _p(mat), _p(kern), f, c, k)
    return kern


def reshape_channels_matrix(matrix, h, w):
    """AS WRITTEN (lib/conv.c:174-187): matrix [HW][C] -> channels [C][H][W]."""
    matrix = _c(matrix); c = matrix.shape[1]
    ch = np.empty((c, h, w), matrix.dtype)
    _call("reshape_channels_matrix", matrix.dtype, _p(ch), _p(matrix), c, h * w)
    return ch


def reshape_matrix_channels(channels):
    """AS WRITTEN (lib/conv.c:190-203): channels [C][H][W] -> matrix [HW][C]."""
    channels = _c(channels); c, h, w = channels.shape
    m = np.empty((h * w, c), channels.dtype)
    _call("reshape_matrix_channels", channels.dtype, _p(m), _p(channels), c, h * w)
    return m


def conv_intended(x, kern, s):
    x, kern = _c(x), _c(kern)
    c, h, w = x.shape; f, _, k, _ = kern.shape
    ho, wo = out_hw(h, w, s)
    ws = dict(im2col=np.empty((ho * wo, k * k * c), x.dtype), kmat=np.empty((k * k * c, f), x.dtype),
              product=np.empty((ho * wo, f), x.dtype), output=np.empty((f, ho, wo), x.dtype))
    _call("conv_intended", x.dtype, _p(x), _p(kern), _p(ws["im2col"]), _p(ws["kmat"]), _p(ws["product"]),
          _p(ws["output"]), h, w, k, c, f, s)
    return ws


def conv_as_written(x, kern, s, stale_output):
    x, kern, stale_output = _c(x), _c(kern), _c(stale_output)
    c, h, w = x.shape; f, _, k, _ = kern.shape
    ho, wo = out_hw(h, w, s)
    ws = dict(im2col=np.empty((ho * wo, k * k * c), x.dtype), kmat=np.empty((k * k * c, f), x.dtype),
              product=np.empty((ho * wo, f), x.dtype), output=stale_output)
    _call("conv_as_written", x.dtype, _p(x), _p(kern), _p(ws["im2col"]), _p(ws["kmat"]), _p(ws["product"]),
          _p(stale_output), h, w, k, c, f, s)
    return ws


def conv_ddx_intended(del_y, im2col_m, kmat, c_in, k, s=1):
    del_y, im2col_m, kmat = _c(del_y), _c(im2col_m), _c(kmat)
    f, h, w = del_y.shape
    dt = del_y.dtype
    out = dict(del_q=np.empty((h * w, f), dt), del_kmat=np.empty((k * k * c_in, f), dt),
               del_kern=np.empty((f, c_in, k, k), dt), del_col=np.empty((h * w, k * k * c_in), dt),
               del_x=np.empty((c_in, h, w), dt))
    ok = _call("conv_ddx_intended", dt, _p(del_y), _p(im2col_m), _p(kmat), _p(out["del_q"]), _p(out["del_kmat"]),
               _p(out["del_kern"]), _p(out["del_col"]), _p(out["del_x"]), h, w, k, c_in, f, s, restype=C.c_int)
    return out if ok else None


# ---- lib/norm.c -------------------------------------------------------------
def group_norm(x, group_size):
    x = _c(x); c, h, w = x.shape
    ng = (c + group_size - 1) // group_size
    out = np.empty_like(x); sd = np.empty(ng, x.dtype); mu = np.empty(ng, x.dtype)
    _call("group_norm", x.dtype, _p(x), _p(out), _p(sd), _p(mu), c, group_size, h * w)
    return out, sd, mu


def group_norm_ddx(source, data, means, stdevs, group_size):
    source = _c(source); data = _c(data, source.dtype)
    means = _c(means, source.dtype); stdevs = _c(stdevs, source.dtype)
    c, h, w = source.shape
    dest = np.empty_like(source)
    _call("group_norm_ddx", source.dtype, _p(source), _p(dest), _p(data), _p(means), _p(stdevs), c, group_size, h * w)
    return dest


# ---- model/mnist_nn.c:218-315 -------------------------------------------------
def bucket_sizes(sizes):
    n0, n1, n2, n3 = sizes
    return [n1 * n0, n1, n2 * n1, n2, n3 * n2, n3]


def mnist_step(params, x_raw, y, colsum_intended=False):
    """One SGD step.  params = [w1,b1,w2,b2,w3,b3] (updated copies are returned).
    Returns (new_params, acts dict, grads list) or None where the as-written
    col_sum would be out of bounds."""
    dt = np.dtype(params[0].dtype)
    ps = [_c(p, dt).copy() for p in params]
    x_raw, y = _c(x_raw, dt), _c(y, dt)
    n1, n0 = ps[0].shape; n2 = ps[2].shape[0]; n3 = ps[4].shape[0]; B = x_raw.shape[1]
    acts = np.empty(2 * (n1 + n2 + n3) * B, dt)
    grads = np.empty(sum(bucket_sizes((n0, n1, n2, n3))), dt)
    ok = _call("mnist_step", dt, *[_p(p) for p in ps], _p(x_raw), _p(y), n0, n1, n2, n3, B,
               int(colsum_intended), _p(acts), _p(grads), restype=C.c_int)
    if not ok:
        return None
    names = ["z1", "a1", "z2", "a2", "z3", "a3"]
    rows = [n1, n1, n2, n2, n3, n3]
    ad, off = {}, 0
    for nm, r in zip(names, rows):
        ad[nm] = acts[off:off + r * B].reshape(r, B); off += r * B
    gl, off = [], 0
    for sz, p in zip(bucket_sizes((n0, n1, n2, n3)), ps):
        gl.append(grads[off:off + sz].reshape(p.shape)); off += sz
    return ps, ad, gl


# ---- lib/layer.c:6-107 (single-sample dense layers; small shapes: numpy over the restated products) ------------
def layer_net(sizes, weights, biases, x, act, act_ddx, expectations, learn_rate):
    """feed_forward on every non-input layer (lib/layer.c:6-20: raw = W.a_prev + b, nodes = act(raw)), then back_propagate_errors on the
    output layer (:80-107) with its recursion toward the input (:48-78).  act / act_ddx work in place on a vector like the reference's
    callbacks.  The reference passes learn_rate as a float and negates it as a float (:65,95).  Returns ([(nodes, raw)], [(W_new, b_new)])."""
    dt = np.dtype(weights[0].dtype)
    lr = dt.type(-np.float32(learn_rate))
    a = [_c(x, dt).reshape(-1, 1)]; raw = [None]
    for w, b in zip(weights, biases):
        z = matmul(_c(w, dt), a[-1]) + _c(b, dt).reshape(-1, 1)          # matrix_multiply, matrix_add (:10-11)
        raw.append(z.copy()); n = z.copy().ravel(); act(n); a.append(n.reshape(-1, 1))
    n_layers = len(weights)
    new = [None] * n_layers

    def step(l, cost_ddx_act):
        """layer index l (1-based): the shared tail of :64-77 / :92-106"""
        d = raw[l].copy().ravel(); act_ddx(d); d = d.reshape(-1, 1)
        db = hadamard(d, cost_ddx_act) * lr                               # matrix_multiply_elementwise, matrix_scale
        dw = matmul(db, transpose(a[l - 1]))                              # transpose, multiply, transpose back
        if l - 1 >= 1:                                                    # do_back_propagate_errors(previous, this, ...) with the OLD weights
            g = raw[l].copy().ravel(); act_ddx(g); g = hadamard(g.reshape(-1, 1), cost_ddx_act)
            step(l - 1, matmul(transpose(_c(weights[l - 1], dt)), g))
        new[l - 1] = (_c(weights[l - 1], dt) + dw, _c(biases[l - 1], dt).reshape(-1, 1) + db)
    e = _c(np.asarray(expectations, np.float32).astype(dt), dt).reshape(-1, 1)   # float* expectations (:80)
    step(n_layers, 2 * (a[-1] - e))
    return [(a[i], raw[i]) for i in range(1, n_layers + 1)], new


def print_matrix_text(m):
    """print_matrix + print_matrix_dim, lib/matrix.c:71-93, as the bytes the reference writes to stdout (C printf semantics: exact zero
    -> "0 ", anything below 0.01 -- negatives included -- "%.2e ", the rest "%.2f ")."""
    m = np.asarray(m)
    out = ["%d x %d matrix\n" % m.shape]
    flat = m.ravel()
    for i, v in enumerate(flat):
        v = float(v)
        if i % m.shape[1] == 0:
            out.append("[ ")
        out.append("0 " if v == 0 else ("%.2e " % v if v < 0.01 else "%.2f " % v))
        if (i + 1) % m.shape[1] == 0:
            out.append("]\n")
    out.append("\n")
    out.append("%d x %d matrix\n" % m.shape)
    return "".join(out).encode()


def mnist_metrics(a3, y):
    """model/mnist_nn.c:237-257 for one batch: (batch_loss, num_correct).  Prediction = first row whose probability exceeds every earlier
    one, starting from 0 (:241-247); correct when the one-hot matrix has a 1 there (:248); loss = cross_entropy_loss (:83-91, LOSS_EPSILON
    1e-15) over the flat chunks [10k, 10k+10) of both matrices as :252-254 takes them, added in k order."""
    a3 = np.ascontiguousarray(a3, np.float64); y = np.ascontiguousarray(y, np.float64)
    n3, B = a3.shape
    fa, fy = a3.ravel(), y.ravel()
    correct, loss = 0, 0.0
    for k in range(B):
        pred, best = 0, 0.0
        for p in range(n3):
            if a3[p, k] > best:
                best, pred = a3[p, k], p
        correct += int(fy[k + pred * B] == 1)
        chunk = 0.0
        for i in range(k * n3, k * n3 + n3):
            chunk += -1 * (fy[i] * np.log(fa[i] + 1e-15))
        loss += chunk
    return loss, correct


# ---- model/cifar_unet.c forward() :1099-1166 / backward() :1351-1436, composed from the restated blocks above -------------------
def conv_backward_any_stride(del_y, x, kern, s):
    """(del_kern, del_x) of conv(): conv_ddx (lib/conv.c:214-229) where the reference defines it (stride 1); for other strides the same chain
    with the adjoint _col2im (col2im_adjoint) -- what model/cifar_unet.c:1412,1420,1430 need for the stride-2 convolutions."""
    x = _c(x); kern = _c(kern, x.dtype); del_y = _c(del_y, x.dtype)
    cin, h, w = x.shape; f, _, k, _ = kern.shape
    fw = conv_intended(x, kern, s)
    if s == 1:
        dd = conv_ddx_intended(del_y, fw["im2col"], fw["kmat"], cin, k)
        return dd["del_kern"], dd["del_x"]
    dq = reshape_matrix_channels(del_y)
    del_kern = matrix_to_kernels(matmul(transpose(fw["im2col"]), dq), cin, k)
    return del_kern, col2im_adjoint(matmul(dq, transpose(fw["kmat"])), cin, h, w, k, s)


def unet(cfg, P, x, temb, noise, drop=None):
    """The intended wiring of the reference's U-Net (see big-linear-algebra_amd/csrc/bla_unet_model.hip for the four call sites where the
    reference's work-in-progress code differs).  cfg: dict(image_h, image_w, in_channels, dims[4], time_dim, kernel, group_size, key_dim);
    P: tensor name -> array, names as bla_unet_tensor_info gives them.  Returns (prediction, gradients by the same names)."""
    D, gs, k = cfg["dims"], cfg["group_size"], cfg["kernel"]
    dt = np.dtype(x.dtype)
    G = {}
    off = [0]

    def take_drop(n):
        d = np.zeros(n, np.uint8) if drop is None else drop[off[0]:off[0] + n]
        off[0] += n
        return d
    res_f, att_f = {}, {}

    def res(name, xin):
        cout = P[name + ".conv_1_kernels"].shape[0]
        kres = P.get(name + ".residual_conv_kernels")
        f = resnet_forward(xin, temb, P[name + ".conv_1_kernels"], P[name + ".conv_2_kernels"], P[name + ".time_weights"], P[name + ".time_biases"],
                           kres, take_drop(cout * xin.shape[1] * xin.shape[2]), gs)
        res_f[name] = (xin, f)
        return f["result"]

    def att(name, xin):
        f = attention_forward(xin, P[name + ".Q_proj"], P[name + ".K_proj"], P[name + ".V_proj"], P[name + ".weights"], P[name + ".biases"])
        att_f[name] = (xin, f)
        return f["out"]

    def res_b(name, g):
        xin, f = res_f[name]
        o = resnet_backward(g, xin, temb, P[name + ".conv_1_kernels"], P[name + ".conv_2_kernels"], P.get(name + ".residual_conv_kernels"), f, gs)
        G[name + ".conv_1_kernels"], G[name + ".conv_2_kernels"] = o["dk1"], o["dk2"]
        G[name + ".time_weights"], G[name + ".time_biases"] = o["dtw"], o["dtb"]
        if o["dkres"] is not None:
            G[name + ".residual_conv_kernels"] = o["dkres"]
        return o["del_x"]

    def att_b(name, g):
        xin, f = att_f[name]
        o = attention_backward(g, xin, P[name + ".Q_proj"], P[name + ".K_proj"], P[name + ".V_proj"], P[name + ".weights"], f)
        G[name + ".Q_proj"], G[name + ".K_proj"], G[name + ".V_proj"], G[name + ".weights"] = o["del_wq"], o["del_wk"], o["del_wv"], o["del_w"]
        G[name + ".biases"] = np.zeros_like(P[name + ".biases"])      # the reference computes no gradient for the attention bias
        return o["del_x"]

    def conv_b(name, g, xin, s):
        G[name], dx = conv_backward_any_stride(g, xin, P[name], s)
        return dx
    # ---- forward
    r11 = res("down_1_resnet_1", x); s1 = res("down_1_resnet_2", r11)
    c1 = conv_intended(s1, P["down_1_conv_kernels"], 2)["output"]
    r21 = res("down_2_resnet_1", c1); a21 = att("down_2_self_attention_1", r21); s2 = res("down_2_resnet_2", a21); a22 = att("down_2_self_attention_2", s2)
    c2 = conv_intended(a22, P["down_2_conv_kernels"], 2)["output"]
    r31 = res("down_3_resnet_1", c2); s3 = res("down_3_resnet_2", r31)
    c3 = conv_intended(s3, P["down_3_conv_kernels"], 2)["output"]
    r41 = res("down_4_resnet_1", c3); s4 = res("down_4_resnet_2", r41)
    m1 = res("mid_resnet_1", s4); ma = att("mid_self_attention", m1); m2 = res("mid_resnet_2", ma)
    ups = {}

    def upsample(i, t, target):
        nn = nearest_neighbours(t, target.shape[1], target.shape[2], 2)
        name = f"up_{i}_conv_kernels"
        ups[i] = (t, nn)
        return conv_intended(nn, P[name], 1)["output"] if name in P else nn
    cat1 = np.concatenate([m2, s4]); u11 = res("up_1_resnet_1", cat1); u12 = res("up_1_resnet_2", u11)
    cat2 = np.concatenate([upsample(1, u12, s3), s3]); u21 = res("up_2_resnet_1", cat2); u22 = res("up_2_resnet_2", u21)
    cat3 = np.concatenate([upsample(2, u22, s2), s2]); u31 = res("up_3_resnet_1", cat3); a31 = att("up_3_self_attention_1", u31)
    u32 = res("up_3_resnet_2", a31); a32 = att("up_3_self_attention_2", u32)
    cat4 = np.concatenate([upsample(3, a32, s1), s1]); u41 = res("up_4_resnet_1", cat4); u42 = res("up_4_resnet_2", u41)
    gn, sd, mu = group_norm(u42, gs); orelu = relu(gn)
    out = conv_intended(orelu, P["output_conv_kernels"], 1)["output"]
    # ---- backward
    g = 2 * (out - _c(noise, dt))
    g = conv_b("output_conv_kernels", g, orelu, 1)
    g = group_norm_ddx(relu_mask(g, orelu), u42, mu, sd, gs)

    def upsample_b(i, g):
        t, nn = ups[i]
        name = f"up_{i}_conv_kernels"
        if name in P:
            g = conv_b(name, g, nn, 1)
        return nearest_neighbours_ddx(g, t.shape[1], t.shape[2], 2)
    g = res_b("up_4_resnet_2", g); g = res_b("up_4_resnet_1", g); n = D[0]; gs1 = g[n:]; g = upsample_b(3, g[:n])
    g = att_b("up_3_self_attention_2", g); g = res_b("up_3_resnet_2", g); g = att_b("up_3_self_attention_1", g); g = res_b("up_3_resnet_1", g)
    n = D[1]; gs2 = g[n:]; g = upsample_b(2, g[:n])
    g = res_b("up_2_resnet_2", g); g = res_b("up_2_resnet_1", g); n = D[2]; gs3 = g[n:]; g = upsample_b(1, g[:n])
    g = res_b("up_1_resnet_2", g); g = res_b("up_1_resnet_1", g); n = D[3]; gs4 = g[n:]; g = g[:n]
    g = res_b("mid_resnet_2", g); g = att_b("mid_self_attention", g); g = res_b("mid_resnet_1", g)
    g = res_b("down_4_resnet_2", g + gs4); g = res_b("down_4_resnet_1", g)
    g = conv_b("down_3_conv_kernels", g, s3, 2) + gs3
    g = res_b("down_3_resnet_2", g); g = res_b("down_3_resnet_1", g)
    g = conv_b("down_2_conv_kernels", g, a22, 2)
    g = att_b("down_2_self_attention_2", g) + gs2
    g = res_b("down_2_resnet_2", g); g = att_b("down_2_self_attention_1", g); g = res_b("down_2_resnet_1", g)
    g = conv_b("down_1_conv_kernels", g, s1, 2) + gs1
    g = res_b("down_1_resnet_2", g); g = res_b("down_1_resnet_1", g)
    return out, G


# ---- error-bound helpers ------------------------------------------------------
def matmul_f32_acc64(a, b):
    a, b = _c(a, np.float32), _c(b, np.float32)
    c = np.empty((a.shape[0], b.shape[1]), np.float64)
    lib().ora_matmul_f32_acc64(_p(a), _p(b), _p(c), a.shape[0], a.shape[1], b.shape[1])
    return c


def matmul_abs_f32(a, b):
    a, b = _c(a, np.float32), _c(b, np.float32)
    c = np.empty((a.shape[0], b.shape[1]), np.float64)
    lib().ora_matmul_abs_f32(_p(a), _p(b), _p(c), a.shape[0], a.shape[1], b.shape[1])
    return c


# ---- model/cifar_unet.c glue ops --------------------------------------------------------------------------
def relu_mask(source, relu_result):
    source = _c(source); relu_result = _c(relu_result, source.dtype)
    dest = np.empty_like(source)
    _call("relu_mask", source.dtype, _p(dest), _p(source), _p(relu_result), source.size)
    return dest


def add_channel_bias(x, t):
    x = _c(x).copy(); t = _c(t, x.dtype)
    _call("add_channel_bias", x.dtype, _p(x), _p(t), x.shape[0], int(np.prod(x.shape[1:])))
    return x


def dropout_apply(x, drop):
    x = _c(x); drop = _c(drop, np.uint8)
    y = np.empty_like(x)
    _call("dropout_apply", x.dtype, _p(x), _p(y), _p(drop), x.size)
    return y


def dropout_mask(x, dropout_result):
    x = _c(x).copy(); dropout_result = _c(dropout_result, x.dtype)
    _call("dropout_mask", x.dtype, _p(x), _p(dropout_result), x.size)
    return x


def nearest_neighbours(x, out_h, out_w, scale):
    x = _c(x); c, h, w = x.shape
    out = np.empty((c, out_h, out_w), x.dtype)
    _call("nearest_neighbours", x.dtype, _p(x), _p(out), c, w, h * w, out_h, out_w, scale)
    return out


def nearest_neighbours_ddx(source, dh, dw, scale):
    source = _c(source); c, sh, sw = source.shape
    dest = np.empty((c, dh, dw), source.dtype)
    _call("nearest_neighbours_ddx", source.dtype, _p(source), _p(dest), c, sh, sw, dh, dw, scale)
    return dest


def softmax_ddx(softmax_output, gradient):
    s = _c(softmax_output); g = _c(gradient, s.dtype)
    out = np.empty_like(s)
    _call("softmax_ddx", s.dtype, _p(s), _p(g), _p(out), s.shape[0], s.shape[1])
    return out


# ---- self-attention block (model/cifar_unet.c:999-1022, 1261-1337), intended composition -------------------
def attention_forward(x, wq, wk, wv, w, bias):
    x = _c(x); dt = x.dtype
    wq, wk, wv, w, bias = [_c(a, dt) for a in (wq, wk, wv, w, bias)]
    c = x.shape[0]; s = int(np.prod(x.shape[1:])); d = wq.shape[1]
    o = dict(q=np.empty((s, d), dt), k=np.empty((s, d), dt), v=np.empty((s, d), dt), raw=np.empty((s, s), dt),
             wts=np.empty((s, s), dt), att=np.empty((s, d), dt), out=np.empty(x.shape, dt))
    _call("attention_forward", dt, _p(x), _p(wq), _p(wk), _p(wv), _p(w), _p(bias), _p(o["q"]), _p(o["k"]), _p(o["v"]), _p(o["raw"]),
          _p(o["wts"]), _p(o["att"]), _p(o["out"]), c, s, d)
    return o


def attention_backward(del_y, x, wq, wk, wv, w, fwd, jacobian_from_raw=False):
    del_y = _c(del_y); dt = del_y.dtype
    x, wq, wk, wv, w = [_c(a, dt) for a in (x, wq, wk, wv, w)]
    c = x.shape[0]; s = int(np.prod(x.shape[1:])); d = wq.shape[1]
    o = dict(del_wq=np.empty((c, d), dt), del_wk=np.empty((c, d), dt), del_wv=np.empty((c, d), dt), del_w=np.empty((d, c), dt),
             del_x=np.empty(x.shape, dt))
    f = {n: _c(fwd[n], dt) for n in ("q", "k", "v", "raw", "wts", "att")}
    _call("attention_backward", dt, _p(del_y), _p(x), _p(wq), _p(wk), _p(wv), _p(w), _p(f["q"]), _p(f["k"]), _p(f["v"]), _p(f["raw"]),
          _p(f["wts"]), _p(f["att"]), _p(o["del_wq"]), _p(o["del_wk"]), _p(o["del_wv"]), _p(o["del_w"]), _p(o["del_x"]), c, s, d,
          int(jacobian_from_raw))
    return o


# ---- ResNet block (model/cifar_unet.c:1044-1072, 1180-1227), intended composition ---------------------------
def resnet_forward(x, temb, k1, k2, tw, tb, kres, drop, group_size):
    x = _c(x); dt = x.dtype
    temb, k1, k2, tw, tb = [_c(a, dt) for a in (temb, k1, k2, tw, tb)]
    cin, h, w = x.shape; cout, _, k, _ = k1.shape; tdim = tw.shape[0]; hw = h * w
    kres = _c(kres, dt) if kres is not None else None
    drop = _c(drop, np.uint8)
    g1 = (cin + group_size - 1) // group_size; g2 = (cout + group_size - 1) // group_size
    o = dict(mu1=np.empty(g1, dt), sd1=np.empty(g1, dt), relu1=np.empty((cin, h, w), dt), c1=np.empty((cout, h, w), dt), tdense=np.empty(cout, dt),
             mu2=np.empty(g2, dt), sd2=np.empty(g2, dt), relu2=np.empty((cout, h, w), dt), dp=np.empty((cout, h, w), dt), c2=np.empty((cout, h, w), dt),
             res=np.zeros((cout, h, w), dt), result=np.empty((cout, h, w), dt))
    _call("resnet_forward", dt, _p(x), _p(temb), _p(k1), _p(k2), _p(tw), _p(tb), _p(kres) if kres is not None else None, _p(drop),
          *[_p(o[n]) for n in ("mu1", "sd1", "relu1", "c1", "tdense", "mu2", "sd2", "relu2", "dp", "c2", "res", "result")],
          h, w, cin, cout, k, tdim, group_size)
    return o


def resnet_backward(del_out, x, temb, k1, k2, kres, fwd, group_size):
    del_out = _c(del_out); dt = del_out.dtype
    x, temb, k1, k2 = [_c(a, dt) for a in (x, temb, k1, k2)]
    cin, h, w = x.shape; cout, _, k, _ = k1.shape; tdim = temb.size
    kres = _c(kres, dt) if kres is not None else None
    o = dict(dk1=np.empty_like(k1), dk2=np.empty_like(k2), dtw=np.empty((tdim, cout), dt), dtb=np.empty(cout, dt),
             dkres=np.empty((cout, cin, 1, 1), dt) if kres is not None else None, del_x=np.empty_like(x))
    f = {n: _c(fwd[n], dt) for n in ("mu1", "sd1", "relu1", "c1", "mu2", "sd2", "relu2", "dp")}
    _call("resnet_backward", dt, _p(del_out), _p(x), _p(temb), _p(k1), _p(k2), _p(kres) if kres is not None else None,
          *[_p(f[n]) for n in ("mu1", "sd1", "relu1", "c1", "mu2", "sd2", "relu2", "dp")],
          _p(o["dk1"]), _p(o["dk2"]), _p(o["dtw"]), _p(o["dtb"]), _p(o["dkres"]) if kres is not None else None, _p(o["del_x"]),
          h, w, cin, cout, k, tdim, group_size)
    return o

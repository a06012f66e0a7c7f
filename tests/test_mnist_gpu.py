"""GPU parity of the device-resident MNIST-NN trainer (bla_mnist_nn_*, the hot loop of
model/mnist_nn.c:218-315) against the golden step the REFERENCE produced from its own trained weights
(tests/golden/mnist_step.npz, mnist_nn_params.npz), plus sharding identities for data parallelism.

Tolerance: fp32 device vs fp64 reference, elementwise |got - ref| <= 1e-5 * (|ref| + mean|ref|)
(north_star: forward/backward outputs within 1e-5 relative of the CPU reference)."""
import os

import numpy as np
import pytest

from conftest import ROOT, golden
from inputs import uniform, randint

pytestmark = pytest.mark.gpu
PN = ["w1", "b1", "w2", "b2", "w3", "b3"]
GN = ["dw1", "db1", "dw2", "db2", "dw3", "db3"]
RTOL = 1e-5


@pytest.fixture(scope="module")
def dev(pkg):
    pkg.init(0)
    return pkg


def real_params():
    z = np.load(os.path.join(ROOT, "tests", "golden", "mnist_nn_params.npz"))
    return [z[n] for n in PN]


def batch(B, n_in=784, n_out=10, seeds=None):
    s1, s2 = seeds if seeds else (3000 + B, 3100 + B)
    x_raw = randint(s1, (n_in, B), 256).astype(np.float32)
    lab = randint(s2, (B,), n_out)
    y = np.zeros((n_out, B), np.float32); y[lab, np.arange(B)] = 1
    return x_raw, y


def check(g, name, got):
    g.check(name, got, rtol=RTOL, atol=RTOL * g.mean_abs(name))


def run_and_check(dev, g, tag, params, x_raw, y, mode, sizes):
    nn = dev.mnist_nn.MnistNN(x_raw.shape[1], sizes, mode)
    nn.set_params(params)
    nn.load_batch(x_raw, y)
    nn.train_step()
    for n in ["z1", "a1", "z2", "a2", "z3", "a3"]:
        check(g, f"{tag}_{n}", nn.activation(n))
    for n, v in zip(GN, nn.grads()):
        check(g, f"{tag}_{n}", v)
    for n, v in zip(PN, nn.get_params()):
        g.check(f"{tag}_new_{n}", v, rtol=RTOL, atol=1e-7)   # parameters move by lr*grad ~ 1e-5..1e-3: absolute floor = fp32 eps of |w|
    return nn


def test_golden_step_real_weights(dev):
    g = golden("mnist_step")
    for tag, B, mode in [("b256_aswritten", 256, 0), ("b64_intended", 64, 1), ("b256_intended", 256, 1)]:
        x_raw, y = batch(B)
        run_and_check(dev, g, tag, real_params(), x_raw, y, mode, (784, 256, 128, 10))


def test_golden_step_tiny(dev):
    g = golden("mnist_step")
    tp = [uniform(3200, (8, 12), dtype=np.float32), uniform(3201, (8, 1), dtype=np.float32), uniform(3202, (6, 8), dtype=np.float32),
          uniform(3203, (6, 1), dtype=np.float32), uniform(3204, (4, 6), dtype=np.float32), uniform(3205, (4, 1), dtype=np.float32)]
    x_raw, y = batch(16, 12, 4, (3206, 3207))
    for tag, mode in [("tiny_aswritten", 0), ("tiny_intended", 1)]:
        run_and_check(dev, g, tag, tp, x_raw, y, mode, (12, 8, 6, 4))


def test_as_written_colsum_refused_where_reference_is_undefined(dev):
    nn = dev.mnist_nn.MnistNN(64, colsum_mode=0)           # 256 x 64: rows > cols -> heap over-read in the reference (Q2)
    nn.set_params(real_params()); nn.load_batch(*batch(64))
    with pytest.raises(dev.BlaError) as e:
        nn.train_step()
    assert e.value.status == 5


def test_graph_replay_matches_eager(dev):
    """Replaying the captured forward/backward + separate update is bit-identical to the eager step (same kernels, same order).
    The one-graph step with the update folded into the weight-gradient products (W += lr * dZ.A^T through alpha / beta, b through the
    scaled row sum; 6 launches) differs only by the rounding of that fused multiply-add."""
    x_raw, y = batch(256)
    a = dev.mnist_nn.MnistNN(256); b = dev.mnist_nn.MnistNN(256); c = dev.mnist_nn.MnistNN(256); d = dev.mnist_nn.MnistNN(256)
    for nn in (a, b, c, d):
        nn.set_params(real_params()); nn.load_batch(x_raw, y)
    p0 = real_params()
    for _ in range(3):
        a.train_step()
        b.graph_step(with_update=False); b.apply()
        c.graph_step()
        d.fused_step()                                       # the same six launches issued directly
    dev.sync()
    for pc, pd in zip(c.get_params(), d.get_params()):
        assert np.array_equal(pc, pd)
    for pa, pb, pc, q in zip(a.get_params(), b.get_params(), c.get_params(), p0):
        assert np.array_equal(pa, pb)
        assert np.linalg.norm(pc - pa) <= 1e-6 * np.linalg.norm(pa) + 1e-9
        assert np.linalg.norm((pc - q) - (pa - q)) <= 1e-5 * np.linalg.norm(pa - q) + 1e-9      # the update itself


def test_ten_steps_track_the_oracle(dev, ora):
    """10 consecutive steps (parameters feed back): fp32 device vs fp64 oracle, normwise 1e-5."""
    params = real_params(); p64 = [p.astype(np.float64) for p in params]
    nn = dev.mnist_nn.MnistNN(256); nn.set_params(params)
    for step in range(10):
        x_raw, y = batch(256, seeds=(4000 + step, 4100 + step))
        nn.load_batch(x_raw, y); nn.graph_step()
        p64, _, _ = ora.mnist_step(p64, x_raw.astype(np.float64), y.astype(np.float64), colsum_intended=True)
    for got, ref in zip(nn.get_params(), p64):
        assert np.linalg.norm(got - ref) <= RTOL * np.linalg.norm(ref) + 1e-7


@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_column_shards_sum_to_the_full_batch_gradient(dev, ranks):
    """Data-parallel identity on one GPU: R replicas on B/R columns each; the SUM of their gradient buckets equals
    the single-replica gradient at B (intended col_sum; the as-written one is not column-separable)."""
    B = 256
    x_raw, y = batch(B)
    full = dev.mnist_nn.MnistNN(B); full.set_params(real_params()); full.load_batch(x_raw, y); full.forward_backward()
    want = dev.mnist_nn.flatten_params(full.grads())
    acc = np.zeros_like(want, dtype=np.float64)
    for r in range(ranks):
        lo, hi = dev.mnist_nn.shard_columns(B, ranks, r)
        nn = dev.mnist_nn.MnistNN(hi - lo); nn.set_params(real_params())
        nn.load_batch(np.ascontiguousarray(x_raw[:, lo:hi]), np.ascontiguousarray(y[:, lo:hi]))
        nn.forward_backward()
        acc += dev.mnist_nn.flatten_params(nn.grads())
    assert np.linalg.norm(acc - want) <= 1e-6 * np.linalg.norm(want)       # fp32 summation-order tolerance (SURVEY 8d cfg 4)


def test_rccl_single_rank_with_torch_buckets(dev):
    """The bench path: buckets are torch tensors (so torch.distributed/RCCL can all-reduce them), kernels are ours,
    both in one process on one HIP runtime.  World size 1 here (one GPU); the 8-GPU run is the driver's."""
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x_raw, y = batch(256)
        ref = dev.mnist_nn.MnistNN(256); ref.set_params(real_params()); ref.load_batch(x_raw, y); ref.train_step()
        nn = dev.mnist_nn.MnistNN(256); nn.set_params(real_params()); nn.load_batch(x_raw, y)
        params_t = torch.zeros(nn.count, device="cuda", dtype=torch.float32)
        grads_t = torch.zeros(nn.count, device="cuda", dtype=torch.float32)
        torch.cuda.synchronize()
        nn.use_buckets(params_t.data_ptr(), grads_t.data_ptr())
        ts = torch.cuda.Stream()             # explicit stream: a NULL handle would mean "the library's own stream" to bla
        torch.cuda.set_stream(ts)
        stream = ts.cuda_stream
        dev.mnist_nn.data_parallel_step(lambda: nn.forward_backward(stream), grads_t, lambda: nn.apply(stream=stream), dist)
        torch.cuda.synchronize()
        got = dev.mnist_nn.split_bucket(params_t.cpu().numpy())
        for a, b in zip(got, ref.get_params()):
            assert np.array_equal(a, b)
    finally:
        dist.destroy_process_group()

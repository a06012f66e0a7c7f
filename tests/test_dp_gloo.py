"""The N > 1 data-parallel path on CPU: world_size-2 gloo processes run the package's own host logic
(shard_columns + data_parallel_step: local gradients -> SUM all-reduce of the flat bucket -> identical update)
with the oracle standing in for the device kernels, and must reproduce the single-process full-batch step."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_pkg

PN = ["w1", "b1", "w2", "b2", "w3", "b3"]


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]
    import torch
    import torch.distributed as dist
    import oracle
    from inputs import randint
    mn = load_pkg().mnist_nn
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(ROOT, "tests", "golden", "mnist_nn_params.npz"))
    params = [z[n].astype(np.float64) for n in PN]
    B = 64
    for step in range(2):
        x_raw = randint(5000 + step, (784, B), 256).astype(np.float64)
        lab = randint(5100 + step, (B,), 10); y = np.zeros((10, B)); y[lab, np.arange(B)] = 1
        lo, hi = mn.shard_columns(B, world, rank)
        grads_t = torch.zeros(sum(p.size for p in params), dtype=torch.float64)

        def local():
            _, _, g = oracle.mnist_step(params, np.ascontiguousarray(x_raw[:, lo:hi]), np.ascontiguousarray(y[:, lo:hi]), colsum_intended=True)
            grads_t.copy_(torch.from_numpy(np.concatenate([a.ravel() for a in g])))

        def apply():
            flat = grads_t.numpy(); off = 0
            for p in params:
                p += mn.LEARN_RATE * flat[off:off + p.size].reshape(p.shape); off += p.size
        mn.data_parallel_step(local, grads_t, apply, dist)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), *params)
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path, ora):
    import torch.multiprocessing as mp
    from inputs import randint
    port = 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z = np.load(os.path.join(ROOT, "tests", "golden", "mnist_nn_params.npz"))
    params = [z[n].astype(np.float64) for n in PN]
    for step in range(2):
        x_raw = randint(5000 + step, (784, 64), 256).astype(np.float64)
        lab = randint(5100 + step, (64,), 10); y = np.zeros((10, 64)); y[lab, np.arange(64)] = 1
        params, _, _ = ora.mnist_step(params, x_raw, y, colsum_intended=True)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    for i, ref in enumerate(params):
        a, b = r0[f"arr_{i}"], r1[f"arr_{i}"]
        assert np.array_equal(a, b)                                       # replicas stay bit-identical
        assert np.linalg.norm(a - ref) <= 1e-12 * np.linalg.norm(ref)     # == full-batch step up to fp64 summation order


def test_shard_columns():
    mn = load_pkg().mnist_nn
    assert [mn.shard_columns(2048, 8, r) for r in (0, 7)] == [(0, 256), (1792, 2048)]
    with pytest.raises(AssertionError):
        mn.shard_columns(100, 8, 0)

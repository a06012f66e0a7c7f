#!/usr/bin/env python3
"""One rank of the multi-process exchange test (tests/test_dp_exchange_gpu.py starts `world` of these on the
one GPU of the box: an IPC mapping of another process's buffer on the same device takes the same code path as a
peer GPU's, minus the xGMI hop).  usage: dp_worker.py RANK WORLD DIR STEPS PER_RANK_BATCH"""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_pkg
from inputs import uniform, randint

rank, world, d, steps, per = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
bla = load_pkg(); bla.init(0)
L = bla.lib(); mn = bla.mnist_nn; native = bla.native


def file_all_gather(tag):
    """host-side channel for the 64-byte handles: write own, wait for everyone's"""
    def gather(b):
        tmp = os.path.join(d, f"{tag}{rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(b)
        os.rename(tmp, os.path.join(d, f"{tag}{rank}.bin"))
        out, t0 = [], time.time()
        for r in range(world):
            p = os.path.join(d, f"{tag}{r}.bin")
            while not os.path.exists(p):
                if time.time() - t0 > 90:
                    raise TimeoutError(p)
                time.sleep(0.01)
            with open(p, "rb") as f:
                out.append(f.read())
        return out
    return gather


res = {}
# --- A: raw exchange, ragged count, `out` and fused `target`, three rounds alternating the parity ---------------
count = 10007
ex = mn.Exchange(rank, world, count, file_all_gather("a"))
out = bla.empty((count,)); tgt = bla.to_device(np.full(count, 1.0, np.float32))
for rnd in range(3):
    g = uniform(1000 * rnd + rank, (count,), -1, 1, np.float32)
    native.check(L.bla_memcpy_h2d(ex.bucket(rnd & 1), g.ctypes.data, g.nbytes, None)); native.sync()
    ex.allreduce(rnd & 1, out=out.ptr, target=tgt.ptr, alpha=0.5)
    native.sync()
    res[f"sum{rnd}"] = out.numpy().copy()
res["target"] = tgt.numpy()
res["status_a"] = np.int32(ex.status())

# --- B: data-parallel MNIST-NN steps (graph launch / direct launches) ----------------------------------------------------------
gB = per * world
nn = mn.MnistNN(per, colsum_mode=mn.COLSUM_INTENDED)
z = np.load(os.path.join(ROOT, "tests", "golden", "mnist_nn_params.npz"))
nn.set_params([z[n] for n in ["w1", "b1", "w2", "b2", "w3", "b3"]])
x_raw = randint(7, (784, gB), 256).astype(np.float32)
lab = randint(8, (gB,), 10); y = np.zeros((10, gB), np.float32); y[lab, np.arange(gB)] = 1
lo, hi = mn.shard_columns(gB, world, rank)
nn.load_batch(np.ascontiguousarray(x_raw[:, lo:hi]), np.ascontiguousarray(y[:, lo:hi]))
ex2 = mn.Exchange(rank, world, nn.count, file_all_gather("b"))
t0 = time.perf_counter()
for i in range(steps):
    nn.dp_step(ex2, graph=(i % 3 != 2))      # graph replays and direct launches mixed: they share the bucket parity
native.sync()
res["wall_per_step_us"] = np.float64((time.perf_counter() - t0) / steps * 1e6)
res["params"] = mn.flatten_params(nn.get_params())
res["status_b"] = np.int32(ex2.status())
np.savez(os.path.join(d, f"result{rank}.npz"), **res)
# keep the mappings alive until every rank has finished reading its peers
file_all_gather("done")(b"x" * 64)
ex2.close(); ex.close()
print(f"rank {rank} ok", flush=True)

#!/usr/bin/env python3
"""Ranks [LO, HI) of a `WORLD`-rank data-parallel group, hosted by THIS process on the one GPU of the box
(tests/test_dp_exchange_gpu.py starts one or several of these).  Each rank has its own bla context (stream, workspace);
ranks of other processes are reached through IPC mappings, ranks of this process directly -- on a one-GPU box either takes the
same kernel code path as a peer GPU's memory, minus the xGMI hop.
usage: dp_worker.py LO HI WORLD DIR STEPS PER_RANK_BATCH"""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_pkg
from inputs import uniform, randint

lo_r, hi_r, world, d, steps, per = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
mine = list(range(lo_r, hi_r))
bla = load_pkg(); bla.init(0)
L = bla.lib(); mn = bla.mnist_nn; native = bla.native
ctxs = {r: mn.Context(0) for r in mine}


def file_all_gather(tag, blobs):
    """host-side channel for the handle blobs: write the local ranks', wait for everyone's"""
    for r, b in blobs.items():
        tmp = os.path.join(d, f"{tag}{r}.tmp")
        with open(tmp, "wb") as f:
            f.write(b)
        os.rename(tmp, os.path.join(d, f"{tag}{r}.bin"))
    out, t0 = [], time.time()
    for r in range(world):
        p = os.path.join(d, f"{tag}{r}.bin")
        while not os.path.exists(p):
            if time.time() - t0 > 120:
                raise TimeoutError(p)
            time.sleep(0.01)
        with open(p, "rb") as f:
            out.append(f.read())
    return out


def make_exchanges(tag, count):
    ex = {}
    for r in mine:
        ctxs[r].make_current()
        ex[r] = mn.Exchange(r, world, count)
    handles = file_all_gather(tag, {r: ex[r].export() for r in mine})
    if world > 1:
        for r in mine:
            ctxs[r].make_current()
            ex[r].connect(handles)
    return ex


res = {r: {} for r in mine}
# --- A: raw exchange, ragged count, `out` and fused `target`, three rounds alternating the parity ---------------
count = 10007
ex = make_exchanges("a", count)
out, tgt = {}, {}
for r in mine:
    ctxs[r].make_current()
    out[r] = bla.empty((count,)); tgt[r] = bla.to_device(np.full(count, 1.0, np.float32))
for rnd in range(3):
    for r in mine:
        ctxs[r].make_current()
        g = uniform(1000 * rnd + r, (count,), -1, 1, np.float32)
        native.check(L.bla_memcpy_h2d(ex[r].bucket(rnd & 1), g.ctypes.data, g.nbytes, None)); native.sync()
    for r in mine:     # every local rank's kernel is in flight before any of them is waited for
        ctxs[r].make_current()
        ex[r].allreduce(rnd & 1, out=out[r].ptr, target=tgt[r].ptr, alpha=0.5)
    for r in mine:
        ctxs[r].make_current()
        native.sync()
        res[r][f"sum{rnd}"] = out[r].numpy().copy()
for r in mine:
    ctxs[r].make_current()
    res[r]["target"] = tgt[r].numpy()
    res[r]["status_a"] = np.int32(ex[r].status())

# --- B: data-parallel MNIST-NN steps (graph launch / direct launches) ----------------------------------------------------------
gB = per * world
z = np.load(os.path.join(ROOT, "tests", "golden", "mnist_nn_params.npz"))
x_raw = randint(7, (784, gB), 256).astype(np.float32)
lab = randint(8, (gB,), 10); y = np.zeros((10, gB), np.float32); y[lab, np.arange(gB)] = 1
nn = {}
for r in mine:
    ctxs[r].make_current()
    nn[r] = mn.MnistNN(per, colsum_mode=mn.COLSUM_INTENDED)
    nn[r].set_params([z[n] for n in ["w1", "b1", "w2", "b2", "w3", "b3"]])
    lo, hi = mn.shard_columns(gB, world, r)
    nn[r].load_batch(np.ascontiguousarray(x_raw[:, lo:hi]), np.ascontiguousarray(y[:, lo:hi]))
ex2 = make_exchanges("b", nn[mine[0]].count)
t0 = time.perf_counter()
for i in range(steps):
    for r in mine:
        ctxs[r].make_current()
        nn[r].dp_step(ex2[r], graph=(i % 3 != 2))      # graph replays and direct launches mixed: they share the bucket parity
for r in mine:
    ctxs[r].make_current()
    native.sync()
wall = (time.perf_counter() - t0) / steps * 1e6
for r in mine:
    ctxs[r].make_current()
    res[r]["wall_per_step_us"] = np.float64(wall)
    res[r]["params"] = mn.flatten_params(nn[r].get_params())
    res[r]["status_b"] = np.int32(ex2[r].status())
    np.savez(os.path.join(d, f"result{r}.npz"), **res[r])
# keep the mappings alive until every rank has finished reading its peers
file_all_gather("done", {r: b"x" for r in mine})
for r in mine:
    ctxs[r].make_current()
    ex2[r].close(); ex[r].close()
print(f"ranks {mine} ok ({wall:.1f} us per step)", flush=True)

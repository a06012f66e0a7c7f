#!/usr/bin/env python3
"""The data-parallel MNIST-NN step with world = 1 (the exchange kernel then only reads its own bucket): the part of the N-GPU step
that is not communication, beside the single-GPU fused-update step.  usage: dp_step_floor.py [STEPS]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_pkg
from inputs import randint
bla = load_pkg(); bla.init(0); mn = bla.mnist_nn
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nn = mn.MnistNN(256, colsum_mode=mn.COLSUM_INTENDED)
z = np.load(os.path.join(ROOT, "tests", "golden", "mnist_nn_params.npz"))
P = [z[n] for n in ["w1", "b1", "w2", "b2", "w3", "b3"]]
nn.set_params(P)
x = randint(7, (784, 256), 256).astype(np.float32); lab = randint(8, (256,), 10); y = np.zeros((10, 256), np.float32); y[lab, np.arange(256)] = 1
nn.load_batch(x, y)
ex = mn.Exchange(0, 1, nn.count)
def t(fn):
    nn.set_params(P)
    for _ in range(50): fn()
    bla.sync(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    bla.sync(); return (time.perf_counter() - t0) / steps * 1e6
print(f"algo {os.environ.get('BLA_DP_ALGO', 'default')}: fused single-GPU step {t(lambda: nn.fused_step()):.1f} us | dp step (world 1) graph {t(lambda: nn.dp_step(ex)):.1f} us, "
      f"direct {t(lambda: nn.dp_step(ex, graph=False)):.1f} us | forward+backward only (graph) {t(lambda: nn.graph_step(with_update=False)):.1f} us")

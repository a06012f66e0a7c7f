#!/usr/bin/env python3
"""Run N MNIST-NN steps (eager or graph) for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from __graft_entry__ import load_pkg
from inputs import randint
bla = load_pkg(); bla.init(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
nn = bla.mnist_nn.MnistNN(B)
z = np.load(os.path.join(ROOT, "tests", "golden", "mnist_nn_params.npz"))
nn.set_params([z[n] for n in ["w1", "b1", "w2", "b2", "w3", "b3"]])
x = randint(7, (784, B), 256).astype(np.float32); lab = randint(8, (B,), 10)
y = np.zeros((10, B), np.float32); y[lab, np.arange(B)] = 1
nn.load_batch(x, y)
def graph_unfused():
    nn.graph_step(with_update=False); nn.apply()
f = {"eager": nn.train_step, "graph": nn.graph_step, "graph_unfused": graph_unfused, "direct": nn.fused_step}[mode]
for _ in range(5): f()
bla.sync(); t0 = time.perf_counter()
for _ in range(steps): f()
bla.sync(); dt = time.perf_counter() - t0
print(f"{mode}: {dt/steps*1e6:.1f} us/step, {steps*B/dt:.0f} samples/s")

#!/usr/bin/env python3
"""End-to-end run of examples/mnist_nn_gpu (the C trainer) on a synthetic MNIST-sized dataset: 60,000 rows x 785 values written as the
reference's CSV, `init`, then `train EPOCHS BATCH`.  Prints the program's per-epoch lines: accuracy / loss (stdout) and end-to-end samples/s
(stderr: sampler, order upload, gather, step, metrics -- everything an epoch does except reading the CSV once at start-up)."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
epochs = sys.argv[2] if len(sys.argv) > 2 else "3"
batch = sys.argv[3] if len(sys.argv) > 3 else "256"
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
d = tempfile.mkdtemp()
os.makedirs(os.path.join(d, "data", "mnist_nn")); os.makedirs(os.path.join(d, "data", "mnist"))
rng = np.random.default_rng(0)
lab = rng.integers(0, 10, rows); px = rng.integers(0, 256, (rows, 784)); px[np.arange(rows), lab * 7] = 255
t0 = time.time()
with open(os.path.join(d, "data", "mnist", "mnist_train.csv"), "w") as f:
    for r in range(rows):
        f.write(str(int(lab[r])) + "," + ",".join(map(str, px[r].tolist())) + ",\n")
print(f"wrote {rows} rows in {time.time() - t0:.1f} s", flush=True)
prog = os.path.join(ROOT, "examples", "mnist_nn_gpu")
subprocess.check_call([prog, "init"], cwd=d)
t0 = time.time()
r = subprocess.run([prog, "train", epochs, batch], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
print(r.stdout.strip()); print("\n".join(l for l in r.stderr.splitlines() if "mnist_nn_gpu" in l))
print(f"whole program (CSV parse + upload + {epochs} epochs + save): {time.time() - t0:.1f} s, exit {r.returncode}")

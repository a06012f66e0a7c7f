"""examples/mnist_nn_gpu.c -- the reference's MNIST program (model/mnist_nn.c: init / train / run) as a C host program over the
device-resident trainer.  CPU part: `init` writes byte-for-byte the files the reference's init() writes; the O(log N) sampler yields the
same example order as lib/mnist_csv2.c's walk.  GPU part: a training run and a prediction run on a synthetic MNIST-shaped dataset against
the oracle restatement of the same loop (same sampler order, fp64), including the epoch's loss / accuracy line; and the same run with two
replicas in one process (BLA_GPUS=2) against the single replica."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden

EX = os.path.join(ROOT, "examples")
BIN = os.path.join(EX, "mnist_nn_gpu")
REF = "/root/reference"
FILES = ["weights_1", "biases_1", "weights_2", "biases_2", "weights_3", "biases_3"]
SHAPES = [(256, 784), (256, 1), (128, 256), (128, 1), (10, 128), (10, 1)]


@pytest.fixture(scope="module")
def prog(pkg):
    pkg.build_native()
    subprocess.check_call(["make", "-s", "-C", EX])
    return BIN


def write_dataset(path, rows, seed):
    """label + 784 pixels per row, every value comma-terminated (lib/mnist_csv2.c's input format)"""
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 10, rows); px = rng.integers(0, 256, (rows, 784))
    px[np.arange(rows), lab * 7] = 255          # a learnable signal: one pixel tied to the label
    with open(path, "w") as f:
        for r in range(rows):
            f.write(",".join(str(int(v)) for v in [lab[r]] + list(px[r])) + ",\n")
    return lab, px


def run(prog, args, cwd, env=None, check=True):
    e = dict(os.environ, **(env or {}))
    r = subprocess.run([prog] + args, cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if check:
        assert r.returncode == 0, r.stdout + r.stderr
    return r


def read_weights(d):
    import oracle  # noqa: F401  (sys.path from conftest)
    out = []
    for f, (r, c) in zip(FILES, SHAPES):
        txt = open(os.path.join(d, f + ".csv")).read().replace("\n", "")
        out.append(np.array([float(v) for v in txt.split(",") if v != ""], np.float64).reshape(r, c))
    return out


def test_usage_messages_match_the_reference_shape(prog, tmp_path):
    r = run(prog, [], str(tmp_path), check=False)
    assert r.returncode == 1 and r.stdout.startswith("Please supply an argument, options:\n\trun [<num predictions>]\n\ttrain <num epochs>")
    r = run(prog, ["train"], str(tmp_path), check=False)
    assert r.returncode == 1 and r.stdout.startswith("Please supply a number of epochs, usage:\n\ttrain <num_epochs>")
    r = run(prog, ["bogus"], str(tmp_path), check=False)
    assert r.returncode == 1 and r.stdout.startswith("Unrecognized argument, options:")


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference sources only exist in the build container")
def test_init_writes_the_reference_files(prog, tmp_path):
    """model/mnist_nn.c:97-142 compiled as is (with the reference's own lib units) against ours: same rand() draws, same float
    expressions, same CSV bytes in all six files."""
    ref_exe = str(tmp_path / "ref_mnist_nn")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-w", "-o", ref_exe, f"{REF}/model/mnist_nn.c", f"{REF}/lib/matrix.c", f"{REF}/lib/csv.c",
                           f"{REF}/lib/mnist_csv2.c", "-lm"])
    a, b = tmp_path / "theirs", tmp_path / "ours"
    for d in (a, b):
        (d / "data" / "mnist_nn").mkdir(parents=True)
    subprocess.check_call([ref_exe, "init"], cwd=str(a))
    run(prog, ["init"], str(b))
    for f in FILES:
        assert open(a / "data" / "mnist_nn" / f"{f}.csv", "rb").read() == open(b / "data" / "mnist_nn" / f"{f}.csv", "rb").read(), f


def test_fast_sampler_equals_the_library_walk(prog, tmp_path):
    """take_order() (Fenwick tree, O(log N) per draw) against get_random_data_take's walk (lib/mnist_csv2.c:41-62): identical picks over
    three passes of the dataset, including the start-over after everything was taken and the walk's habit of landing on taken entries."""
    path = str(tmp_path / "d.csv")
    write_dataset(path, 97, 1)
    fast = run(prog, ["order", path, "300", "fast"], str(tmp_path)).stdout.split()[-300:]
    walk = run(prog, ["order", path, "300", "walk"], str(tmp_path)).stdout.split()[-300:]
    assert fast == walk and len(set(fast[:97])) > 40


def sampler_order(prog, path, draws, cwd):
    return [int(v) for v in run(prog, ["order", path, str(draws), "walk"], cwd).stdout.split()[-draws:]]


def oracle_training(ora, params, lab, px, order, batch, epochs):
    """model/mnist_nn.c:182-343 on the oracle: the same example order, fp64, true row sums for the bias gradients."""
    n = len(lab); lines = []
    at = 0
    for e in range(epochs):
        loss = acc = 0.0
        for j in range(0, n, batch):
            idx = order[at + j: at + min(j + batch, n)]
            x = px[idx].T.astype(np.float64); y = np.zeros((10, len(idx))); y[lab[idx], np.arange(len(idx))] = 1
            new, acts, _ = ora.mnist_step(params, x, y, colsum_intended=True)
            l, c = ora.mnist_metrics(acts["a3"], y)
            loss += l; acc += float(c)
            params = new
        at += n
        lines.append((acc / np.float32(n), loss / np.float32(n)))
    return params, lines


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [1, 2])
def test_train_and_run_against_the_oracle(prog, ora, tmp_path, gpus):
    rows, batch, epochs = 700, 256, 2            # batches of 256, 256, 188: the shorter last batch of :194-195 is exercised
    cwd = tmp_path
    (cwd / "data" / "mnist_nn").mkdir(parents=True); (cwd / "data" / "mnist").mkdir()
    lab, px = write_dataset(str(cwd / "data" / "mnist" / "mnist_train.csv"), rows, 5)
    tlab, tpx = write_dataset(str(cwd / "data" / "mnist" / "mnist_test.csv"), 1300, 6)
    run(prog, ["init"], str(cwd))
    p0 = read_weights(str(cwd / "data" / "mnist_nn"))
    env = {}
    if gpus > 1:     # two replicas in one process, both on device 0 here (on a multi-GPU node: one device each)
        env = dict(BLA_GPUS=str(gpus), BLA_SHARE_GPU="1", GPU_MAX_HW_QUEUES="16", BLA_DP_MAX_BLOCKS="64", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = run(prog, ["train", str(epochs), str(batch)], str(cwd), env)
    got_lines = [l for l in r.stdout.splitlines() if l.startswith("Epoch")]
    got = read_weights(str(cwd / "data" / "mnist_nn"))
    # The program draws `rows` picks per epoch after srand(42), resetting the sampler between epochs; the library's sampler starts over by
    # itself once num_sampled == num_examples -- the very state the program's reset creates -- so 2 x rows picks in one go are the same list.
    order = sampler_order(prog, str(cwd / "data" / "mnist" / "mnist_train.csv"), epochs * rows, str(cwd))
    want, lines = oracle_training(ora, p0, lab, px, order, batch, epochs)
    for e, (acc, loss) in enumerate(lines):
        assert got_lines[e].startswith(f"Epoch {e}:\tAvg accuracy: ")
        g_acc = float(got_lines[e].split("Avg accuracy: ")[1].split("\t")[0]); g_loss = float(got_lines[e].split("Avg loss: ")[1])
        assert abs(g_acc - acc) <= 2.0 / rows + 5e-4, (g_acc, acc)         # at most two near-tie predictions apart, plus the %.3f rounding
        assert abs(g_loss - loss) <= 1e-4 * abs(loss) + 5e-6, (g_loss, loss)
    for g, w, name in zip(got, want, FILES):
        assert np.abs(g - w).max() <= 1e-5 * np.abs(w).max() + 1e-6, name         # fp32 device vs fp64 oracle + the CSV's six decimals
    assert max(np.abs(g - p).max() for g, p in zip(got, p0)) > 1e-3                # it did train
    # run: the whole test set, and a prefix
    z = [g.astype(np.float64) for g in got]
    for n_pred in (-1, 300):
        r = run(prog, ["run"] + ([] if n_pred < 0 else [str(n_pred)]), str(cwd))
        n_eff = 1300 if n_pred < 0 else n_pred
        torder = sampler_order(prog, str(cwd / "data" / "mnist" / "mnist_test.csv"), n_eff, str(cwd))
        x = tpx[torder].T.astype(np.float64); y = np.zeros((10, n_eff)); y[tlab[torder], np.arange(n_eff)] = 1
        _, acts, _ = ora.mnist_step(z, x, y, colsum_intended=True)
        _, c = ora.mnist_metrics(acts["a3"], y)
        line = r.stdout.strip().splitlines()[-1]
        assert line.startswith(f"Running predictions for {n_eff} digits...done! Got "), line
        g = int(line.split("Got ")[1].split(" ")[0])
        assert abs(g - c) <= 2 and line.endswith(f"({np.float32(g) / np.float32(n_eff):.3f})."), (line, c)


@pytest.mark.gpu
def test_device_metrics_against_the_reference_fixture(pkg):
    """Loss / accuracy of the golden B = 256 step (tests/golden/mnist_step.npz: batch_loss by the reference's own cross_entropy_loss,
    num_correct by :240-250) from the device-side accumulators; two steps accumulate, reset clears."""
    from inputs import randint
    pkg.init(0)
    mn = pkg.mnist_nn
    g = golden("mnist_step")
    z = np.load(os.path.join(ROOT, "tests", "golden", "mnist_nn_params.npz"))
    B = 256
    x_raw = randint(3000 + B, (784, B), 256).astype(np.float32); lab = randint(3100 + B, (B,), 10)
    y = np.zeros((10, B), np.float32); y[lab, np.arange(B)] = 1
    nn = mn.MnistNN(B, colsum_mode=mn.COLSUM_INTENDED)
    nn.set_params([z[n] for n in ["w1", "b1", "w2", "b2", "w3", "b3"]]); nn.load_batch(x_raw, y)
    L = pkg.lib(); chk = pkg.native.check
    chk(L.bla_mnist_nn_metrics_enable(nn.h, 1))
    loss, corr = C.c_double(), C.c_longlong()
    chk(L.bla_mnist_nn_forward(nn.h, None, None, None))
    chk(L.bla_mnist_nn_metrics_read(nn.h, C.byref(loss), C.byref(corr), 0))
    want_l, want_c = float(g["b256_intended_batch_loss"]), int(g["b256_intended_num_correct"])
    assert abs(loss.value - want_l) <= 1e-5 * want_l and corr.value == want_c, (loss.value, want_l, corr.value, want_c)
    chk(L.bla_mnist_nn_forward(nn.h, None, None, None))
    chk(L.bla_mnist_nn_metrics_read(nn.h, C.byref(loss), C.byref(corr), 1))
    assert abs(loss.value - 2 * want_l) <= 2e-5 * want_l and corr.value == 2 * want_c
    nn.fused_step()          # metrics ride along a training step as well (same forward pass)
    chk(L.bla_mnist_nn_metrics_read(nn.h, C.byref(loss), C.byref(corr), 1))
    assert abs(loss.value - want_l) <= 1e-5 * want_l and corr.value == want_c

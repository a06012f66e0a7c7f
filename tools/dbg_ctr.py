import os, sys, subprocess, ctypes as C, tempfile, pathlib
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/golden"); sys.path.insert(0, ROOT + "/oracle")
import oracle as ora
ora.build()
import test_c_trainer as T
from __graft_entry__ import load_pkg
pkg = load_pkg(); pkg.init(0)
prog = T.BIN
cwd = pathlib.Path(tempfile.mkdtemp())
(cwd / "data" / "mnist_nn").mkdir(parents=True); (cwd / "data" / "mnist").mkdir()
rows, batch, epochs = 700, 256, 1
lab, px = T.write_dataset(str(cwd / "data" / "mnist" / "mnist_train.csv"), rows, 5)
T.run(prog, ["init"], str(cwd))
p0 = T.read_weights(str(cwd / "data" / "mnist_nn"))
import shutil
TL = "/usr/local/lib/python3.10/dist-packages/torch/lib/libamdhip64.so"
for trial, env in enumerate([{"BLA_MNIST_SELFCHECK": "1"}, {"BLA_MNIST_SELFCHECK": "1"}, {}]):
    T.run(prog, ["init"], str(cwd))
    r = T.run(prog, ["train", str(epochs), str(batch)], str(cwd), dict(env))
    print(trial, env, [l for l in r.stdout.splitlines() if l.startswith("Epoch")], [l for l in r.stderr.splitlines() if "selfcheck] order" in l])
    for tr2 in ("512", "700"):
        pass
lab, px = T.write_dataset(str(cwd / "data" / "mnist" / "mnist_train.csv"), rows, 5)
order = T.sampler_order(prog, str(cwd / "data" / "mnist" / "mnist_train.csv"), epochs * rows, str(cwd))
mn = pkg.mnist_nn
L = pkg.lib(); chk = pkg.native.check
# the same sequence driven from Python on persistent trainers: gather kernel + fused step, tail trainer sharing the bucket
store_X = np.ascontiguousarray(px.T.astype(np.float32)); store_y = lab.astype(np.float32)
dX = pkg.to_device(store_X); dy = pkg.to_device(store_y); dord = pkg.to_device(np.array(order, np.int32), dtype=np.int32)
nn = mn.MnistNN(256, colsum_mode=mn.COLSUM_INTENDED); tail = mn.MnistNN(188, colsum_mode=mn.COLSUM_INTENDED)
chk(L.bla_mnist_nn_metrics_enable(nn.h, 1)); chk(L.bla_mnist_nn_metrics_enable(tail.h, 1))
tail.use_buckets(nn.params_ptr, nn.grads_ptr)
nn.set_params([p.astype(np.float32) for p in p0])
params = p0
tot_l = tot_c = 0
for j in range(0, rows, batch):
    idx = order[j:min(j + batch, rows)]
    t = nn if len(idx) == 256 else tail
    chk(L.bla_mnist_nn_gather_batch(t.h, None, dX.ptr, dy.ptr, rows, dord.ptr + 4 * j))
    t.fused_step()
    x = px[idx].T.astype(np.float64); y = np.zeros((10, len(idx))); y[lab[idx], np.arange(len(idx))] = 1
    new, acts, _ = ora.mnist_step(params, x, y, colsum_intended=True)
    l, c = ora.mnist_metrics(acts["a3"], y)
    loss, corr = C.c_double(), C.c_longlong()
    chk(L.bla_mnist_nn_metrics_read(t.h, C.byref(loss), C.byref(corr), 1))
    got = mn.flatten_params(nn.get_params()); want = mn.flatten_params([p.astype(np.float32) for p in new])
    print(f"batch {j}: oracle loss {l:.5f} correct {c} | device loss {loss.value:.5f} correct {corr.value} | params rel diff {np.linalg.norm(got - want) / np.linalg.norm(want):.2e}")
    params = new; tot_l += l; tot_c += c
print("oracle epoch", tot_c / 700, tot_l / 700)

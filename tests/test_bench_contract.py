"""bench.py's one-line JSON contract: the keys the driver reads, BASELINE.json's metric, the `roofline` and `cpu_baseline` objects.
CPU: the committed line of the last measurement pass (profiles/r03_bench_default_output.json).  GPU: a short live run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

TOP = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"]
ROOF = ["bound", "achieved", "peak", "unit", "frac", "traffic"]
CPU = ["value", "unit", "cores", "kind", "sample"]


def check_line(d, with_cpu_baseline):
    for k in TOP:
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"].split(";")[0].strip()                      # "fp32 GFLOP/s matrix_mul 4096^3"
    assert d["unit"] == "GFLOP/s" and d["dtype"] == "f32" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and d["data"] == "synthetic"
    for k in ROOF:
        assert k in d["roofline"], k
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert abs(d["value"] - d["n_gpus"] * 2 * 4096 ** 3 / (d["ms_per_step"] * 1e-3) / 1e9) <= 2e-3 * d["value"]     # whole-job GFLOP/s over the timed region
    sec, ter = d["secondary"], d["tertiary"]
    assert sec["unit"] == "samples/s" and sec["value"] > 0 and "roofline" in sec
    assert ter["unit"] == "images/s" and ter["value"] > 0 and 0 < ter["roofline"]["frac"] < 1
    un = ter["unet_batch_64"]
    assert un["unit"] == "images/s" and un["value"] > 0 and un["finite_output"] is True and 0 < un["roofline"]["frac"] < 1
    if with_cpu_baseline:
        for obj in (d, sec, ter, un):
            for k in CPU:
                assert k in obj["cpu_baseline"], k
        # the three timed CPU baselines are the reference itself (oracle/_ref), one core each; the U-Net's cites the reference program's own run
        for obj in (d, sec, ter):
            assert obj["cpu_baseline"]["kind"] == "reference" and obj["cpu_baseline"]["cores"] == 1, obj["cpu_baseline"]
        assert un["cpu_baseline"]["kind"] == "reference"
        # the pass the U-Net figure comes from is checked against the committed fp64 prediction of image 0 (tests/golden/unet_refconst.npz)
        assert 0 < un["prediction_rel_err_vs_oracle_image0"] <= 2 * un["reference_fp32_distance"]
        assert d["roofline"]["traffic"] and "r03_gemm4096_traffic.json" in d["roofline"]["traffic_source"]
        assert ter["roofline"]["traffic"] and "profiles/r03/conv128.summary.json" in ter["roofline"]["traffic_source"]


def test_committed_bench_line_keeps_the_contract():
    check_line(json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default_output.json"))), True)


def test_committed_rehearsals_of_the_multi_rank_path():
    """bench.py --gpus N under torch.distributed.run with the ranks sharing the one GPU (tools/measure_r03.sh part a, BLA_BENCH_STRICT=1): the line of rank 0 keeps
    the contract's shape, whole-job value = N replicas, no fault or fallback key set."""
    for name, n in (("r03_bench_two_ranks_shared_gpu_rehearsal.json", 2), ("r03_bench_four_ranks_shared_gpu_rehearsal.json", 4)):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert d["n_gpus"] == n and d["scaling"] == "weak" and d["config"]["parallelism"] == f"replicas x{n}"
        assert abs(d["value"] - n * 2 * 4096 ** 3 / (d["ms_per_step"] * 1e-3) / 1e9) <= 2e-3 * d["value"]
        sec = d["secondary"]
        assert sec["value"] > 0 and not sec.get("exchange_fallback") and sec.get("exchange_fault") is None and sec.get("rccl_fault") is None and "direct" in sec["legs"]


@pytest.mark.gpu
def test_live_bench_line_keeps_the_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--mnist-steps", "20", "--mnist-warmup", "5", "--conv-steps", "2",
                        "--unet-steps", "1", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["steps"] == 3 and d["warmup"] == 2 and d["n_gpus"] == 1
    check_line(d, False)

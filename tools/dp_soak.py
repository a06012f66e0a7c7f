#!/usr/bin/env python3
"""Soak run of the gradient exchange on one GPU box: 3000 data-parallel steps with 2 / 3 / 4 processes, both forms of the kernel;
every rank must report a clean status and bit-identical, finite parameters (run from the repo root)."""
import os, sys, tempfile, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import test_dp_exchange_gpu as t
for world, algo in [(4, "twoshot"), (4, "oneshot"), (2, "oneshot"), (3, "twoshot")]:
    with tempfile.TemporaryDirectory() as d:
        res = t.run_ranks(world, d, 3000, 64, algo)
        ok = all(int(r["status_a"]) == 0 and int(r["status_b"]) == 0 for r in res)
        same = all(np.array_equal(r["params"], res[0]["params"]) for r in res)
        fin = bool(np.isfinite(res[0]["params"]).all())
        print(world, algo, "status ok", ok, "identical", same, "finite", fin, "us/step", float(res[0]["wall_per_step_us"]), flush=True)
        assert ok and same and fin
print("soak ok")

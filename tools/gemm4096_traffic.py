#!/usr/bin/env python3
"""The headline kernel's HBM-side traffic from the three separate rocprofv3 --pmc passes tools/measure_r03.sh makes of `bench.py --no-cpu-baseline
--mnist-steps 0 --conv-steps 0 --steps 10 --warmup 5`:   gemm4096_traffic.py OUTDIR > profiles/r03_gemm4096_traffic.json
OUTDIR/gemm4096_{FETCH_SIZE,WRITE_SIZE,mfma}/.../*_counter_collection.csv.  FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests
at 64 B for 16 B/lane streaming reads, LDS-DMA included); WRITE_SIZE is exact; both in KB per launch, averaged over the launches of the 256x256 kernel."""
import collections, csv, glob, json, os, sys
base = sys.argv[1]


def counters(sub):
    f = glob.glob(os.path.join(base, sub, "*", "*_counter_collection.csv"))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])) if f else []:
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def pick(acc):
    ks = [k for k in acc if "gemm_f32_glds_kernel<256, 256, 16" in k]
    return (ks[0], acc[ks[0]]) if ks else (None, {})


kf, f = pick(counters("gemm4096_FETCH_SIZE")); kw, w = pick(counters("gemm4096_WRITE_SIZE")); km, m = pick(counters("gemm4096_mfma"))
avg = lambda v: sum(v) / len(v) if v else None
fetch, write = avg(f.get("FETCH_SIZE", [])), avg(w.get("WRITE_SIZE", []))
busy, active = avg(m.get("SQ_VALU_MFMA_BUSY_CYCLES", [])), avg(m.get("GRBM_GUI_ACTIVE", []))
out = {"workload": "square fp32 matrix_multiply N=4096", "kernel": "gemm_f32_glds256x256x16_nn_dma_splitk1", "kernel_symbol": kf,
       "launches_averaged": len(f.get("FETCH_SIZE", [])), "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write}
if fetch is not None and write is not None:
    out.update(fetch_bytes_raw=fetch * 1024, fetch_bytes_corrected=2 * fetch * 1024, write_bytes=write * 1024, hbm_bytes_per_launch=2 * fetch * 1024 + write * 1024,
               algorithmic_bytes_per_launch=3 * 4096 * 4096 * 4)
if busy is not None and active:
    # busy = SQ_VALU_MFMA_BUSY_CYCLES (summed over 1,024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1,024) -- tools/mfma_busy_summary.py.  The profiler's per-dispatch
    # value saturates at 2^31, which this launch reaches (2.2 M cycles x 1,024 SIMDs x 0.94): then the figure is a LOWER bound and the unsaturated one comes from
    # the same kernel at a shorter contraction (profiles/r03_mfma_busy_gemm_conv.txt: 4096 x 4096 x 1536)
    cyc = active / 8
    out.update(SQ_VALU_MFMA_BUSY_CYCLES=busy, GRBM_GUI_ACTIVE=active, mfma_pipe_busy=busy / (cyc * 1024), mfma_counter_saturated=bool(busy >= 2 ** 31 - 1))
out["note"] = ("separate --pmc passes on the final round-3 tree (tools/measure_r03.sh): `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace "
               "-- python3 bench.py --no-cpu-baseline --mnist-steps 0 --conv-steps 0 --steps 10 --warmup 5`; Infinity-Cache hits are counted in FETCH_SIZE, so this is "
               "L2-fill traffic (DESIGN.md 3.1: 256 tiles of 256x256, one per CU, (8 + 4) panels x 4 MiB x 8 XCDs = 403 MB; both operands fit the 256 MiB Infinity Cache)")
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from a `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv` run:
busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), effective clock = (GRBM_GUI_ACTIVE / 8) / duration.
usage: mfma_busy_summary.py DIR [min_us]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
disp = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    d = disp[r["Dispatch_Id"]]
    d["name"] = r["Kernel_Name"]; d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d["grid"] = r["Grid_Size"]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(list)
for d in disp.values():
    if d["us"] < min_us or "GRBM_GUI_ACTIVE" not in d or "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
        continue
    cyc = d["GRBM_GUI_ACTIVE"] / 8
    agg[(d["name"].split("(")[0][-100:], d["grid"])].append((d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), cyc / d["us"] / 1e3, d["us"]))
for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(x[2] for x in kv[1])):
    n = len(v)
    print(f"{name:100s} grid {grid:>9s} n {n:4d}  avg {sum(x[2] for x in v) / n:9.1f} us  MFMA pipe busy {sum(x[0] for x in v) / n:5.3f}  clock {sum(x[1] for x in v) / n:5.3f} GHz")

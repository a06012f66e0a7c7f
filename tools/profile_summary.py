#!/usr/bin/env python3
"""Join the rocprofv3 passes of one tools/profile_targets.py workload into one table:
    profile_summary.py OUTDIR TARGET    (OUTDIR/TARGET/{stats,fetch,write}/... as written by tools/profile_r02.sh)
Per kernel: calls, average duration (kernel trace), FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes), HBM-side traffic with the
gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-B requests at 64 B for wide streaming reads: doubled; WRITE_SIZE exact).
The warm-up launches (3) are in the averages, like the r01 files."""
import collections, csv, glob, json, os, sys
base, target = sys.argv[1], sys.argv[2]
d = os.path.join(base, target)


def short(n):
    n = n.replace("void bla::", "").replace("bla::", "")
    return n.split("(")[0] if "<" not in n else n[:n.rfind(">") + 1] if n.rfind(">") > 0 else n


def trace(sub):
    f = glob.glob(os.path.join(d, sub, "*", "*_kernel_trace.csv"))
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])) if f else []:
        rows[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return rows


def counter(sub, name):
    f = glob.glob(os.path.join(d, sub, "*", "*_counter_collection.csv"))
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])) if f else []:
        if r["Counter_Name"] == name:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


dur, fe, wr = trace("stats"), counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
info = {}
for f in glob.glob(os.path.join(d, "stats.json")):
    for line in open(f):                      # the target's own JSON line (the profiler may print around it)
        if line.startswith("{"):
            info = json.loads(line)
out = {"target": target, "info": info, "kernels": []}
tot_us = tot_bytes = 0.0
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if k.startswith("__amd_rocclr"):
        continue
    per = len(v) / (info.get("iters", 0) + 3) if info.get("iters") else None
    fk = sum(fe[k]) / len(fe[k]) if k in fe else None
    wk = sum(wr[k]) / len(wr[k]) if k in wr else None
    traffic = (2 * fk + wk) * 1024 if fk is not None and wk is not None else None
    avg = sum(v) / len(v)
    row = {"kernel": k, "calls": len(v), "launches_per_iteration": per, "avg_us": round(avg, 2), "min_us": round(min(v), 2),
           "FETCH_SIZE_KB": fk and round(fk, 1), "WRITE_SIZE_KB": wk and round(wk, 1), "traffic_bytes_per_launch": traffic and round(traffic),
           "traffic_GBps": traffic and round(traffic / avg / 1e3, 1)}
    out["kernels"].append(row)
    if per:
        tot_us += avg * per
        if traffic:
            tot_bytes += traffic * per
out["iteration"] = {"kernel_us": round(tot_us, 2), "traffic_bytes": round(tot_bytes)}
for key in ("bytes", "bytes_forward", "bytes_backward"):
    if key in info:
        out["iteration"]["algorithmic_" + key] = info[key]
if "bytes" in info and tot_us:
    out["iteration"]["algorithmic_GBps"] = round(info["bytes"] / tot_us / 1e3, 1)
    out["iteration"]["frac_of_8TBps"] = round(info["bytes"] / tot_us / 1e3 / 8000, 3)
    out["iteration"]["traffic_over_algorithmic"] = round(tot_bytes / info["bytes"], 2) if tot_bytes else None
fl = info.get("flops") or (info.get("flops_forward", 0) + info.get("flops_backward", 0))
if fl and tot_us:
    out["iteration"]["TFLOPs"] = round(fl / tot_us / 1e6, 2)
    out["iteration"]["frac_of_157.3"] = round(fl / tot_us / 1e6 / 157.3, 3)
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Average of one (derived) counter per kernel from a rocprofv3 --pmc NAME --kernel-trace --output-format csv run.  usage: pmc_avg.py DIR [min_us]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if us >= min_us:
        agg[(r["Kernel_Name"].split("(")[0][-90:], r["Counter_Name"])].append((float(r["Counter_Value"]), us))
for (name, ctr), v in sorted(agg.items()):
    print(f"{name:90s} {ctr:18s} n {len(v):4d}  avg {sum(x[0] for x in v) / len(v):12.3f}   ({sum(x[1] for x in v) / len(v):8.1f} us)")

#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name average duration and the average idle gap before it
(start - previous kernel's end), over the last `steps` repetitions of a 9-kernel step.  usage: trace_gaps.py kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-9 * 40:]
dur = collections.OrderedDict(); gap = collections.defaultdict(list)
prev_end = None
for r in tail:
    n = r["Kernel_Name"][:70]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur.setdefault(n, []).append(e - s)
    if prev_end is not None: gap[n].append(s - prev_end)
    prev_end = e
tot_d = tot_g = 0
for n, d in dur.items():
    g = gap[n]
    print(f"{n:<72} x{len(d):3d}  dur {sum(d)/len(d)/1e3:7.2f} us   gap before {sum(g)/max(len(g),1)/1e3:6.2f} us")
    tot_d += sum(d); tot_g += sum(g)
print(f"per step (9 kernels): busy {tot_d/40/1e3:.1f} us + gaps {tot_g/40/1e3:.1f} us")

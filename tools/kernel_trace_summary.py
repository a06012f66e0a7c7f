#!/usr/bin/env python3
"""Median duration per (kernel, grid) of a rocprofv3 --kernel-trace --output-format csv run.  usage: kernel_trace_summary.py DIR [SUBSTRING ...]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
want = sys.argv[2:]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if not want or any(w in n for w in want):
        d[(n.split('(')[0][-48:], r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    v.sort(); print(f"{k[0]:50s} grid {k[1]:>8s} n {len(v):5d} median {v[len(v) // 2]:7.2f} us")

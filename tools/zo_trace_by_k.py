#!/usr/bin/env python
"""Per-launch durations of the ZoomOut iteration's kernels by map size, from a rocprofv3 kernel trace.
usage: python tools/zo_trace_by_k.py <s_kernel_trace.csv>   (trace of `bench.py --workload zoomout --steps 1 --warmup 1`)"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    for key in ("simnn_pipe_kernel", "zo_embed_split", "p2pfm_direct", "zo_merge", "zo_exact"):
        if key in n:
            per[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# us per launch by iteration (last 150 launches of each kernel = the timed step; k = 50 + iteration)")
print("k      " + "  ".join(f"{k:>18s}" for k in per))
L = {k: v[-150:] for k, v in per.items()}
for it in range(0, 150, 6):
    print(f"{50 + it:4d}   " + "  ".join(f"{L[k][it]:18.1f}" if it < len(L[k]) else " " * 18 for k in per))
print("mean   " + "  ".join(f"{sum(L[k]) / len(L[k]):18.1f}" for k in per))

#!/usr/bin/env python
"""Timeline of the last compute_surface_map_batch call in a rocprofv3 kernel trace (tools/surface_map_streams.py 2 under
rocprofv3 --kernel-trace): per queue the busy intervals by kernel group, and where no kernel of any queue runs."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows]
ev.sort()
# calls are separated by idle gaps > 20 ms? no: take the last 0.6 s and cut at the largest gap
t_end = ev[-1][1]
ev = [e for e in ev if e[0] > t_end - int(0.9e9)]
gaps = sorted(((ev[i + 1][0] - max(x[1] for x in ev[:i + 1]), i) for i in range(len(ev) - 1)), reverse=True)
# the last call starts after the last gap longer than 3 ms
cut = 0
for g, i in gaps:
    if g > 3e6:
        cut = max(cut, i + 1)
ev = ev[cut:]
t0 = ev[0][0]
print(f"last call: {len(ev)} kernels over {(ev[-1][1] - t0) / 1e6:.1f} ms")


def group(name):
    for key in ("fit_fused", "lsa_", "eig_", "spmm", "unit_columns", "jacobi", "polar", "precise", "simnn", "lap_", "gemm", "gred", "embed"):
        if key in name:
            return key.strip("_")
    return name[:24]


byq = defaultdict(list)
for s, e, n, q in ev:
    byq[q].append((s - t0, e - t0, group(n)))
for q, lst in byq.items():
    # merge consecutive kernels of the same group (gaps < 0.3 ms)
    out = []
    for s, e, g in lst:
        if out and out[-1][2] == g and s - out[-1][1] < 3e5:
            out[-1][1] = max(out[-1][1], e)
            out[-1][3] += (e - s)
        else:
            out.append([s, e, g, e - s])
    print(f"queue {q}:")
    for s, e, g, busy in out:
        if e - s > 1e6:
            print(f"   {s / 1e6:7.1f} .. {e / 1e6:7.1f} ms  {g:14s} busy {busy / 1e6:6.1f} ms")
# idle: no kernel anywhere
iv = sorted((s - t0, e - t0) for s, e, _, _ in ev)
idle, cur_end = 0, 0
holes = []
for s, e in iv:
    if s > cur_end:
        idle += s - cur_end
        if s - cur_end > 5e5:
            holes.append((cur_end, s))
    cur_end = max(cur_end, e)
print(f"no kernel running: {idle / 1e6:.1f} ms; holes > 0.5 ms: " + ", ".join(f"{a / 1e6:.1f}-{b / 1e6:.1f}" for a, b in holes))

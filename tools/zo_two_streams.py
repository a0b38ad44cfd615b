#!/usr/bin/env python
"""Would two independent half-batches on two streams overlap ZoomOut's dependent launches?  One call with 32 pairs against two
concurrent calls with 16 pairs each (two engines, two streams, two host threads).  usage: python tools/zo_two_streams.py"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from densematcher_amd.engine import MatchEngine

w = dict(bench.WORKLOADS["zoomout"])
host = bench.make_batch(w, 0)
B = w["B"]
dev0 = torch.device("cuda", 0)
full = {k_: torch.as_tensor(v).to(dev0) for k_, v in host.items() if k_ in ("Phi1", "Phi2", "a2")}
C0 = torch.eye(50, dtype=torch.float64, device=dev0).repeat(B, 1, 1)
eng = MatchEngine(0)


def run(e, d, c0, reps):
    for _ in range(reps):
        e.zoomout(d["Phi1"], d["Phi2"], d["a2"], c0, nit=150, step=1)


run(eng, full, C0, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(eng, full, C0, 3)
torch.cuda.synchronize()
t_one = (time.perf_counter() - t0) / 3
print(f"one call, {B} pairs: {1e3 * t_one:.2f} ms  = {B / t_one:.1f} pairs/s", flush=True)

for G in (2, 4):
    streams = [torch.cuda.Stream(dev0) for _ in range(G)]
    engs, parts, c0s = [], [], []
    for g, s in enumerate(streams):
        with torch.cuda.stream(s):
            engs.append(MatchEngine(0))
        sl = slice(g * B // G, (g + 1) * B // G)
        parts.append({k_: v[sl].contiguous() for k_, v in full.items()})
        c0s.append(C0[sl].contiguous())
    torch.cuda.synchronize()

    def worker(g, reps):
        with torch.cuda.stream(streams[g]):
            run(engs[g], parts[g], c0s[g], reps)
            streams[g].synchronize()

    for reps in (1, 3):
        ths = [threading.Thread(target=worker, args=(g, reps)) for g in range(G)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    print(f"{G} concurrent calls, {B // G} pairs each: {1e3 * dt:.2f} ms  = {B / dt:.1f} pairs/s", flush=True)

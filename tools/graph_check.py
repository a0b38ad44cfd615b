#!/usr/bin/env python
"""Does replaying a captured HIP graph shorten the gaps between the dependent launches of a call?  ZoomOut (756 launches per
step), ICP (610) and the config-2 step (11) as plain launches on a side stream against one graph launch per step.
usage: python tools/graph_check.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from densematcher_amd.engine import MatchEngine

side = torch.cuda.Stream()
with torch.cuda.stream(side):
    eng = MatchEngine(0)                      # the context is bound to the side stream (dm_create takes any stream)
    for wl in ("zoomout", "icp", "fmap"):
        w = dict(bench.WORKLOADS[wl])
        host = bench.make_batch(w, 0, "f64")
        dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
        B, k = w["B"], w["k"]
        if wl == "zoomout":
            C0 = torch.eye(50, dtype=torch.float64, device=eng.device).repeat(B, 1, 1)
            step = lambda: eng.zoomout(dev["Phi1"], dev["Phi2"], dev["a2"], C0, nit=150, step=1)
        elif wl == "icp":
            C0 = torch.eye(k, dtype=torch.float64, device=eng.device).repeat(B, 1, 1)
            step = lambda: eng.icp(dev["Phi1"], dev["Phi2"], C0, nit=10)
        else:
            step = lambda: eng.match(dev, k=k)
        for _ in range(3):
            out = step()
        side.synchronize()
        reps = 20 if wl == "fmap" else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            out = step()
        side.synchronize()
        t_plain = (time.perf_counter() - t0) / reps
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=side):
                outg = step()
            g.replay(); side.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            side.synchronize()
            t_graph = (time.perf_counter() - t0) / reps
            same = True
            a, b = (out, outg)
            if isinstance(a, dict):
                same = all(torch.equal(a[n], b[n]) for n in a)
            elif isinstance(a, (tuple, list)):
                same = all(torch.equal(x, y) for x, y in zip(a, b))
            else:
                same = torch.equal(a, b)
            print(f"{wl}: plain launches {1e3 * t_plain:.3f} ms per step, graph replay {1e3 * t_graph:.3f} ms  ({B / t_plain:.1f} -> {B / t_graph:.1f} pairs/s), same results: {same}", flush=True)
        except Exception as e:                    # noqa: BLE001
            print(f"{wl}: plain launches {1e3 * t_plain:.3f} ms per step; capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)

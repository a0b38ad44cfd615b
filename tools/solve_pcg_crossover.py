#!/usr/bin/env python
"""r06: where does the batched iteration (solve_pcg = 1) beat the direct solvers (0) as the batch grows?  Solver kernel time per call (us).
usage: python tools/solve_pcg_crossover.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402

eng = MatchEngine(0)
for wl, Bs in (("fmap", (1, 4, 8, 16, 24, 32, 64)), ("stress", (1, 2, 4, 8, 16))):
    w = dict(bench.WORKLOADS[wl]); w["B"] = max(Bs)
    host = bench.make_batch(w, 0, "f64")
    k = w["k"]
    for B in Bs:
        dev = {q: torch.as_tensor(v[:B]).to(eng.device) for q, v in host.items()}
        row = []
        for mode in (0, 1):
            eng.set_option("solve_pcg", mode)
            for _ in range(3):
                eng.fmap_fit(dev["Phi1"], dev["Phi2"], dev["a1"], dev["a2"], dev["F1"], dev["F2"], dev["lam1"], dev["lam2"], 1e4, 1e3, k1=k, k2=k)
            torch.cuda.synchronize()
            eng.profile_kernel("*")
            for _ in range(5):
                eng.fmap_fit(dev["Phi1"], dev["Phi2"], dev["a1"], dev["a2"], dev["F1"], dev["F2"], dev["lam1"], dev["lam2"], 1e4, 1e3, k1=k, k2=k)
            rep = eng.profile_report()
            eng.profile_kernel("")
            row.append(sum(1e3 * ms / 5 for q, (cnt, ms) in rep.items() if q.startswith("fmap_solve")))
        print(f"{wl:6s} k = {k} B = {B:3d} ({B * k:6d} systems): direct {row[0]:8.1f} us   iteration (+ pack, fall-back launch) {row[1]:8.1f} us", flush=True)
eng.set_option("solve_pcg", 1)

#!/usr/bin/env python
"""Time fmap_solve_chol alone (config 2: 64 pairs, k = 128) with the library's own launch profiler."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from densematcher_amd.engine import MatchEngine

w = dict(bench.WORKLOADS["fmap"])
host = bench.make_batch(w, 0)
eng = MatchEngine(0)
dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
k = int(sys.argv[1]) if len(sys.argv) > 1 else w["k"]
A = eng.project(dev["Phi1"], dev["a1"], dev["F1"], k)
Bm = eng.project(dev["Phi2"], dev["a2"], dev["F2"], k)
c00 = eng.c00(dev["Phi1"], dev["Phi2"], dev["a1"], dev["a2"])
lam1, lam2 = dev["lam1"][:, :k].contiguous(), dev["lam2"][:, :k].contiguous()
for rep in range(3):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        C = eng.fmap_solve(A, Bm, lam1, lam2, c00, 1e4, 1e3)
        torch.cuda.synchronize()
    eng.profile_kernel("fmap_solve_chol")
    for _ in range(20):
        C = eng.fmap_solve(A, Bm, lam1, lam2, c00, 1e4, 1e3)
    torch.cuda.synchronize()
    n, ms = eng.profile_read()
    eng.profile_kernel("")
    print(f"fmap_solve_chol k={k}: {1e3 * ms / n:8.1f} us per launch", flush=True)

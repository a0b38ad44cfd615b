#!/usr/bin/env python
"""Time fmap_solve_chol alone (64 pairs, k from the command line, random descriptors) with the library's own launch profiler."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from densematcher_amd.engine import MatchEngine

import numpy as np
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
eng = MatchEngine(0)
rng = np.random.default_rng(0)
B, D = 64, 384
A = torch.as_tensor(rng.standard_normal((B, k, D)).astype(np.float32) * 0.1).to(eng.device)
Bm = torch.as_tensor(rng.standard_normal((B, k, D)).astype(np.float32) * 0.1).to(eng.device)
lam = np.sort(rng.uniform(0, 50, (B, k)), axis=1); lam[:, 0] = 0
lam1 = torch.as_tensor(lam).to(eng.device)
lam2 = torch.as_tensor(lam * 1.1).to(eng.device)
c00 = torch.ones(B, dtype=torch.float64, device=eng.device)
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

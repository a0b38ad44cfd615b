#!/usr/bin/env python
"""Experiment: time the packed-storage solver (k = 200) with phases disabled (DM_SOLVE_DEBUG)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densematcher_amd.engine import MatchEngine
eng = MatchEngine(0)
k, D, B = int(sys.argv[1]) if len(sys.argv) > 1 else 200, 384, 64
rng = np.random.default_rng(0)
A = torch.tensor(rng.standard_normal((B, k, D)).astype(np.float32) * 0.05, device="cuda")
Bm = torch.tensor(rng.standard_normal((B, k, D)).astype(np.float32) * 0.05, device="cuda")
lam = torch.tensor(np.sort(rng.uniform(0, 100, (B, k)), axis=1), device="cuda")
c00 = torch.ones(B, dtype=torch.float64, device="cuda")
os.environ["DM_SOLVE_PACKED"] = "1"
for mode in ["0", "1", "4", "5", "2", "3"]:
    os.environ["DM_SOLVE_DEBUG"] = mode
    eng.fmap_solve(A, Bm, lam, lam, c00, 1e4, 1e3)
    eng.profile_kernel("fmap_solve_chol")
    for _ in range(3):
        eng.fmap_solve(A, Bm, lam, lam, c00, 1e4, 1e3)
    n, ms = eng.profile_read()
    eng.profile_kernel("")
    print(f"k={k} DM_SOLVE_DEBUG={mode}: fmap_solve_chol avg {ms / n * 1e3:.0f} us")

#!/usr/bin/env python
"""Experiment: time the fused G-tile kernel with the epilogue or the main loop disabled (DM_GRED_DEBUG)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densematcher_amd.engine import MatchEngine
eng = MatchEngine(0)
B, N, k = 64, 2048, 128
g = torch.Generator(device="cuda").manual_seed(0)
Phi1 = torch.randn(B, N, k, device="cuda", generator=g) * 0.02
Phi2 = torch.randn(B, N, k, device="cuda", generator=g) * 0.02
a1 = torch.rand(B, N, device="cuda", generator=g) / N
C = torch.randn(B, k, k, device="cuda", dtype=torch.float64, generator=g)
configs = [("0", "2", "0"), ("0", "0", "0"), ("1", "2", "0"), ("2", "2", "0")]
res = {c: [] for c in configs}
for rep in range(3):          # interleaved rounds: the first launches of a process run at different clocks
    for c in configs:
        os.environ["DM_GRED_DEBUG"], os.environ["DM_GRED_STAGGER"], os.environ["DM_GRED_PRIO"] = c
        eng.fm_to_p2p(Phi1, Phi2, a1, C)
        eng.profile_kernel("gred_f64")
        for _ in range(5):
            eng.fm_to_p2p(Phi1, Phi2, a1, C)
        n, ms = eng.profile_read()
        eng.profile_kernel("")
        res[c].append(ms / n * 1e3)
for c, v in res.items():
    print(f"DM_GRED_DEBUG={c[0]} STAGGER={c[1]} PRIO={c[2]}: gred_f64 avg us per round: " + " ".join(f"{x:.1f}" for x in v)
          + f"   best {2.0 * N * N * k * B / (min(v) * 1e-6) / 1e12:.1f} TFLOP/s")

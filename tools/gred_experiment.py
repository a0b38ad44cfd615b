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
for mode, stag, prio in [("0", "0", "0"), ("0", "2", "0"), ("0", "0", "1"), ("0", "2", "1"), ("0", "4", "1"), ("1", "2", "1"), ("2", "2", "1")]:
    os.environ["DM_GRED_DEBUG"] = mode
    os.environ["DM_GRED_STAGGER"] = stag
    os.environ["DM_GRED_PRIO"] = prio
    for _ in range(2):
        eng.fm_to_p2p(Phi1, Phi2, a1, C)
    eng.profile_kernel("gred_f64")
    for _ in range(5):
        eng.fm_to_p2p(Phi1, Phi2, a1, C)
    n, ms = eng.profile_read()
    eng.profile_kernel("")
    print(f"DM_GRED_DEBUG={mode} STAGGER={stag} PRIO={prio}: gred_f64 avg {ms / n * 1e3:.1f} us  ({2.0 * N * N * k * B / (ms / n * 1e-3) / 1e12:.1f} TFLOP/s algorithmic)")

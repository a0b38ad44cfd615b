#!/usr/bin/env python
"""Is the config-3 kernel bound by the schedule or by the power cap?  The same launch (same instruction stream, same LDS and HBM
traffic) on operands that toggle fewer and fewer bits: random features (the bench's), a constant, zeros.  A schedule-bound
kernel takes the same time on all three; a power-bound one speeds up as the data quiets down.
The exp build's schedule variants run the same cases.
usage: python tools/simnn_power_check.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from densematcher_amd.engine import MatchEngine

from densematcher_amd import _build
VARIANTS = {"product (reads / DMA issued between the MFMAs)": {}, "reads / DMA in front of the MFMAs (r03 product)": {"DM_SIMNN_DEBUG": "320"},
            "five-slot ring, reads in front": {"DM_SIMNN_DEBUG": str(0x20000)}}
if len(sys.argv) < 2:
    import subprocess
    for name, env in VARIANTS.items():
        print("==", name, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=False)
    sys.exit(0)
eng = MatchEngine(0, lib_path=_build.LIB_EXP)
peaks = bench.measured_peaks(eng)
print("   bare MFMA loop: zero operands", peaks["mfma_f16_zero_operands_tflops"], "TFLOP/s, N(0,1) operands", peaks["mfma_f16_random_operands_tflops"], "TFLOP/s", flush=True)
w = bench.WORKLOADS["simnn"]
n, D, B = w["nu"] * w["nv"], w["D"], w["B"]
feats = bench.simnn_features(B, n, D, 0)
cases = {"bench features": (feats["F1"], feats["F2"]),
         "uniform random": tuple(np.random.default_rng(i).uniform(-1, 1, (B, n, D)).astype(np.float16) for i in range(2)),
         "constant 0.5": tuple(np.full((B, n, D), 0.5, np.float16) for _ in range(2)),
         "zeros": tuple(np.zeros((B, n, D), np.float16) for _ in range(2))}
flops = 2.0 * B * n * n * D
for name, (a, b) in cases.items():
    F1 = torch.as_tensor(a).to(eng.device)
    F2 = torch.as_tensor(b).to(eng.device)
    for _ in range(30):
        eng.simnn(F2, F1)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        eng.profile_kernel("simnn_f16_mfma")
        for _ in range(20):
            eng.simnn(F2, F1)
        nl, ms = eng.profile_read()
        ts.append(1e3 * ms / nl)
    eng.profile_kernel("")
    t = float(np.median(ts))
    print(f"{name:16s} kernel us {t:7.1f}   ({' '.join(f'{x:.1f}' for x in ts)})   {flops / t / 1e6:7.1f} TFLOP/s   frac of 2.5 PF {flops / t / 1e6 / 2500.0:.3f}", flush=True)

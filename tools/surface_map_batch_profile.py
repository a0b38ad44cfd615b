#!/usr/bin/env python
"""compute_surface_map_batch (64 raw pairs, notebook parameters) and compute_surface_map (one pair): per-kernel device time next to
the wall time of the call."""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd import functional_map as fmod, synth  # noqa: E402
from densematcher_amd.engine import default_engine  # noqa: E402
from densematcher_amd.pyFM.mesh import laplacian as _lap  # noqa: E402

_lap.set_robust_backend("restated")
w = bench.WORKLOADS["surface_map"]
nu, nv, D, k = w["nu"], w["nv"], w["D"], w["k"]
eng = default_engine()
NSTREAMS = int(os.environ.get("NSTREAMS", "1"))
for B in [int(a) for a in sys.argv[1:]] or [64, 1]:
    m1, m2, F1s, F2s = [], [], [], []
    for i in range(B):
        v1, f1 = synth.torus_mesh(nu, nv, perturb=0.03, seed=3 + 2 * i)
        v2, f2 = synth.torus_mesh(nu, nv, perturb=0.08, seed=4 + 2 * i)
        F1, F2, _ = synth.feature_pair(nu * nv, nu * nv, D, 1000 + i, 2000 + i, sigma=0.5, perm="identity")
        m1.append(bench._Duck(v1, f1)); m2.append(bench._Duck(v2, f2)); F1s.append(F1); F2s.append(F2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for rep in range(3):
            if rep == 2:
                eng.profile_kernel("*")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if B > 1:
                res = fmod.compute_surface_map_batch(m1, m2, F1s, F2s, n_ev=k, compute_extra=True, optimizer="L-BFGS-B", fit_params=dict(bench.NOTEBOOK_FIT), streams=NSTREAMS)
            else:
                res = fmod.compute_surface_map(m1[0], m2[0], F1s[0], F2s[0], n_ev=k, compute_extra=True, optimizer="L-BFGS-B", fit_params=dict(bench.NOTEBOOK_FIT))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if rep < 2:
                print(f"B = {B}: call {rep}: {1e3 * dt:.1f} ms", flush=True)
    rep_ = eng.profile_report(); eng.profile_kernel("")
    tot = sum(ms for _, ms in rep_.values())
    print(f"B = {B}: profiled call {1e3 * dt:.1f} ms wall, {tot:.1f} ms in {sum(n for n, _ in rep_.values())} bracketed launches")
    groups = {}
    for name, (n, ms) in rep_.items():
        g = name.split("_")[0]
        groups.setdefault(g, [0, 0.0])
        groups[g][0] += n; groups[g][1] += ms
    print("   by prefix:", ", ".join(f"{g} {v[1]:.1f} ms / {v[0]}" for g, v in sorted(groups.items(), key=lambda kv: -kv[1][1])[:10]))
    for name, (n, ms) in sorted(rep_.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"      {name:32s} {n:6d} x {1e3 * ms / n:9.2f} us = {ms:8.2f} ms")

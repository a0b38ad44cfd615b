#!/usr/bin/env python
"""cProfile of one compute_surface_map_batch call (64 raw pairs): where the host time goes."""
import cProfile
import os
import pstats
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd import functional_map as fmod, synth  # noqa: E402
from densematcher_amd.pyFM.mesh import laplacian as _lap  # noqa: E402

_lap.set_robust_backend("restated")
w = bench.WORKLOADS["surface_map"]
nu, nv, D, k = w["nu"], w["nv"], w["D"], w["k"]
B = 64
m1, m2, F1s, F2s = [], [], [], []
for i in range(B):
    v1, f1 = synth.torus_mesh(nu, nv, perturb=0.03, seed=3 + 2 * i)
    v2, f2 = synth.torus_mesh(nu, nv, perturb=0.08, seed=4 + 2 * i)
    F1, F2, _ = synth.feature_pair(nu * nv, nu * nv, D, 1000 + i, 2000 + i, sigma=0.5, perm="identity")
    m1.append(bench._Duck(v1, f1)); m2.append(bench._Duck(v2, f2)); F1s.append(F1); F2s.append(F2)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(2):
        fmod.compute_surface_map_batch(m1, m2, F1s, F2s, n_ev=k, compute_extra=True, optimizer="L-BFGS-B", fit_params=dict(bench.NOTEBOOK_FIT))
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    fmod.compute_surface_map_batch(m1, m2, F1s, F2s, n_ev=k, compute_extra=True, optimizer="L-BFGS-B", fit_params=dict(bench.NOTEBOOK_FIT))
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)

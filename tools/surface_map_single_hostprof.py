#!/usr/bin/env python
"""cProfile of one compute_surface_map call (one raw pair, notebook parameters): where the host time goes."""
import cProfile
import os
import pstats
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd import functional_map as fmod, synth  # noqa: E402
from densematcher_amd.pyFM.mesh import laplacian as _lap  # noqa: E402

_lap.set_robust_backend("restated")
w = bench.WORKLOADS["surface_map"]
nu, nv, D, k = w["nu"], w["nv"], w["D"], w["k"]
v1, f1 = synth.torus_mesh(nu, nv, perturb=0.03, seed=3)
v2, f2 = synth.torus_mesh(nu, nv, perturb=0.08, seed=4)
F1, F2, _ = synth.feature_pair(nu * nv, nu * nv, D, 1000, 2000, sigma=0.5, perm="identity")
m1, m2 = bench._Duck(v1, f1), bench._Duck(v2, f2)
call = lambda: fmod.compute_surface_map(m1, m2, F1, F2, n_ev=k, compute_extra=True, optimizer="L-BFGS-B", fit_params=dict(bench.NOTEBOOK_FIT))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); call(); torch.cuda.synchronize()
        print(f"call {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms")
    pr = cProfile.Profile()
    pr.enable(); call(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 40)

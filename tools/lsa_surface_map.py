#!/usr/bin/env python
"""The three assignment problems of one compute_surface_map call (notebook parameters): each matrix alone with the warm
start (lsa_reg=2) and in SciPy's order (lsa_reg=1), and the batched call."""
import os
import sys
import time
import warnings

import numpy as np
import scipy.optimize
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd import functional_map as fmod, synth  # noqa: E402
from densematcher_amd.engine import default_engine  # noqa: E402

from densematcher_amd.pyFM.mesh import laplacian as _lap
_lap.set_robust_backend("restated")
w = bench.WORKLOADS["surface_map"]
nu, nv, D, k = w["nu"], w["nv"], w["D"], w["k"]
(v1, f1), (v2, f2) = synth.torus_mesh(nu, nv, perturb=0.03, seed=3), synth.torus_mesh(nu, nv, perturb=0.08, seed=1)
F1, F2, _ = synth.feature_pair(nu * nv, nu * nv, D, 1000, 2000, sigma=0.5, perm="identity")
captured = []
orig = fmod._assign_many


def capture(matrices):
    devs = [m.device_tensor() if hasattr(m, "device_tensor") else m for m in matrices]
    captured[:] = [(d if d.dim() == 2 else d[0]).clone() for d in devs]
    return orig(matrices)


fmod._assign_many = capture
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    fmod.compute_surface_map(bench._Duck(v1, f1), bench._Duck(v2, f2), F1, F2, n_ev=k, compute_extra=True, optimizer="L-BFGS-B",
                             fit_params=dict(bench.NOTEBOOK_FIT))
eng = default_engine()


def run(d, mode):
    eng.set_option("lsa_reg", mode)
    eng.linear_sum_assignment(d, maximize=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = eng.linear_sum_assignment(d, maximize=True).cpu().numpy()
    return (time.perf_counter() - t0) * 1e3, got


for name, m in zip(("indicator", "precise map", "indicator after ICP"), captured):
    host = m.cpu().numpy()
    nz = float((host != 0).mean())
    t0 = time.perf_counter()
    ref = scipy.optimize.linear_sum_assignment(host, maximize=True)[1]
    t_cpu = (time.perf_counter() - t0) * 1e3
    t2, g2 = run(m[None], 2)
    t1, g1 = run(m[None], 1)
    print(f"{name:20s} nonzero {nz:.4f}  warm start {t2:7.1f} ms  SciPy's order {t1:7.1f} ms  SciPy on the host {t_cpu:7.1f} ms  "
          f"equal to SciPy: {np.array_equal(g2[0], ref)} {np.array_equal(g1[0], ref)}", flush=True)
t2, _ = run(torch.stack(captured), 2)
print(f"the three in one call: {t2:.1f} ms")

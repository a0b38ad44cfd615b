#!/usr/bin/env python
"""Per-kernel times of the iterative fit of the notebook's compute_surface_map call (device L-BFGS, one pair)."""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd import synth  # noqa: E402
from densematcher_amd.engine import default_engine  # noqa: E402
from densematcher_amd.pyFM.functional import FunctionalMapping  # noqa: E402
from densematcher_amd.pyFM.mesh import TriMesh  # noqa: E402

w = bench.WORKLOADS["surface_map"]
nu, nv, D, k = w["nu"], w["nv"], w["D"], w["k"]
(v1, f1), (v2, f2) = synth.torus_mesh(nu, nv, perturb=0.03, seed=3), synth.torus_mesh(nu, nv, perturb=0.08, seed=1)
F1, F2, _ = synth.feature_pair(nu * nv, nu * nv, D, 1000, 2000, sigma=0.5, perm="identity")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = FunctionalMapping(TriMesh(v1, f1), TriMesh(v2, f2))
    model.preprocess(n_ev=(k, k), descr_type="neural", descr1=F1, descr2=F2, verbose=False)
eng = default_engine()
fp = dict(bench.NOTEBOOK_FIT)
from densematcher_amd.pyFM import functional as _fn  # noqa: E402
if len(sys.argv) > 1:
    _fn.LBFGS_OPTIONS["ftol"] = float(sys.argv[1])
Cs = []
for rep in range(2):
    if rep == 1:
        eng.profile_kernel("*")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.fit(**fp, verbose=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
rep_ = eng.profile_report(); eng.profile_kernel("")
tot = sum(ms for _, ms in rep_.values())
print(f"ftol {_fn.LBFGS_OPTIONS['ftol']:g}  |C| max {np.abs(model.FM).max():.6f}  energy {float(model.fit_result.fun[0]):.12e}")
np.save("/tmp/fit_C_%s.npy" % (sys.argv[1] if len(sys.argv) > 1 else "default"), model.FM)
print(f"fit: {1e3 * dt:.1f} ms wall, {tot:.1f} ms of kernel time, evaluations {int(model.fit_result.nfev[0])}")
for name, (n, ms) in sorted(rep_.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{name:32s} {n:6d} x {1e3 * ms / n:8.2f} us = {ms:8.2f} ms")

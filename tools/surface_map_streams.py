#!/usr/bin/env python
"""compute_surface_map_batch (64 raw pairs, notebook parameters): wall time of a call against the number of chunk streams."""
import os
import sys
import time
import warnings

import numpy as np
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
ref = None
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for ns in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
        if os.environ.get("AB_EARLY"):            # A/B on the same box: the last chunk's early assignments off / on, alternating
            from densematcher_amd import functional_map as _f
            for trial in range(3):
                line = []
                for flag in (False, True):
                    _f.EARLY_ASSIGNMENTS = flag
                    tt = []
                    for rep in range(3):
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                        fmod.compute_surface_map_batch(m1, m2, F1s, F2s, n_ev=k, compute_extra=True, optimizer="L-BFGS-B", fit_params=dict(bench.NOTEBOOK_FIT), streams=ns)
                        torch.cuda.synchronize(); tt.append(time.perf_counter() - t0)
                    line.append(f"early={flag}: {[round(1e3 * t) for t in tt]}")
                print(f"streams = {ns} trial {trial}: " + "   ".join(line), flush=True)
            _f.EARLY_ASSIGNMENTS = True
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = fmod.compute_surface_map_batch(m1, m2, F1s, F2s, n_ev=k, compute_extra=True, optimizer="L-BFGS-B", fit_params=dict(bench.NOTEBOOK_FIT), streams=ns)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        sig = [np.concatenate([np.asarray(r[s]).ravel() for s in (0, 1, 4, 5, 10, 11, 12, 13)] + [r[2][1], r[3][1], r[6][1]]) for r in res]
        same = True if ref is None else all(np.array_equal(a, b) for a, b in zip(sig, ref))
        ref = sig if ref is None else ref
        print(f"streams = {ns}: calls {[round(1e3 * t) for t in ts]} ms -> {B / min(ts[1:]):.1f} pairs/s; every slot equal to the one-stream call: {same}", flush=True)

#!/usr/bin/env python
"""dm_eigenbasis on the 128 meshes of a 64-pair compute_surface_map_batch call (k = 20): time, rounds and accuracy against the
guard / degree / tolerance of the subspace iteration."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densematcher_amd import synth  # noqa: E402
from densematcher_amd.engine import default_engine  # noqa: E402

eng = default_engine()
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 128
meshes = [synth.torus_mesh(64, 32, perturb=0.03 if q % 2 == 0 else 0.08, seed=3 + q) for q in range(nm)]
covers = eng.tufted_covers(meshes)
ell = eng.laplacian_ell([c[0] for c in covers], lens=[c[1] for c in covers], verts=[v for v, _ in meshes], scale=0.5)
k = 20
ref = None
for guard, degree, tol in ((32, 30, 1e-10), (32, 30, 1e-8), (16, 30, 1e-10), (12, 30, 1e-10), (16, 40, 1e-10), (16, 20, 1e-10), (24, 30, 1e-10), (32, 40, 1e-10), (32, 20, 1e-10)):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lam, Phi, resid, rounds = eng.eigenbasis(None, None, k, guard=guard, degree=degree, tol=tol, ell=ell)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lam = lam.cpu().numpy()
    if ref is None:
        ref = lam
    print(f"guard {guard:2d} degree {degree:2d} tol {tol:g}: {1e3 * dt:7.1f} ms, rounds {rounds}, max resid / lam_k {float((resid / lam[:, -1].max()).max()) if False else float((resid.cpu().numpy() / lam[:, -1]).max()):.1e}, "
          f"max |lam - lam_ref| / lam_k {np.abs(lam - ref).max() / ref[:, -1].max():.1e}", flush=True)

#!/usr/bin/env python
"""Linear-assignment timing on the kind of matrix compute_surface_map produces (the mapped indicator of a k = 15 map,
N = 2048: rank 15, long augmenting paths) and on a random one, for every implementation (dm_set_option "lsa_reg")."""
import os
import sys
import time

import numpy as np
import scipy.linalg
import scipy.optimize
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densematcher_amd import synth  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402
from oracle import dm_oracle as orc  # noqa: E402

nu, nv, D, k = 64, 32, 512, 15
(v1, f1), (v2, f2) = synth.torus_mesh(nu, nv, perturb=0.03, seed=3), synth.torus_mesh(nu, nv, perturb=0.08, seed=1)
F1, F2, _ = synth.feature_pair(nu * nv, nu * nv, D, 1000, 2000, sigma=0.5, perm="identity")
bases = []
for v, f in ((v1, f1), (v2, f2)):
    W, m = synth.cotan_laplacian(v, f)
    lam, phi = scipy.linalg.eigh(W.toarray(), np.diag(m), subset_by_index=[0, 19])
    bases.append((lam[:k], phi[:, :k], m))
(l1, e1, m1), (l2, e2, m2) = bases
C = orc.fit(e1, e2, l1, l2, m1, m2, F1, F2, 1e4, 1e3)
M = orc.mapped_indicator(C, e1, e2, m1)
R = np.random.default_rng(0).standard_normal((2048, 2048))
eng = MatchEngine(0)
for name, mat in (("indicator", M), ("random", R)):
    t0 = time.perf_counter()
    ref = scipy.optimize.linear_sum_assignment(mat, maximize=True)
    t_cpu = time.perf_counter() - t0
    d = torch.as_tensor(mat[None]).to(eng.device)
    for mode in (2, 1, 0):
        eng.set_option("lsa_reg", mode)
        eng.linear_sum_assignment(d, maximize=True)
        torch.cuda.synchronize()
        eng.profile_kernel("*")
        t0 = time.perf_counter()
        got = eng.linear_sum_assignment(d, maximize=True).cpu().numpy()[0]
        dt = time.perf_counter() - t0
        rep = eng.profile_report()
        eng.profile_kernel("")
        print(f"{name:10s} lsa_reg={mode}: {dt * 1e3:9.1f} ms  equal to SciPy: {np.array_equal(got, ref[1])}   (SciPy on the host: {t_cpu * 1e3:.0f} ms)  "
              + " ".join(f"{n}={ms:.1f}ms" for n, (c, ms) in rep.items()), flush=True)

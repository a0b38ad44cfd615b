#!/usr/bin/env python
"""How long does each matrix of a 64-pair compute_surface_map_batch call take in the linear assignment?  (A launch lasts as long as
its slowest matrix.)  Every pair's three matrices alone, then all of them in one launch."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m1, m2, F1s, F2s = [], [], [], []
for i in range(B):
    v1, f1 = synth.torus_mesh(nu, nv, perturb=0.03, seed=3 + 2 * i)
    v2, f2 = synth.torus_mesh(nu, nv, perturb=0.08, seed=4 + 2 * i)
    F1, F2, _ = synth.feature_pair(nu * nv, nu * nv, D, 1000 + i, 2000 + i, sigma=0.5, perm="identity")
    m1.append(bench._Duck(v1, f1)); m2.append(bench._Duck(v2, f2)); F1s.append(F1); F2s.append(F2)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    res = fmod.compute_surface_map_batch(m1, m2, F1s, F2s, n_ev=k, compute_extra=True, optimizer="L-BFGS-B", fit_params=dict(bench.NOTEBOOK_FIT), streams=1)
eng = default_engine()
st = lambda f: np.stack([f(r[7]) for r in res])
P1, P2 = st(lambda m: m.mesh1.eigenvectors[:, :k]), st(lambda m: m.mesh2.eigenvectors[:, :k])
a1 = st(lambda m: m.mesh1.A.diagonal())
C0, Ci = st(lambda m: m._FM_base), st(lambda m: m._FM_icp)
faces = np.stack([r[7].mesh1.facelist for r in res]).astype(np.int32)
prec = eng.precise_map(P1, P2, C0, faces, dense=True)[2]


def timed(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


t_plain = [timed(lambda: eng.lsa_indicator(P1[b:b + 1], P2[b:b + 1], a1[b:b + 1], C0[b:b + 1])) for b in range(B)]
t_icp = [timed(lambda: eng.lsa_indicator(P1[b:b + 1], P2[b:b + 1], a1[b:b + 1], Ci[b:b + 1])) for b in range(B)]
t_prec = [timed(lambda: eng.linear_sum_assignment(prec[b:b + 1], maximize=True)) for b in range(B)]
for name, t in (("indicator of the fitted map", t_plain), ("indicator of the ICP map", t_icp), ("precise map (dense)", t_prec)):
    t = np.sort(t)
    print(f"{name:30s} alone: min {t[0]:6.1f}  median {t[len(t) // 2]:6.1f}  90 % {t[int(0.9 * len(t))]:6.1f}  max {t[-1]:6.1f} ms")
print("all indicators by factors + precise maps dense, one launch: %.1f ms" %
      timed(lambda: eng.lsa_indicator(np.concatenate([P1, P1]), np.concatenate([P2, P2]), np.concatenate([a1, a1]), np.concatenate([C0, Ci]), dense=prec)))
print("indicators only, one launch: %.1f ms" % timed(lambda: eng.lsa_indicator(np.concatenate([P1, P1]), np.concatenate([P2, P2]), np.concatenate([a1, a1]), np.concatenate([C0, Ci]))))
print("precise maps only, one launch: %.1f ms" % timed(lambda: eng.linear_sum_assignment(prec, maximize=True)))
if os.environ.get("LSA_DUMP"):
    nb = min(B, 8)
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez_compressed("gpurun_out/lsa_factors.npz", P1=P1[:nb], P2=P2[:nb], a1=a1[:nb], C0=C0[:nb], Ci=Ci[:nb])
if os.environ.get("LSA_OUTLIERS"):
    # the slowest ICP indicators: time in the warm-start mode and in SciPy's order
    order = np.argsort(t_icp)[::-1][:3]
    for b in order:
        line = f"pair {b}: ICP indicator {t_icp[b]:.1f} ms (fitted {t_plain[b]:.1f})"
        for mode in (2, 1):
            eng.set_option("lsa_reg", mode)
            line += f"   lsa_reg={mode}: {timed(lambda: eng.lsa_indicator(P1[b:b + 1], P2[b:b + 1], a1[b:b + 1], Ci[b:b + 1])):.1f} ms"
        eng.reset_options()
        s = np.linalg.svd(Ci[b], compute_uv=False)
        d = np.abs(Ci[b] - C0[b]).max()
        line += f"   |Ci - C0|max {d:.2f}, sigma(Ci) in [{s.min():.3f}, {s.max():.3f}], |diag Ci| mean {np.abs(np.diag(Ci[b])).mean():.2f} (fitted {np.abs(np.diag(C0[b])).mean():.2f})"
        print(line)

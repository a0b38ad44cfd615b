#!/usr/bin/env python
"""compute_surface_map_batch (64 pairs, two chunks): when does each chunk's thread leave its stages?  Host timestamps (ms after the call
started) at the return of the eigenbases, the fit (which ends with a device read), ICP and the assignments, per chunk thread, for several
calls -- to see what differs between a fast and a slow call."""
import os
import sys
import threading
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd import functional_map as fmod, synth  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402
from densematcher_amd.pyFM.mesh import TriMesh, laplacian as _lap  # noqa: E402

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
log, t_call = [], [0.0]
STREAMS = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--streams=")), None)     # (None: the call's default)
REPS = next((int(a) for a in sys.argv[1:] if a.isdigit()), 8)
print("streams =", STREAMS if STREAMS else "default", flush=True)


def stamp(label, fn):
    def wrapper(*a, **kw):
        t_in = 1e3 * (time.perf_counter() - t_call[0])
        out = fn(*a, **kw)
        log.append((threading.get_ident(), label, t_in, 1e3 * (time.perf_counter() - t_call[0])))
        return out
    return wrapper


pm = TriMesh.__dict__["process_many"].__func__
TriMesh.process_many = staticmethod(stamp("eigenbases", pm))
for name in ("fit_general", "icp", "precise_map", "lsa_indicator"):
    setattr(MatchEngine, name, stamp(name, getattr(MatchEngine, name)))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(REPS):
        log.clear()
        torch.cuda.synchronize(); t_call[0] = time.perf_counter()
        fmod.compute_surface_map_batch(m1, m2, F1s, F2s, n_ev=k, compute_extra=True, optimizer="L-BFGS-B", fit_params=dict(bench.NOTEBOOK_FIT), streams=STREAMS)
        torch.cuda.synchronize(); total = 1e3 * (time.perf_counter() - t_call[0])
        tids = sorted({t for t, *_ in log}, key=lambda t: min(x[2] for x in log if x[0] == t))
        line = f"call {rep}: {total:6.1f} ms |"
        for q, tid in enumerate(tids):
            line += f" chunk {q}:" + "".join(f" {lab[:4]} {a:.0f}-{b:.0f}" for t, lab, a, b in log if t == tid) + " |"
        print(line, flush=True)

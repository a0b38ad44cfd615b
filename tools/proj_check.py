#!/usr/bin/env python
"""The one-pass projection kernel (running scale per workgroup, csrc/dm_project.hip: proj_onepass_kernel) against the r03 pair
of launches it replaces (maxima pass + tile kernel on an fp32 copy; dm_set_option "proj_onepass" = 0) and against a float64
product, on the config-2 shape and on inputs that move the running scale (magnitudes growing / shrinking along the vertices).
usage: python tools/proj_check.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from densematcher_amd.engine import MatchEngine

from densematcher_amd import _build
eng = MatchEngine(0, lib_path=_build.LIB_EXP if os.environ.get('DM_PROJ_FLIP') is not None else None)
w = dict(bench.WORKLOADS["fmap"])
host = bench.make_batch(w, 0)
dev = {k_: torch.as_tensor(v).to(eng.device) for k_, v in host.items()}
k = w["k"]


def ref(Phi, a, F):
    return torch.einsum("bnk,bnd->bkd", Phi[:, :, :k].float().double() * a.float().double()[:, :, None], F.double())


def timed(fn, name="project_f16split_mfma", reps=10):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    out = {}
    for nm in (name, "project_absmax"):
        eng.profile_kernel(nm)
        for _ in range(reps):
            fn()
        nl, ms = eng.profile_read()
        out[nm] = 1e3 * ms / reps
    eng.profile_kernel("")
    return out


for dt in ("f64", "f32"):
    Phi, a = dev["Phi1"], dev["a1"]
    if dt == "f32":
        Phi, a = Phi.float(), a.float()
    R = ref(Phi, a, dev["F1"])
    sc = R.abs().amax(dim=(1, 2), keepdim=True)
    for opt in (1, 0):
        eng.set_option("proj_onepass", opt)
        A = eng.project(Phi, a, dev["F1"], k)
        err = ((A.double() - R).abs() / sc).max().item()
        ts = timed(lambda: eng.project(Phi, a, dev["F1"], k))
        print(f"basis {dt} proj_onepass={opt}: max |A - A_f64| / max|A| = {err:.2e}   us per call: " +
              "  ".join(f"{n} {v:.1f}" for n, v in ts.items()), flush=True)
eng.set_option("proj_onepass", 1)
# running scale: magnitudes that grow / shrink by 2^40 along the vertices, zero stretches, one huge entry at the end
rng = np.random.default_rng(0)
B, N, kk, D = 2, 2500, 77, 200
for name, prof in (("growing", np.exp2(np.linspace(-20, 20, N))), ("shrinking", np.exp2(np.linspace(20, -20, N))),
                   ("zeros then data", np.concatenate([np.zeros(1100), np.ones(N - 1100)])),
                   ("spike at the end", np.concatenate([np.ones(N - 1), [1e6]])), ("all zero", np.zeros(N))):
    Phi = rng.standard_normal((B, N, kk + 4)) * prof[None, :, None]      # odd row stride; what lies behind column k must not matter
    Phi[:, :, kk:] = np.nan
    Phi = torch.as_tensor(Phi).to(eng.device)
    a = torch.as_tensor(rng.uniform(0.5, 1.5, (B, N))).to(eng.device)
    F = torch.as_tensor(rng.standard_normal((B, N, D)).astype(np.float16)).to(eng.device)
    R = torch.einsum("bnk,bnd->bkd", Phi[:, :, :kk].float().double() * a.float().double()[:, :, None], F.double())
    # error scale: sum_n |X| |F| per output (what an fp32-class product can promise)
    S = torch.einsum("bnk,bnd->bkd", (Phi[:, :, :kk].float().double() * a.float().double()[:, :, None]).abs(), F.double().abs()).clamp_min(1e-300)
    for opt in (1, 0):
        eng.set_option("proj_onepass", opt)
        A = eng.project(Phi, a, F, kk)
        print(f"{name:18s} proj_onepass={opt}: max |A - A_f64| / sum|x||f| = {((A.double() - R).abs() / S).max().item():.2e}  finite {bool(torch.isfinite(A).all())}", flush=True)

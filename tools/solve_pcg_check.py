#!/usr/bin/env python
"""r06: the batched conjugate-gradient solver (dm_set_option solve_pcg, csrc/dm_pcg.h) against the direct register-resident solver on the
bench's config-2 batch and on harder descriptor families: max |C_pcg - C_direct|, the kernels' times, pairs sent to the fall-back.
usage: python tools/solve_pcg_check.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd import synth  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402

eng = MatchEngine(0)
w = dict(bench.WORKLOADS["fmap"])
host = bench.make_batch(w, 0, "f64")
B, n, D = host["F1"].shape
k = w["k"]


def run(name, F1, F2, w_descr=1e4, w_lap=1e3):
    dev = {q: torch.as_tensor(v).to(eng.device) for q, v in host.items()}
    dev["F1"], dev["F2"] = torch.as_tensor(F1).to(eng.device), torch.as_tensor(F2).to(eng.device)
    out = {}
    for pcg in (0, 1):
        eng.set_option("solve_pcg", pcg)
        for _ in range(3):
            C = eng.fmap_fit(dev["Phi1"], dev["Phi2"], dev["a1"], dev["a2"], dev["F1"], dev["F2"], dev["lam1"][:, :k].contiguous(), dev["lam2"][:, :k].contiguous(), w_descr, w_lap, k1=k, k2=k)
        torch.cuda.synchronize()
        eng.profile_kernel("*")
        for _ in range(5):
            C = eng.fmap_fit(dev["Phi1"], dev["Phi2"], dev["a1"], dev["a2"], dev["F1"], dev["F2"], dev["lam1"][:, :k].contiguous(), dev["lam2"][:, :k].contiguous(), w_descr, w_lap, k1=k, k2=k)
        rep = eng.profile_report()
        eng.profile_kernel("")
        out[pcg] = (C.cpu().numpy(), {q: 1e3 * ms / cnt for q, (cnt, ms) in rep.items() if q.startswith("fmap_solve")})
    d = np.abs(out[1][0] - out[0][0]).max()
    print(f"{name:28s} max |C_pcg - C_direct| = {d:.2e} (max |C| = {np.abs(out[0][0]).max():.2e});  direct {out[0][1]}  pcg {out[1][1]}", flush=True)


run("sigma 0.1 (headline)", host["F1"], host["F2"])
F1 = np.empty_like(host["F1"]); F2 = np.empty_like(host["F2"])
for i in range(B):
    F1[i], F2[i], _ = synth.feature_pair(n, n, D, 1000 + i, 2000 + i, sigma=1.0, perm="identity")
run("sigma 1.0", F1, F2)
for i in range(B):
    F1[i], F2[i] = synth.smooth_feature_pair(host["Phi1"][i].astype(np.float64), host["Phi2"][i].astype(np.float64), D, 1000 + i, 2000 + i)
run("smooth", F1, F2)
run("smooth, w_lap = 0", F1, F2, 1e4, 0.0)
run("sigma 1.0, w_lap = 1e-3 w_descr = 1e-1 (API defaults)", F1 * 0 + host["F1"], host["F2"], 1e-1, 1e-3)
# rank-deficient descriptors: 40 distinct channels repeated -> P singular up to the Laplacian term
Fr1 = np.tile(host["F1"][:, :, :40], (1, 1, D // 40 + 1))[:, :, :D].copy(); Fr2 = np.tile(host["F2"][:, :, :40], (1, 1, D // 40 + 1))[:, :, :D].copy()
run("rank-40 descriptors", Fr1, Fr2)
eng.set_option("solve_pcg", 1)

# config 5's size: n = 199 (the streamed-fragment kernel), 16 pairs
print("== k = 200, N = 8192 (16 pairs)")
w5 = dict(bench.WORKLOADS["stress"]); w5["B"] = 16
host = bench.make_batch(w5, 0, "f64")
B, n, D = host["F1"].shape
k = w5["k"]
run("config-5 inputs", host["F1"], host["F2"])
F1 = np.empty_like(host["F1"]); F2 = np.empty_like(host["F2"])
for i in range(B):
    F1[i], F2[i] = synth.smooth_feature_pair(host["Phi1"][i].astype(np.float64), host["Phi2"][i].astype(np.float64), D, 1000 + i, 2000 + i)
run("smooth", F1, F2)
k = 150                                     # n = 149: the blocked LDS solver's range
run("smooth, k = 150", F1, F2)

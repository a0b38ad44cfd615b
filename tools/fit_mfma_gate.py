#!/usr/bin/env python
"""r06: the fp32 element loop of the fused fit with its two 16-deep products on v_mfma_f32_16x16x4_f32 (dm_set_option("fit_mfma", 1), maps
up to 16 x 16) against the packed vector FMA (0): one evaluation against the oracle, the fit under the reference's stopping rule, batch
invariance, time per evaluation at 64 pairs.  usage: python tools/fit_mfma_gate.py"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import dm_oracle as orc  # noqa: E402
from densematcher_amd.engine import default_engine  # noqa: E402
import test_gpu_fitfuse as tf  # noqa: E402

eng = default_engine()
NB = tf.NOTEBOOK_W
fx = dict(np.load(os.path.join(REPO, "tests", "golden", "fx_cfg1.npz")))

print("== one evaluation (fp32 element loop) against the oracle (relative to |E|, max |G|)")
for (N1, N2, k1, k2) in [(300, 517, 15, 13), (1000, 777, 16, 16), (129, 65, 7, 9), (2048, 2048, 15, 15), (200, 4100, 12, 16)]:
    rng = np.random.default_rng(N1 + 3 * N2 + k1)
    e1, e2, a1, C, A, Bm, lam1, lam2 = tf._random_problem(rng, 2, N1, N2, k1, k2)
    C[1] *= 40.0
    for w in ({"w_ent": 0.3}, dict(NB), {"w_p2p": 0.5, "w_ent": 0.3, "w_range01": 1.5, "w_sumto1": 2.0, "w_descr": 1.0, "w_lap": 0.1}):
        row = []
        for mf in (0, 1):
            eng.set_option("fit_mfma", mf)
            E, G = eng.energy_grad_fused(C, A, Bm, lam1, lam2, w, e1, e2, a1, precision="f32")
            worst_e = worst_g = 0.0
            for b in range(2):
                ev = orc.ev_sqdiff(lam1[b], lam2[b])
                Eo, Go = orc.energy_grad_general(C[b], A[b].astype(np.float64), Bm[b].astype(np.float64), ev, e1[b], e2[b], a1[b], w)
                worst_e = max(worst_e, abs(float(E[b]) - Eo) / abs(Eo))
                worst_g = max(worst_g, np.abs(G[b].cpu().numpy() - Go).max() / np.abs(Go).max())
            row.append((worst_e, worst_g))
        print(f"N1={N1:5d} N2={N2:5d} k=({k1},{k2}) terms={sorted(w)}: vector dE {row[0][0]:.1e} dG {row[0][1]:.1e} | matrix dE {row[1][0]:.1e} dG {row[1][1]:.1e}", flush=True)

print("== the notebook's fit of the fixture (k = 15, the reference's stopping rule: fp32 loop)")
k = 15
x0 = orc.get_x0(k, k, float(fx["Phi1"][0, 0]), float(fx["Phi2"][0, 0]), float(fx["a1"].astype(np.float64).sum()), float(fx["a2"].astype(np.float64).sum()))
one = {n: v[:1] for n, v in tf._fit_batch(fx, k, 1, None).items()}
res = {}
for mf in (0, 1):
    eng.set_option("fit_mfma", mf)
    C1, r1 = eng.fit_general(one, NB, x0[None])
    res[mf] = C1[0]
    print(f"fit_mfma={mf}: nit {int(r1.nit[0])} nfev {int(r1.nfev[0])} status {int(r1.status[0])} loop {r1.element_loop}", flush=True)
print("|C_matrix - C_vector| =", np.abs(res[1] - res[0]).max())

print("== batch invariance (matrix form): a pair in batches of 1, 3, 140")
eng.set_option("fit_mfma", 1)
b3 = tf._fit_batch(fx, k, 3, None)
C3, r3 = eng.fit_general(b3, NB, np.stack([x0] * 3))
C1, r1 = eng.fit_general({n: v[:1] for n, v in b3.items()}, NB, x0[None])
big = tf._fit_batch(fx, k, 140, None)
Cb, rb = eng.fit_general(big, NB, np.stack([x0] * 140))
print("1 vs 3:", np.array_equal(C1[0], C3[0]), " 140 vs 3:", all(np.array_equal(Cb[b], C3[b % 3]) for b in (0, 1, 2, 137, 139)))

print("== time per evaluation, 64 pairs of N = 2048, k = 15 (the batched documented call's fit)")
from densematcher_amd import synth  # noqa: E402
B = 64
batch = synth.make_pair_batch(B, 64, 32, 512, 15, sigma=0.3, n_distinct_meshes=2)
dev = {n: torch.as_tensor(v).to(eng.device) for n, v in batch.items()}
x0b = np.zeros((B, 15, 15)); x0b[:, 0, 0] = 1.0
for mf in (0, 1, 0, 1):
    eng.set_option("fit_mfma", mf)
    torch.cuda.synchronize()
    eng.profile_kernel("fit_fused_eval")
    t0 = time.time()
    Cf, rf = eng.fit_general(dev, NB, x0b, k=15)
    torch.cuda.synchronize()
    dt = time.time() - t0
    n, ms = eng.profile_read()
    eng.profile_kernel("")
    print(f"fit_mfma={mf}: fit {1e3 * dt:7.1f} ms, {n} launches, {1e3 * ms / max(n, 1):7.1f} us per launch (mean), nfev max {int(rf.nfev.max())}, loop {rf.element_loop}", flush=True)
eng.set_option("fit_mfma", 1)

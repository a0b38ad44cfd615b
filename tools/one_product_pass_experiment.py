#!/usr/bin/env python
"""would a ONE-product first pass (high fp16 halves only, K = k instead of 3k) of the four-map kernel pay?
Its scores carry an error of 2^-10 |t||s| (each operand's dropped low half is <= 2^-11 of it) instead of ~5.5e-5: the experiments
build adds that to the bound (DM_TAU_EXTRA_LOG2=10), so the merge queues exactly the rows such a pass would have to repair, and the
exact kernel's time at that queue size is measured.  Kill criterion: the repair's growth exceeds what the main loop could save
(<= half of it: the stage time is set by the delivery of the operands, which halves; profiles/r04_ubench_lds.txt) on ANY of
sigma = 0.1 / 1.0 / smooth.   usage: DM_TAU_EXTRA_LOG2=10 python tools/one_product_pass_experiment.py   (and without the variable)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd import _build, synth  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402

eng = MatchEngine(0, lib_path=_build.LIB_EXP)
w = dict(bench.WORKLOADS["fmap"])
host = bench.make_batch(w, 0, "f64")
dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
B, n, D = host["F1"].shape
k = w["k"]
print("DM_TAU_EXTRA_LOG2 =", os.environ.get("DM_TAU_EXTRA_LOG2", "(unset: the product's bound)"))


def variant(name, F1, F2):
    d = dict(dev)
    d["F1"] = torch.as_tensor(F1).to(eng.device)
    d["F2"] = torch.as_tensor(F2).to(eng.device)
    for _ in range(4):
        eng.match(d, k=k)
    torch.cuda.synchronize()
    eng.profile_kernel("*")
    for _ in range(6):
        eng.match(d, k=k)
    rep = eng.profile_report()
    eng.profile_kernel("")
    eng.match(d, k=k)
    rows = eng.last_requeued_rows()
    tot = sum(ms for _, ms in rep.values()) / 6
    g = lambda nm: (1e3 * rep[nm][1] / rep[nm][0]) if nm in rep else float("nan")
    print(f"{name:10s} step {tot:7.4f} ms  tile pass {g('simnn4_f16_mfma'):7.1f} us  merge {g('simnn_merge'):6.1f} us  exact {g('fm_split_exact_f64'):8.1f} us  "
          f"requeued rows (knn21, ind21, knn12, ind12): " + ", ".join(f"{r / (B * n):.4f}" for r in rows), flush=True)


variant("sigma 0.1", host["F1"], host["F2"])
F1 = np.empty_like(host["F1"]); F2 = np.empty_like(host["F2"])
for i in range(B):
    F1[i], F2[i], _ = synth.feature_pair(n, n, D, 1000 + i, 2000 + i, sigma=1.0, perm="identity")
variant("sigma 1.0", F1, F2)
for i in range(B):
    F1[i], F2[i] = synth.smooth_feature_pair(host["Phi1"][i].astype(np.float64), host["Phi2"][i].astype(np.float64), D, 1000 + i, 2000 + i)
variant("smooth", F1, F2)

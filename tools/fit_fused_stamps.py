#!/usr/bin/env python
"""Where a launch of the fused fit spends its time (experiments build, DM_FF_DEBUG=1: time stamps of pair 0's chain)."""
import os
import sys

import numpy as np
import torch

os.environ.setdefault("DM_FF_DEBUG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densematcher_amd import _build, synth  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402

eng = MatchEngine(0, lib_path=_build.LIB_EXP)
W = dict(w_descr=1e4, w_lap=1e3, w_ent=1e-1, w_sumto1=1e1)
k = 15
for B in [int(a) for a in sys.argv[1:]] or [1, 64]:
    host = synth.make_pair_batch(B, 64, 32, 512, k, sigma=0.5, n_distinct_meshes=min(B, 2))
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    x0 = np.zeros((B, k, k))
    x0[:, 0, 0] = 1.0
    for rep in range(2):
        C, res = eng.fit_general(dev, W, x0)
    print("B", B, "evaluations", res.nfev.max(), flush=True)

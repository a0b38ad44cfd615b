#!/usr/bin/env python
"""Ablations of the fused fit's evaluation kernel at full load (experiments build; DM_FF_MODE variants give WRONG results):
average launch time over the first iterations of a 64-pair fit."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import torch
    sys.path.insert(0, REPO)
    from densematcher_amd import _build, synth
    from densematcher_amd.engine import MatchEngine
    eng = MatchEngine(0, lib_path=_build.LIB_EXP)
    eng.set_option("fit_mfma", int(os.environ.get("DM_FIT_MFMA", "0")))      # (the element loop's products on the fp32 matrix instruction)
    B, k = 64, 15
    host = synth.make_pair_batch(B, 64, 32, 64, k, sigma=0.5, n_distinct_meshes=1, basis="random")
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    x0 = np.zeros((B, k, k)); x0[:, 0, 0] = 1.0
    W = dict(w_descr=1e4, w_lap=1e3, w_ent=1e-1, w_sumto1=1e1)
    for rep in range(2):
        if rep == 1:
            eng.profile_kernel("fit_fused_eval")
        try:
            C, res = eng.fit_general(dev, W, x0, maxiter=40)
        except Exception as e:
            print("  (fit ended with", type(e).__name__, ")")
    nl, ms = eng.profile_read()
    ne = int(res.nfev.max())
    print(f"  fit_mfma={os.environ.get('DM_FIT_MFMA', '0')} DM_FF_MODE={os.environ.get('DM_FF_MODE', '0')}: {nl} launches, evaluations min {int(res.nfev.min())} max {ne}; {1e3 * ms / ne:.1f} us per evaluation "
          f"(empty launches behind the last evaluation included: a few us each)", flush=True)
else:
    for mode, label in ((0, "product"), (1, "no unit epilogue"), (2, "every column reads Psi row 0 (scalar-cache hits)"), (4, "no element-wise terms (products only)"),
                        (3, "no epilogue + cache hits"), (7, "products only, no epilogue, cache hits")):
        print(label, flush=True)
        subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, DM_FF_MODE=str(mode)))

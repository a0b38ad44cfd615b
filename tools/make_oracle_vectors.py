#!/usr/bin/env python
"""
Vectors produced by the ORACLE (oracle/dm_oracle.py), not by the reference: expensive float64 minimisations that the
GPU tests compare against, committed so that the GPU box does not spend minutes of NumPy on them.  Each is also
bracketed by a reference-generated vector in the tests (the reference's own fp32 fit() output of the same problem,
tests/golden/fx_cfg1_terms.npz), so the oracle stays pinned to the reference.

    tests/golden/oracle_cfg1_fits.npz   float64 L-BFGS-B minimisers (tight tolerances) of the config-1 pair for
                                        the notebook's fit_params (w_ent = 0.1, w_sumto1 = 10) and for a mix of every
                                        implemented term
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import dm_oracle as orc  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")
MIX = dict(w_descr=1e4, w_lap=1e3, w_dcomm=0.5, w_p2p=0.05, w_stochastic=0.02, w_ent=0.1, w_range01=1.0, w_sumto1=2.0)
NOTEBOOK = dict(w_descr=1e4, w_lap=1e3, w_ent=1e-1, w_sumto1=1e1)


def main():
    fx = dict(np.load(os.path.join(GOLDEN, "fx_cfg1.npz")))
    k = int(fx["k"])
    args = (fx["Phi1"][:, :k], fx["Phi2"][:, :k], fx["lam1"][:k], fx["lam2"][:k], fx["a1"], fx["a2"])
    out = {}
    C, res = orc.fit_general(*args, fx["F1"], fx["F2"], NOTEBOOK)
    print("notebook:", res.nit, res.nfev, res.message)
    out["C_nb"] = C
    nd = 12                                                    # a dozen descriptors keep the operator list small
    C, res = orc.fit_general(*args, fx["F1"][:, :nd], fx["F2"][:, :nd], MIX)
    print("mix:", res.nit, res.nfev, res.message)
    out["C_mix"], out["mix_ndescr"] = C, nd
    np.savez_compressed(os.path.join(GOLDEN, "oracle_cfg1_fits.npz"), **out)


if __name__ == "__main__":
    main()

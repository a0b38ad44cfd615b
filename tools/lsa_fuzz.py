#!/usr/bin/env python
"""Fuzz of dm_linear_sum_assignment against scipy.optimize.linear_sum_assignment: random sizes (square and rectangular), real,
integer (ties), sparse, constant-column and duplicated-entry matrices, both senses, every implementation (lsa_reg 2 / 1)."""
import os
import sys

import numpy as np
import scipy.optimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densematcher_amd.engine import MatchEngine  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 700        # (above 4096 columns: the 1024-thread shape)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
eng = MatchEngine(0)
bad = 0
for case in range(n_cases):
    nr = int(rng.integers(max(1, max_n // 3 if max_n > 700 else 1), max_n))
    nc = nr if rng.random() < 0.6 else int(rng.integers(1, max_n))
    kind = int(rng.integers(0, 6))
    c = rng.standard_normal((nr, nc))
    if kind == 1:
        c = np.round(rng.uniform(1, 6) * c)
    elif kind == 2:
        c = c * (rng.random((nr, nc)) < rng.uniform(0.002, 0.2))
    elif kind == 3 and nc >= 2:
        cols = rng.choice(nc, size=min(nc, int(rng.integers(2, 6))), replace=False)
        c[:, cols] = rng.standard_normal()
    elif kind == 4:
        k = int(rng.integers(1, 40))
        c[rng.integers(0, nr, k), rng.integers(0, nc, k)] = c[0, 0]
    elif kind == 5:
        r = int(rng.integers(1, 8))
        c = rng.standard_normal((nr, r)) @ rng.standard_normal((r, nc))      # low rank: long searches
    for mx in (False, True):
        r0, c0 = scipy.optimize.linear_sum_assignment(c, maximize=mx)
        for mode in (2, 1):
            eng.set_option("lsa_reg", mode)
            got = eng.linear_sum_assignment(c[None], maximize=mx).cpu().numpy()[0]
            rows = np.nonzero(got >= 0)[0]
            if not (np.array_equal(rows, r0) and np.array_equal(got[rows], c0)):
                bad += 1
                print("MISMATCH", case, nr, nc, kind, mx, mode, flush=True)
print(f"{n_cases} cases x 2 senses x 2 implementations: {bad} mismatches")

#!/usr/bin/env python
"""Experiment: per-phase cycle counts of the blocked Cholesky solve (one workgroup), GPU box only.
Builds a -DDM_SOLVE_TIMING copy of the library next to the real one."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from densematcher_amd import _build, _lib  # noqa: E402

dbg_lib = os.path.join(REPO, "densematcher_amd", "libdensematch_timing.so")
srcs = [os.path.join(_build.CSRC, s) for s in _build.SOURCES]
subprocess.check_call([_build._hipcc(), *_build.FLAGS, "-DDM_SOLVE_TIMING", "-shared", *srcs, "-o", dbg_lib])
_lib.LIB_PATH = dbg_lib
from densematcher_amd.engine import MatchEngine  # noqa: E402

eng = MatchEngine(0)
eng.lib.dm_debug_solve_timing.restype = C.c_int
eng.lib.dm_debug_solve_timing.argtypes = [C.c_void_p]
k, D, B = 128, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
A = torch.tensor(rng.standard_normal((B, k, D)).astype(np.float32) * 0.05, device="cuda")
Bm = torch.tensor(rng.standard_normal((B, k, D)).astype(np.float32) * 0.05, device="cuda")
lam = torch.tensor(np.sort(rng.uniform(0, 100, (B, k)), axis=1), device="cuda")
c00 = torch.ones(B, dtype=torch.float64, device="cuda")
reps = 5
for _ in range(reps):
    eng.fmap_solve(A, Bm, lam, lam, c00, 1e4, 1e3)
torch.cuda.synchronize()
out = (C.c_longlong * 16)()
assert eng.lib.dm_debug_solve_timing(out) == 0
names = ["block load", "(a) diag block (wave 0) + barrier", "(b) panel MFMA + barrier", "whole solve (a+b+c+back substitution)",
         "store", "prologue (eigenvalue scale)", "penalty staging", "(c) trailing MFMA + rhs + barrier", "block load: global loads"]
tot = out[0] + out[3] + out[4] + out[5] + out[6] + out[8]
for n_, v in zip(names, out[:9]):
    print(f"{n_:28s} {v / reps:12.0f} cycles/solve  {100.0 * v / tot:5.1f} %")
print("total", tot / reps)

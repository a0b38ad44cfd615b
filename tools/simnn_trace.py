#!/usr/bin/env python
"""Stage timeline of the config-3 similarity kernel: s_memtime stamps of every stage of workgroup 0's third
tile, all eight waves (libdensematch_exp.so, DM_SIMNN_DEBUG=0x10000: the product instruction stream + seven stamps per stage, no
wait behind a stamp; the log goes through LDS and is copied out once per tile).

Stamps per stage (shader cycles):  t0 stage start | t1 after the six fragment reads of k-step 1 + first half of the stage's LDS-DMA
are ISSUED | t2 after the eight MFMAs of k-step 0 are issued | t3 after s_waitcnt vmcnt(n) lgkmcnt(0) | t4 after s_barrier |
t5 after the next stage's first fragment reads + second DMA half are issued | t6 after the eight MFMAs of k-step 1 are issued |
(t0 of the next stage = after the closing lgkmcnt(0)).
usage: python tools/simnn_trace.py > profiles/r04_simnn_stage_timeline.txt"""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ["DM_SIMNN_DEBUG"] = str(0x10000)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd import _build  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402

eng = MatchEngine(0, lib_path=_build.LIB_EXP)
w = bench.WORKLOADS["simnn"]
n, D, B = w["nu"] * w["nv"], w["D"], w["B"]
feats = bench.simnn_features(B, n, D, 0)
F1 = torch.as_tensor(feats["F1"]).to(eng.device)
F2 = torch.as_tensor(feats["F2"]).to(eng.device)
for _ in range(30):
    eng.simnn(F2, F1)
torch.cuda.synchronize()
eng.profile_kernel("simnn_f16_mfma")
for _ in range(10):
    eng.simnn(F2, F1)
nl, ms = eng.profile_read()
eng.profile_kernel("")
buf = (C.c_ulonglong * (8 * 256))()
eng.lib.dm_debug_simnn_trace.restype = C.c_int
eng.lib.dm_debug_simnn_trace.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
rc = eng.lib.dm_debug_simnn_trace(eng.ctx, buf)
assert rc == 0, rc
t = np.frombuffer(buf, dtype=np.uint64).reshape(8, 256).astype(np.int64)
print(f"# simnn_f16_mfma with stamps: {1e3 * ms / nl:.1f} us per launch (product kernel: see profiles/r04_simnn_bench.json)")
print(f"# config 3: 64 pairs, N = 2048, D = 768: 24 stages of 32 halves per 256 x 256 tile, 16 MFMA 32x32x16 per wave and stage")
print("# (= 512 cycles of matrix pipe per wave, 1024 per SIMD with its two waves); workgroup 0, third tile, waves 0..7")
nst = 24
ev = t[:, :8 * nst].reshape(8, nst, 8)[:, :, :7]
t00 = ev[:, 0, 0].min()
end = t[:, 8 * nst]                      # stamp after the main loop (before the epilogue)
print("\n## per-stage durations, cycles (median over stages 3..22), per wave")
names = ["issue reads(k1)+DMA a", "8 MFMA (k0)", "wait vmcnt/lgkm", "s_barrier", "issue reads(k0')+DMA b", "8 MFMA (k1)", "closing lgkmcnt(0)"]
print("wave  " + "  ".join(f"{nm:>22s}" for nm in names) + "      stage")
for wv in range(8):
    d = np.empty((nst - 1, 7))
    for s_ in range(nst - 1):
        e = ev[wv, s_]
        nxt = ev[wv, s_ + 1, 0]
        d[s_] = [e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3], e[5] - e[4], e[6] - e[5], nxt - e[6]]
    med = np.median(d[3:22], axis=0)
    print(f"{wv:4d}  " + "  ".join(f"{x:22.0f}" for x in med) + f"  {med.sum():9.0f}")
print("\n## absolute timeline of stages 10..12, cycles since the tile's first stamp (t0 t1 t2 t3 t4 t5 t6)")
for s_ in (10, 11, 12):
    for wv in range(8):
        print(f"stage {s_:2d} wave {wv}: " + " ".join(f"{x - t00:7d}" for x in ev[wv, s_]))
print("\n## main loop of the tile (first stamp -> stamp after the last stage), cycles: " + " ".join(str(int(e - ev[w_, 0, 0])) for w_, e in enumerate(end)))
print("## barrier skew per stage (max - min of t3 over the 8 waves), stages 3..22: median %.0f, max %.0f cycles" %
      (np.median(ev[:, 3:22, 3].max(0) - ev[:, 3:22, 3].min(0)), (ev[:, 3:22, 3].max(0) - ev[:, 3:22, 3].min(0)).max()))

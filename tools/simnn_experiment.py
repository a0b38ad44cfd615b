#!/usr/bin/env python
"""Experiment: time the similarity kernel with parts disabled (DM_SIMNN_DEBUG)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densematcher_amd.engine import MatchEngine
eng = MatchEngine(0)
B, N, D = 64, 2048, 768
g = torch.Generator(device="cuda").manual_seed(0)
S = torch.randn(B, N, D, device="cuda", generator=g).to(torch.float16)
T = torch.randn(B, N, D, device="cuda", generator=g).to(torch.float16)
configs = [("0", "0", "0"), ("0", "1", "0"), ("1", "0", "0"), ("1", "1", "0")]
res = {c: [] for c in configs}
for rep in range(4):
    for c in configs:
        mode, ph, ex = c
        os.environ["DM_SIMNN_DEBUG"] = mode
        os.environ["DM_SIMNN_PIPE"] = ph
        os.environ["DM_SIMNN_EXP"] = ex
        nn = eng.simnn(T, S)
        if mode == "0" and rep == 0:
            ref = (T[3].double() @ S[3].double().T).argmax(dim=1)
            assert torch.equal(nn[3].long(), ref), "wrong result"
        eng.profile_kernel("simnn_f16_mfma")
        for _ in range(10):
            eng.simnn(T, S)
        n, ms = eng.profile_read()
        eng.profile_kernel("")
        res[c].append(ms / n * 1e3)
for c, v in res.items():
    print(f"DM_SIMNN_DEBUG={c[0]} PIPE={c[1]} EXP={c[2]}: simnn_f16_mfma avg us per round: " + " ".join(f"{x:.1f}" for x in v)
          + f"   best {2.0 * N * N * D * B / (min(v) * 1e-6) / 1e12:.0f} TFLOP/s")

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
for mode, ph in [("0", "0"), ("1", "0"), ("0", "1"), ("1", "1"), ("3", "1")]:
    os.environ["DM_SIMNN_DEBUG"] = mode
    os.environ["DM_SIMNN_PHASED"] = ph
    for _ in range(2):
        nn = eng.simnn(T, S)
    if mode == "0":
        ref = (T[3].double() @ S[3].double().T).argmax(dim=1)
        assert torch.equal(nn[3].long(), ref), "wrong result"
    eng.profile_kernel("simnn_f16_mfma")
    for _ in range(5):
        eng.simnn(T, S)
    n, ms = eng.profile_read()
    eng.profile_kernel("")
    print(f"DM_SIMNN_DEBUG={mode} PHASED={ph}: simnn_f16_mfma avg {ms / n * 1e3:.1f} us  ({2.0 * N * N * D * B / (ms / n * 1e-3) / 1e12:.0f} TFLOP/s algorithmic)")

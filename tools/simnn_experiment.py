#!/usr/bin/env python
"""Experiment (needs the -DDM_EXPERIMENTS build, tools only): time the similarity kernel of config 3 with parts
disabled (DM_SIMNN_DEBUG = variant bits of simnn_pipe_kernel: 1 no epilogue, 7 LDS-DMA only, 8 no LDS-DMA, 16 / 32 K stagger by one
stage / by ns / tilesS, 64 fragment reads pinned first, 128 second wave of each SIMD runs its MFMAs first) and with one
workgroup per tile instead of the persistent walk.  Interleaved rounds in one process; writes one JSON line."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from densematcher_amd import _build
from densematcher_amd.engine import MatchEngine

eng = MatchEngine(0, lib_path=_build.LIB_EXP)
B, N, D = 64, 2048, int(os.environ.get("SIMNN_EXP_D", "768"))
g = torch.Generator(device="cuda").manual_seed(0)
S = torch.randn(B, N, D, device="cuda", generator=g)
T = S[:, torch.randperm(N, device="cuda", generator=g)] + torch.randn(B, N, D, device="cuda", generator=g)
S = (S / S.norm(dim=2, keepdim=True)).to(torch.float16)
T = (T / T.norm(dim=2, keepdim=True)).to(torch.float16)
# (DM_SIMNN_DEBUG, simnn_persist)
W8, W4 = 256, 512     # + 256: the 8-wave 256 x 256 shape (one workgroup per CU); + 512: 4 waves, 128 x 256, two per CU
configs = [(W8 + 64, 1), (W8 + 1024 + 64, 1), (W8 + 64 + 1, 1), (W8 + 1024 + 64 + 1, 1), (W8 + 64 + 9, 1), (W8 + 1024 + 64 + 9, 1)] if len(sys.argv) < 2 else \
    [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
res = {c: [] for c in configs}
ref = (T[3].double() @ S[3].double().T).argmax(dim=1)
for rep in range(3):
    for c in configs:
        os.environ["DM_SIMNN_DEBUG"] = str(c[0])
        eng.set_option("simnn_persist", c[1])
        nn = eng.simnn(T, S)
        if (c[0] & 15) == 0 and rep == 0 and c[0] >= 256:
            assert torch.equal(nn[3].long(), ref), f"wrong result in config {c}"
        eng.profile_kernel("simnn_f16_mfma")
        for _ in range(10):
            eng.simnn(T, S)
        n, ms = eng.profile_read()
        eng.profile_kernel("")
        res[c].append(ms / n * 1e3)
out = {}
for c, v in res.items():
    tf = 2.0 * N * N * D * B / (min(v) * 1e-6) / 1e12
    print(f"DM_SIMNN_DEBUG={c[0]:2d} persist={c[1]}: simnn_f16_mfma avg us per round: " + " ".join(f"{x:.1f}" for x in v)
          + f"   best {tf:.0f} TFLOP/s", flush=True)
    out[f"dbg{c[0]}_persist{c[1]}"] = {"us": [round(x, 1) for x in v], "best_tflops": round(tf, 1)}
print(json.dumps(out))

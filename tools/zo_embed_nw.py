#!/usr/bin/env python
"""zo_embed_split with 8 waves per workgroup (the product) against 4 (a second library built with -DZE_NW_DEF=4): time per
launch by map size.  usage: python tools/zo_embed_nw.py build   (here, no GPU)  /  python tools/zo_embed_nw.py   (on the GPU)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from densematcher_amd import _build

ALT = os.path.join(_build.HERE, "libdensematch_nw4.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    print(_build.build(extra_flags=("-DZE_NW_DEF=4",), lib=ALT, objdir=os.path.join(_build.BUILD, "nw4")))
    sys.exit(0)
import torch
import bench
from densematcher_amd.engine import MatchEngine

w = dict(bench.WORKLOADS["zoomout"])
host = bench.make_batch(w, 0, "f64")
B = w["B"]
K0S = [50, 66, 82, 98, 114, 130, 146, 162, 178, 192]
print("k0 (8 iterations from there):     " + " ".join(f"{k:7d}" for k in K0S))
for name, lib in (("8 waves", _build.LIB), ("4 waves", ALT)):
    eng = MatchEngine(0, lib_path=lib)
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    row = []
    for k0 in K0S:
        C0 = torch.eye(k0, dtype=torch.float64, device=eng.device).repeat(B, 1, 1)
        step = lambda: eng.zoomout(dev["Phi1"], dev["Phi2"], dev["a2"], C0, nit=8, step=1)
        step(); step(); torch.cuda.synchronize()
        eng.profile_kernel("zo_embed_split")
        for _ in range(3):
            step()
        c, ms = eng.profile_report()["zo_embed_split"]
        eng.profile_kernel("")
        row.append(1e3 * ms / c)
    print(f"zo_embed_split, {name}: us per launch " + " ".join(f"{v:7.1f}" for v in row), flush=True)

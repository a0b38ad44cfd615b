#!/usr/bin/env python
"""p2p_to_FM tile shapes (libdensematch_exp.so, DM_P2PFM_SHAPE = (R << 4 | C) << 8 forces R x C blocks per wave; bit 3 = four
slices for small maps too): time per launch at B = 32, N = 2048 for every map size class, chosen shape against forced ones.
usage: python tools/p2pfm_shapes.py   (runs itself once per setting)"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KS = [64, 80, 96, 112, 128, 144, 160, 176, 192, 200]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, REPO)
    import bench
    from densematcher_amd import _build
    from densematcher_amd.engine import MatchEngine
    eng = MatchEngine(0, lib_path=_build.LIB_EXP)
    w = dict(bench.WORKLOADS["zoomout"])
    B = int(os.environ.get("P2PFM_B", w["B"]))
    w["B"] = B
    host = bench.make_batch(w, 0, "f64")
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    N = dev["Phi1"].shape[1]
    g = torch.Generator(device="cpu").manual_seed(1)
    p21 = torch.randint(0, N, (B, N), generator=g, dtype=torch.int32).to(eng.device)
    out = []
    for k in KS:
        for _ in range(3):
            eng.p2p_to_fm(p21, dev["Phi1"], dev["Phi2"], dev["a2"], k, k)
        torch.cuda.synchronize()
        eng.profile_kernel("p2pfm_tn_f64")
        for _ in range(10):
            eng.p2p_to_fm(p21, dev["Phi1"], dev["Phi2"], dev["a2"], k, k)
        rep = eng.profile_report()
        eng.profile_kernel("")
        c, ms = rep["p2pfm_tn_f64"]
        out.append(f"{1e3 * ms / c:6.1f}")
    print(f"SHAPE {os.environ.get('DM_P2PFM_SHAPE', '0'):>6s} B {B:3d}  " + " ".join(out), flush=True)
else:
    print("k:                  " + " ".join(f"{k:6d}" for k in KS))
    sets = sys.argv[1:] or ["0", "16"]
    for sh in sets:
        v = sh
        if "x" in sh:                      # RxC[+8]
            rc, _, extra = sh.partition("+")
            r, c = rc.split("x")
            v = str(((int(r) << 4 | int(c)) << 8) + int(extra or 0))
        env = dict(os.environ, DM_P2PFM_SHAPE=v)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)

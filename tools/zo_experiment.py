#!/usr/bin/env python
"""Ablations of the fused ZoomOut kernels (libdensematch_exp.so, -DDM_EXPERIMENTS: WRONG results on purpose).
DM_ZO_DEBUG bits: zo_embed_split: 1 no split-row stores, 2 no float64 row stores, 4 one contraction stage only, 8 no epilogue;
zo_exact: 16 no work, 256 first two kept blocks of a row only, 512 every block reads the pair's first rows; p2pfm: 32 no reduction / stores, 64 three k-steps only, 128 every step re-reads the same vertices.
DM_P2PFM_SHAPE: 1 = 8 slices for every size, 2 = one row block per wave for every size.
usage: python tools/zo_experiment.py  (runs itself once per setting)"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, REPO)
    import bench
    from densematcher_amd import _build
    from densematcher_amd.engine import MatchEngine
    eng = MatchEngine(0, lib_path=_build.LIB_EXP)
    w = dict(bench.WORKLOADS["zoomout"])
    host = bench.make_batch(w, 0, "f64")
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    B = w["B"]
    C0 = torch.eye(50, dtype=torch.float64, device=eng.device).repeat(B, 1, 1)
    step = lambda: eng.zoomout(dev["Phi1"], dev["Phi2"], dev["a2"], C0, nit=150, step=1)
    step(); torch.cuda.synchronize()
    eng.profile_kernel("*")
    step()
    rep = eng.profile_report()
    eng.profile_kernel("")
    print("DBG", os.environ.get("DM_ZO_DEBUG", "0"), "SHAPE", os.environ.get("DM_P2PFM_SHAPE", "0"),
          " ".join(f"{n}={1e3 * ms / c:.1f}us" for n, (c, ms) in rep.items() if c > 10), flush=True)
else:
    # settings on the command line as dbg:shape tokens, e.g. 0:0 16:0 3:0
    sets = [tuple(a.split(":")) for a in sys.argv[1:]] or (("0", "0"), ("0", "4"), ("0", "5"))
    for dbg, shape in sets:
        env = dict(os.environ, DM_ZO_DEBUG=dbg, DM_P2PFM_SHAPE=shape)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)

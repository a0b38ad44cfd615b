#!/usr/bin/env python
"""Gram kernel of the config-2 step with 1 ... 4 stages of operand loads in flight (libdensematch_exp.so, DM_GRAM_NPRE).
usage: python tools/gram_npre.py   (runs itself once per setting)"""
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
    w = dict(bench.WORKLOADS["fmap"])
    host = bench.make_batch(w, 0, "f64")
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    for _ in range(20):
        out = eng.match(dev, k=w["k"])
    torch.cuda.synchronize()
    eng.profile_kernel("gram_nt_f64")
    for _ in range(20):
        out = eng.match(dev, k=w["k"])
    c, ms = eng.profile_report()["gram_nt_f64"]
    eng.profile_kernel("")
    print(f"NPRE {os.environ.get('DM_GRAM_NPRE', '3')}: gram_nt_f64 {1e3 * ms / c:.1f} us per launch; C checksum {float(out['C'].double().abs().sum()):.12e}", flush=True)
else:
    for n in ("1", "2", "3", "4"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, DM_GRAM_NPRE=n), check=False)

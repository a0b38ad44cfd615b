#!/usr/bin/env python
"""Per-kernel time table of one bench workload step (HIP events around every launch, dm_profile_kernel("*")).
usage: python tools/step_profile.py [fmap|stress|simnn|zoomout|icp] [--f64]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "fmap"
w = dict(bench.WORKLOADS[wl])
host = bench.make_batch(w, 0, "f32" if "--f32" in sys.argv else "f64")
eng = MatchEngine(0)
dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
k, B = w["k"], w["B"]
if wl in ("fmap", "stress"):
    step = lambda: eng.match(dev, k=k)
elif wl == "simnn":
    step = lambda: eng.simnn(dev["F2"], dev["F1"])
elif wl == "icp":
    C0 = torch.eye(k, dtype=torch.float64, device=eng.device).repeat(B, 1, 1)
    step = lambda: eng.icp(dev["Phi1"], dev["Phi2"], C0, nit=10)
else:
    C0 = torch.eye(50, dtype=torch.float64, device=eng.device).repeat(B, 1, 1)
    step = lambda: eng.zoomout(dev["Phi1"], dev["Phi2"], dev["a2"], C0, nit=150, step=1)
for _ in range(20 if wl != "zoomout" else 2):
    step()
torch.cuda.synchronize()
reps = 10 if wl != "zoomout" else 2
eng.profile_kernel("*")
for _ in range(reps):
    step()
rep = eng.profile_report()
eng.profile_kernel("")
tot = sum(ms for _, ms in rep.values())
print(f"# {wl}{' (fp32 basis)' if '--f32' in sys.argv else ' (float64 basis)'}: {tot / reps:.4f} ms of kernel time per step, {sum(n for n, _ in rep.values()) // reps} launches")
for name, (n, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:28s} {n // reps:4d} x {1e3 * ms / n:9.2f} us = {ms / reps:8.4f} ms  {100 * ms / tot:5.1f} %")

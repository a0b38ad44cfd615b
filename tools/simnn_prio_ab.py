#!/usr/bin/env python
"""A/B of dm_set_option("simnn_prio", 0 | 1): s_setprio(1) around the matrix-instruction clusters of the tile kernels
(simnn_pipe_kernel, XV bit 8192).  Per workload: the tile kernel's time and the step's kernel time with and without, and that the
results are identical.   usage: python tools/simnn_prio_ab.py [simnn fmap stress zoomout]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402

eng = MatchEngine(0)
wls = [a for a in sys.argv[1:] if not a.startswith("-")] or ["simnn", "fmap", "stress", "zoomout"]
for wl in wls:
    w = dict(bench.WORKLOADS[wl])
    if wl == "stress":
        w["B"] = 16
    host = bench.make_batch(w, 0, "f64")
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    k, B = w["k"], w["B"]
    if wl in ("fmap", "stress"):
        step = lambda: eng.match(dev, k=k)
        key = lambda r: [r[n].cpu().numpy() for n in ("knn21", "knn12", "ind21", "ind12")]
    elif wl == "simnn":
        step = lambda: eng.simnn(dev["F2"], dev["F1"])
        key = lambda r: [r.cpu().numpy()]
    else:
        C0 = torch.eye(50, dtype=torch.float64, device=eng.device).repeat(B, 1, 1)
        step = lambda: eng.zoomout(dev["Phi1"], dev["Phi2"], dev["a2"], C0, nit=40, step=1, return_p2p=True)
        key = lambda r: [r[1].cpu().numpy(), r[0].cpu().numpy()]
    res = {}
    for rnd in range(2):                       # two rounds, alternating: a drifting clock shows
        for prio in (0, 1):
            eng.set_option("simnn_prio", prio)
            for _ in range(6 if wl != "zoomout" else 1):
                out = step()
            torch.cuda.synchronize()
            reps = 8 if wl != "zoomout" else 2
            eng.profile_kernel("*")
            for _ in range(reps):
                step()
            rep = eng.profile_report()
            eng.profile_kernel("")
            tot = sum(ms for _, ms in rep.values()) / reps
            tile = {n: 1e3 * ms / cnt for n, (cnt, ms) in rep.items() if n.startswith("simnn") and n.endswith("f16_mfma")}
            res.setdefault(prio, []).append((tot, tile, key(out)))
    same = all(np.array_equal(a, b) for a, b in zip(res[0][0][2], res[1][0][2]))
    for prio in (0, 1):
        for rnd, (tot, tile, _) in enumerate(res[prio]):
            print(f"{wl:8s} prio={prio} round {rnd}: step kernels {tot:8.4f} ms; " + ", ".join(f"{n} {us:8.1f} us" for n, us in tile.items()))
    print(f"{wl:8s} identical results: {same}", flush=True)
    del dev
    torch.cuda.empty_cache()
eng.set_option("simnn_prio", 0)

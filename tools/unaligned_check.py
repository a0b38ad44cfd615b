#!/usr/bin/env python
"""The headline step on meshes whose vertex count is not a multiple of the 256-row tile (N = 2000, what DenseMatcher's meshes
look like): padded split rows + masked edge tiles (default) against the float64 G kernel (p2p_split = 0)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402

eng = MatchEngine(0)
for N in (2048, 2000, 1900):
    w = dict(bench.WORKLOADS["fmap"]); w["N"] = N
    host = bench.make_batch(w, 0, "f64")
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    ref = None
    for split in (2, 0):
        eng.set_option("p2p_split", split)
        for _ in range(5):
            out = eng.match(dev, k=w["k"])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            out = eng.match(dev, k=w["k"])
        e1.record(); torch.cuda.synchronize()
        maps = [out[k] for k in ("knn21", "knn12", "ind21", "ind12")]
        if ref is None:
            ref = [m.clone() for m in maps]
        same = all(bool((a == b).all()) for a, b in zip(ref, maps))
        print(f"N={N} p2p_split={split} (path {eng.p2p_split_active(N, N, w['k'])}): {e0.elapsed_time(e1) / 20:.3f} ms per 64 pairs, maps identical: {same}", flush=True)
    eng.reset_options()

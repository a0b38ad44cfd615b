#!/usr/bin/env python
"""Tile-order sweep of the similarity kernels: dm_set_option("simnn_band", b) for config 3 (simnn), the four-map pass of
config 2 (fmap) and of config 5 (stress); kernel time by HIP events, same process, same box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402

eng = MatchEngine(0)
for wl, kernel in (("simnn", "simnn_f16_mfma"), ("fmap", "simnn4_f16_mfma"), ("stress", "simnn4_f16_mfma")):
    w = dict(bench.WORKLOADS[wl])
    host = bench.make_batch(w, 0)
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    step = (lambda: eng.simnn(dev["F2"], dev["F1"])) if wl == "simnn" else (lambda: eng.match(dev, k=w["k"]))
    for band in (0, 2, 4, 8, 16, 0, 8):
        eng.set_option("simnn_band", band)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        eng.profile_kernel(kernel)
        for _ in range(10):
            step()
        n, ms = eng.profile_read()
        eng.profile_kernel("")
        print(f"{wl:7s} band {band:2d}: {kernel} {1e3 * ms / n:9.1f} us", flush=True)
    del dev
    torch.cuda.empty_cache()

#!/usr/bin/env python
"""Host time of one config-2 step (Python + ctypes + HIP launches and memsets, nothing waited for) against its device time:
is the step ever waiting for the host?  usage: python tools/host_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from densematcher_amd.engine import MatchEngine

eng = MatchEngine(0)
w = dict(bench.WORKLOADS["fmap"])
host = bench.make_batch(w, 0, "f64")
dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
k = w["k"]
for _ in range(20):
    eng.match(dev, k=k)
torch.cuda.synchronize()
for K in (1, 5, 20, 100):
    t0 = time.perf_counter()
    for _ in range(K):
        eng.match(dev, k=k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{K:4d} steps: enqueued in {1e3 * (t1 - t0) / K:.3f} ms per step (host), finished in {1e3 * (t2 - t0) / K:.3f} ms per step", flush=True)

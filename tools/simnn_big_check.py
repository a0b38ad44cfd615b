#!/usr/bin/env python
"""Eight waves of 128 x 64 against four waves of 128 x 128 (dm_set_option "simnn_big") on the tile kernels of configs 3, 2, 5."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402

eng = MatchEngine(0)
for wl, name in (("simnn", "simnn_f16_mfma"), ("fmap", "simnn4_f16_mfma"), ("stress", "simnn4_f16_mfma")):
    w = dict(bench.WORKLOADS[wl])
    host = bench.make_batch(w, 0, "f64")
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    step = (lambda: eng.simnn(dev["F2"], dev["F1"])) if wl == "simnn" else (lambda: eng.match(dev, k=w["k"]))
    ref = None
    for big in (0, 1, 0, 1):
        eng.set_option("simnn_big", big)
        for _ in range(4):
            out = step()
        torch.cuda.synchronize()
        eng.profile_kernel(name)
        for _ in range(8):
            out = step()
        n, ms = eng.profile_read()
        eng.profile_kernel("")
        outs = [out] if torch.is_tensor(out) else ([out[k] for k in ("knn21", "knn12", "ind21", "ind12")] if isinstance(out, dict) else list(out)[:1])
        if ref is None:
            ref = [o.clone() for o in outs]
        same = all(bool((a == b).all()) for a, b in zip(ref, outs))
        print(f"{wl:7s} simnn_big={big}: {name} {1e3 * ms / n:9.1f} us   same maps as the first run: {same}", flush=True)
    del dev
    torch.cuda.empty_cache()

#!/usr/bin/env python
"""Experiment (needs the -DDM_EXPERIMENTS build, tools only): the four-map pass (simnn_pipe_kernel<.., 3>) of config 2 and
config 5 with its row-direction / column-direction reductions removed (DM_SIMNN_DEBUG 0x1000 / 0x2000: WRONG results), to
price the main loop and the two halves of the epilogue."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd import _build  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402

eng = MatchEngine(0, lib_path=_build.LIB_EXP)
for wl in ("fmap", "stress"):
    w = dict(bench.WORKLOADS[wl])
    host = bench.make_batch(w, 0)
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    for dbg, what in ((0, "full"), (0x2000, "no column reductions"), (0x1000, "no row reductions"), (0x3000, "main loop only"), (0, "full")):
        os.environ["DM_SIMNN_DEBUG"] = str(dbg)
        for _ in range(4):
            eng.match(dev, k=w["k"])
        torch.cuda.synchronize()
        eng.profile_kernel("simnn4_f16_mfma")
        for _ in range(8):
            eng.match(dev, k=w["k"])
        n, ms = eng.profile_read()
        eng.profile_kernel("")
        print(f"{wl:7s} {what:22s}: simnn4_f16_mfma {1e3 * ms / n:9.1f} us", flush=True)
    del dev
    torch.cuda.empty_cache()

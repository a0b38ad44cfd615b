#!/usr/bin/env python
"""The five-slot-ring variant of the config-3 kernel (DM_SIMNN_DEBUG=0x20000, libdensematch_exp.so) against the product kernel:
same arg-max, kernel time.  usage: python tools/simnn_early_check.py"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import torch
    sys.path.insert(0, REPO)
    import bench
    from densematcher_amd import _build
    from densematcher_amd.engine import MatchEngine
    eng = MatchEngine(0, lib_path=_build.LIB_EXP)
    w = bench.WORKLOADS["simnn"]
    n, D, B = w["nu"] * w["nv"], w["D"], w["B"]
    feats = bench.simnn_features(B, n, D, 0)
    F1 = torch.as_tensor(feats["F1"]).to(eng.device)
    F2 = torch.as_tensor(feats["F2"]).to(eng.device)
    for _ in range(30):
        nn = eng.simnn(F2, F1)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        eng.profile_kernel("simnn_f16_mfma")
        for _ in range(10):
            eng.simnn(F2, F1)
        nl, ms = eng.profile_read()
        ts.append(1e3 * ms / nl)
    eng.profile_kernel("")
    np.save(f"/tmp/nn_{os.environ.get('DM_SIMNN_DEBUG', '0')}.npy", nn.cpu().numpy())
    print("DM_SIMNN_DEBUG", os.environ.get("DM_SIMNN_DEBUG", "0"), "kernel us:", " ".join(f"{t:.1f}" for t in ts), flush=True)
    # smaller / odd shapes through the same variant
    rng = np.random.default_rng(0)
    from oracle import dm_oracle as orc
    for (b, n2, n1, d) in [(2, 512, 768, 192), (1, 256, 256, 224), (3, 1024, 512, 416)]:
        A = rng.standard_normal((b, n2, d)).astype(np.float16); Bm = rng.standard_normal((b, n1, d)).astype(np.float16)
        got = eng.simnn(A, Bm).cpu().numpy()
        ok = all(np.array_equal(got[i], orc.simnn(A[i], Bm[i])) for i in range(b))
        print("  shape", (b, n2, n1, d), "equals oracle:", ok, flush=True)
else:
    import numpy as np
    for v in ("0", str(0x20000)):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, DM_SIMNN_DEBUG=v), check=False)
    a, b = np.load("/tmp/nn_0.npy"), np.load(f"/tmp/nn_{0x20000}.npy")
    print("same arg-max as the product variant:", np.array_equal(a, b))

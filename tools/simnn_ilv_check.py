#!/usr/bin/env python
"""The interleaved-issue variant of the tile kernels (XV bit 4096: fragment reads and LDS-DMA issued between the matrix
instructions of a k-step; DM_SIMNN_ILV=1, libdensematch_exp.so) against the product variant (reads pinned in front):
same maps, kernel time of simnn_f16_mfma (config 3), simnn4_f16_mfma (config 2) and simnn1_f16_mfma (config 4).
usage: python tools/simnn_ilv_check.py"""
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
    tag = os.environ.get("DM_SIMNN_ILV", "0")

    def kernel_us(step, name, reps=10, blocks=3, warm=5):
        for _ in range(warm):
            out = step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(blocks):
            eng.profile_kernel(name)
            for _ in range(reps):
                step()
            nl, ms = eng.profile_read()
            ts.append(1e3 * ms / max(nl, 1))
        eng.profile_kernel("")
        return out, ts

    # config 3
    w = bench.WORKLOADS["simnn"]
    n, D, B = w["nu"] * w["nv"], w["D"], w["B"]
    feats = bench.simnn_features(B, n, D, 0)
    F1 = torch.as_tensor(feats["F1"]).to(eng.device)
    F2 = torch.as_tensor(feats["F2"]).to(eng.device)
    nn, ts = kernel_us(lambda: eng.simnn(F2, F1), "simnn_f16_mfma", warm=30)
    np.save(f"/tmp/ilv_nn_{tag}.npy", nn.cpu().numpy())
    print("ILV", tag, "simnn_f16_mfma us:", " ".join(f"{t:.1f}" for t in ts), flush=True)
    del F1, F2
    # config 2
    w = dict(bench.WORKLOADS["fmap"])
    host = bench.make_batch(w, 0)
    dev = {k_: torch.as_tensor(v).to(eng.device) for k_, v in host.items()}
    out, ts = kernel_us(lambda: eng.match(dev, k=w["k"]), "simnn4_f16_mfma", reps=5)
    np.save(f"/tmp/ilv_m_{tag}.npy", np.stack([out[key].cpu().numpy() for key in ("knn21", "knn12", "ind21", "ind12")]))
    print("ILV", tag, "simnn4_f16_mfma us:", " ".join(f"{t:.1f}" for t in ts), flush=True)
    del dev
    # config 4
    w = dict(bench.WORKLOADS["zoomout"])
    host = bench.make_batch(w, 0)
    dev = {k_: torch.as_tensor(v).to(eng.device) for k_, v in host.items()}
    C0 = torch.eye(50, dtype=torch.float64, device=eng.device).repeat(w["B"], 1, 1)
    (C, p), ts = kernel_us(lambda: eng.zoomout(dev["Phi1"], dev["Phi2"], dev["a2"], C0, nit=150, step=1, return_p2p=True), "simnn1_f16_mfma",
                           reps=1, warm=1)
    np.save(f"/tmp/ilv_z_{tag}.npy", p.cpu().numpy())
    print("ILV", tag, "simnn1_f16_mfma us:", " ".join(f"{t:.1f}" for t in ts), flush=True)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.zoomout(dev["Phi1"], dev["Phi2"], dev["a2"], C0, nit=150, step=1)
    torch.cuda.synchronize()
    print("ILV", tag, "zoomout pairs/s:", round(3 * w["B"] / (time.perf_counter() - t0), 1), flush=True)
else:
    import numpy as np
    for v in ("0", "1"):     # one-key kernel: 0 = reads in front (DM_SIMNN_DEBUG=320), 1 = the product; key-set kernels: 0 = the product
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"],
                       env=dict(os.environ, DM_SIMNN_ILV=v, DM_SIMNN_DEBUG="320" if v == "0" else "0"), check=False)
    for f in ("nn", "m", "z"):
        a, b = np.load(f"/tmp/ilv_{f}_0.npy"), np.load(f"/tmp/ilv_{f}_1.npy")
        print(f"{f}: same result as the product variant:", np.array_equal(a, b), a.shape)

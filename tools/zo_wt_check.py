import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from densematcher_amd.engine import MatchEngine
eng = MatchEngine(0)
w = dict(bench.WORKLOADS["zoomout"])
host = bench.make_batch(w, 0, "f64")
dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
B = w["B"]
C0 = torch.eye(50, dtype=torch.float64, device=eng.device).repeat(B, 1, 1)
res = {}
for wt in (4, 2):
    eng.set_option("simnn1_wt", wt)
    step = lambda: eng.zoomout(dev["Phi1"], dev["Phi2"], dev["a2"], C0, nit=150, step=1, return_p2p=True)
    out = step(); torch.cuda.synchronize()
    res[wt] = out
    t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    eng.profile_kernel("*"); step(); rep = eng.profile_report(); eng.profile_kernel("")
    print(f"wt={wt}: {1e3*dt:.2f} ms = {B/dt:.1f} pairs/s;", " ".join(f"{n}={1e3 * ms / c:.1f}us" for n, (c, ms) in rep.items() if c > 10), flush=True)
print("same p:", torch.equal(res[4][1], res[2][1]), "same C:", torch.equal(res[4][0], res[2][0]))

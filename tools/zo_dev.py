#!/usr/bin/env python
"""Development check of the fused ZoomOut iteration (dm_zoomfuse.hip) and the direct p2p_to_FM kernel:
the new paths against the old ones (dm_set_option zoomout_fused / p2pfm_direct = 0) and the oracle, then timings.
usage: python tools/zo_dev.py [--time-only] [--no-time]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd import synth  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402
from oracle import dm_oracle as orc  # noqa: E402

eng = MatchEngine(0)
ok = True


def check(name, cond, msg=""):
    global ok
    print(("PASS " if cond else "FAIL ") + name + (" " + msg if msg else ""), flush=True)
    ok = ok and bool(cond)


def np_(t):
    return t.cpu().numpy()


if "--time-only" not in sys.argv:
    rng = np.random.default_rng(0)
    # ---- p2p_to_fm direct vs staged vs oracle
    for (N1, N2, k1, k2, dt) in [(300, 257, 5, 7, np.float32), (600, 900, 50, 50, np.float64), (1000, 777, 113, 96, np.float32),
                                 (2048, 2048, 200, 200, np.float64), (512, 512, 129, 17, np.float64), (640, 512, 208, 208, np.float32)]:
        B = 3
        Phi1 = (rng.standard_normal((B, N1, k1 + 3)) * 0.1).astype(dt)
        Phi2 = (rng.standard_normal((B, N2, k2 + 1)) * 0.1).astype(dt)
        a2 = rng.uniform(0.5, 1.5, (B, N2)).astype(dt)
        p = rng.integers(0, N1, (B, N2)).astype(np.int32)
        eng.set_option("p2pfm_direct", 1)
        Cd = np_(eng.p2p_to_fm(p, Phi1, Phi2, a2, k1, k2))
        eng.set_option("p2pfm_direct", 0)
        Cs = np_(eng.p2p_to_fm(p, Phi1, Phi2, a2, k1, k2))
        eng.set_option("p2pfm_direct", 1)
        Co = np.stack([orc.p2p_to_fm(p[b], Phi1[b][:, :k1].astype(np.float64), Phi2[b][:, :k2].astype(np.float64), a2[b].astype(np.float64)) for b in range(B)])
        sc = max(1.0, np.abs(Co).max())
        check(f"p2pfm direct N1={N1} N2={N2} k1={k1} k2={k2} {dt.__name__}", np.abs(Cd - Co).max() <= 1e-12 * sc and np.abs(Cd - Cs).max() <= 1e-12 * sc,
              f"vs oracle {np.abs(Cd - Co).max():.2e} vs staged {np.abs(Cd - Cs).max():.2e}")
        # batch invariance
        C1 = np_(eng.p2p_to_fm(p[1:2], Phi1[1:2], Phi2[1:2], a2[1:2], k1, k2))
        check("  batch invariant", np.array_equal(C1[0], Cd[1]))

    # ---- zoomout fused vs unfused vs oracle
    cases = [dict(nu=32, nv=16, k0=10, nit=8, step=3, dt=np.float32, B=3),
             dict(nu=32, nv=16, k0=60, nit=6, step=5, dt=np.float64, B=2),
             dict(nu=40, nv=25, k0=20, nit=12, step=4, dt=np.float64, B=2),      # N = 1000: padded tiles
             dict(nu=64, nv=32, k0=50, nit=30, step=5, dt=np.float64, B=2),
             dict(nu=64, nv=32, k0=190, nit=4, step=4, dt=np.float32, B=2)]
    for c in cases:
        kmax = c["k0"] + c["nit"] * c["step"]
        batch = synth.make_pair_batch(c["B"], c["nu"], c["nv"], 8, kmax, sigma=0.1, n_distinct_meshes=2, seed0=7, basis="random", real_dtype=c["dt"])
        C0 = np.stack([np.eye(c["k0"]) + 0.02 * np.random.default_rng(i).standard_normal((c["k0"], c["k0"])) for i in range(c["B"])])
        res = {}
        for fused in (1, 0):
            eng.set_option("zoomout_fused", fused)
            C, p = eng.zoomout(batch["Phi1"], batch["Phi2"], batch["a2"], C0, nit=c["nit"], step=c["step"], return_p2p=True)
            res[fused] = (np_(C), np_(p))
        eng.set_option("zoomout_fused", 1)
        N = c["nu"] * c["nv"]
        same_p = np.array_equal(res[1][1], res[0][1])
        dC = np.abs(res[1][0] - res[0][0]).max()
        Co, po = orc.zoomout_refine(C0[0], batch["Phi1"][0].astype(np.float64), batch["Phi2"][0].astype(np.float64), nit=c["nit"], step=c["step"],
                                    a2=batch["a2"][0].astype(np.float64), return_p2p=True)
        check(f"zoomout fused N={N} k {c['k0']}->{kmax} step {c['step']} {c['dt'].__name__}", same_p and dC <= 1e-11,
              f"p equal {same_p} ({(res[1][1] != res[0][1]).sum()} differ) dC {dC:.2e}; vs oracle p {np.array_equal(res[1][1][0], po)} C {np.abs(res[1][0][0] - Co).max():.2e}")
    # ragged sizes
    N1, N2, k0, nit, step = 520, 700, 10, 4, 3
    kmax = k0 + nit * step
    x1 = np.linspace(0, 1, N1)[:, None]; x2 = np.linspace(0, 1, N2)[:, None]
    f = np.arange(1, kmax + 1)[None, :]
    Phi1 = (np.cos(np.pi * f * x1 + rng.uniform(0, 6.28, (1, kmax))) * np.sqrt(2.0 / N1))
    Phi2 = (np.cos(np.pi * f * x2 + rng.uniform(0, 6.28, (1, kmax))) * np.sqrt(2.0 / N2))
    a2 = (rng.uniform(0.5, 1.5, N2) / N2)
    C0 = np.eye(k0) + 0.05 * rng.standard_normal((k0, k0))
    C, p = eng.zoomout(Phi1[None], Phi2[None], a2[None], C0[None], nit=nit, step=step, return_p2p=True)
    Co, po = orc.zoomout_refine(C0, Phi1, Phi2, nit=nit, step=step, a2=a2, return_p2p=True)
    check("zoomout ragged 520x700", np.array_equal(np_(p)[0], po) and np.abs(np_(C)[0] - Co).max() <= 1e-11, f"{(np_(p)[0] != po).sum()} differ, dC {np.abs(np_(C)[0] - Co).max():.2e}")
    # scale jump: C0 tiny then the map grows by orders of magnitude -> forced exact path, still right
    batch = synth.make_pair_batch(1, 32, 16, 8, 40, sigma=0.1, n_distinct_meshes=2, seed0=3, basis="random", real_dtype=np.float64)
    C0 = 1e-6 * (np.eye(20) + 0.02 * rng.standard_normal((20, 20)))
    C, p = eng.zoomout(batch["Phi1"], batch["Phi2"], batch["a2"], C0[None], nit=4, step=5, return_p2p=True)
    Co, po = orc.zoomout_refine(C0, batch["Phi1"][0], batch["Phi2"][0], nit=4, step=5, a2=batch["a2"][0], return_p2p=True)
    check("zoomout scale jump (forced exact)", np.array_equal(np_(p)[0], po), f"{(np_(p)[0] != po).sum()} differ")

if "--no-time" not in sys.argv:
    w = dict(bench.WORKLOADS["zoomout"])
    host = bench.make_batch(w, 0, "f64")
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    B = w["B"]
    C0 = torch.eye(50, dtype=torch.float64, device=eng.device).repeat(B, 1, 1)
    for fused in (1, 0):
        eng.set_option("zoomout_fused", fused)
        eng.set_option("p2pfm_direct", fused)
        step = lambda: eng.zoomout(dev["Phi1"], dev["Phi2"], dev["a2"], C0, nit=150, step=1)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"# zoomout fused={fused}: {1e3 * dt:.2f} ms per step = {B / dt:.1f} pairs/s", flush=True)
        eng.profile_kernel("*")
        step()
        rep = eng.profile_report()
        eng.profile_kernel("")
        tot = sum(ms for _, ms in rep.values())
        print(f"#   {tot:.3f} ms of kernel time, {sum(n for n, _ in rep.values())} launches")
        for name, (n, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
            print(f"    {name:28s} {n:4d} x {1e3 * ms / n:9.2f} us = {ms:8.4f} ms  {100 * ms / tot:5.1f} %")
    eng.reset_options()
print("ALL PASS" if ok else "SOME FAILED")

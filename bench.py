#!/usr/bin/env python
"""
bench.py -- mesh-pairs/s of the matching hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fmap|simnn|zoomout]

One "step" = one pass of the hot path over one batch of synthetic mesh pairs that is
already resident in HBM.  Default workload = BASELINE.json configs[1]:
    batch = 64 pairs per GPU, N = 2048 vertices (64x32 torus), D = 768 fp16 descriptors,
    k = 128 eigenfunctions:  project -> pinned column -> functional-map solve -> four vertex maps.
Pairs are independent: with N GPUs every rank processes its own 64 pairs (weak scaling, no
data-path collective); the distributed backend is used only for the timing barrier / max.

Prints ONE JSON line (rank 0) with the driver's contract keys plus `roofline` (dominant
kernel, HIP-event timed inside the timed region) and `cpu_baseline` (the NumPy oracle timed
on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from densematcher_amd import synth  # noqa: E402
from densematcher_amd.engine import MatchEngine  # noqa: E402

# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md; f64 MFMA = half the f32 MFMA rate, AMD spec 78.6 TF)
PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3, "f16": 2500.0}

WORKLOADS = {
    # name: (nu, nv, D, k, pairs per GPU)
    "fmap": dict(nu=64, nv=32, D=768, k=128, B=64, cfg="configs[1]: batch=64 pairs, N=2048, D=768, k=128 functional-map solve"),
    "simnn": dict(nu=64, nv=32, D=768, k=0, B=64, cfg="configs[2]: batch=64 pairs, N=2048, D=768 brute-force NN feature similarity + argmax"),
    "zoomout": dict(nu=64, nv=32, D=0, k=200, B=32, cfg="configs[3]: 32 pairs/GPU, N=2048, k=50->200 ZoomOut refinement"),
    "stress": dict(nu=128, nv=64, D=384, k=200, B=64, cfg="configs[4]: batch=64 pairs, N=8192, D=384, k=200 (HBM-bound stress)"),
    "icp": dict(nu=64, nv=32, D=0, k=128, B=64, cfg="SURVEY 8(f) next #1: spectral ICP, 10 iterations, 64 pairs/GPU, N=2048, k=128"),
}


def make_batch(w, rank):
    n = w["nu"] * w["nv"]
    if w["k"]:
        # N = 8192: a mass-orthonormalised seeded Gaussian stands in for the eigenbasis (throughput-only
        # runs; an ARPACK solve per mesh would dominate the set-up and the arithmetic does not depend on it)
        batch = synth.make_pair_batch(w["B"], w["nu"], w["nv"], max(w["D"], 8), w["k"], sigma=0.1, n_distinct_meshes=2,
                                      seed0=100 * rank, basis="eig" if n <= 4096 else "random")
    else:
        batch = {"F1": np.empty((w["B"], n, w["D"]), np.float16), "F2": np.empty((w["B"], n, w["D"]), np.float16)}
        for i in range(w["B"]):
            batch["F1"][i], batch["F2"][i], _ = synth.feature_pair(n, n, w["D"], 1000 + i + 1000 * rank, 2000 + i + 1000 * rank, sigma=1.0)
    return batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="fmap", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only to rehearse the N>1 path)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

    w = dict(WORKLOADS[args.workload])
    if args.batch:
        w["B"] = args.batch
    host = make_batch(w, rank)
    eng = MatchEngine(local_rank)
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in host.items()}
    N = w["nu"] * w["nv"]
    B, D, k = w["B"], w["D"], w["k"]

    if args.workload in ("fmap", "stress"):
        def step():
            return eng.match(dev, k=k)
        kernel, dtype = "gred_f64", "f64"
        flops_per_launch = 2.0 * N * N * k * B                      # G = Phi2 C Phi1^T, SURVEY 8(d): 2 N^2 k per pair
        unit_name = "mesh-pairs/s"
    elif args.workload == "simnn":
        def step():
            return eng.simnn(dev["F2"], dev["F1"])
        kernel, dtype = "simnn_f16_mfma", "f16"
        flops_per_launch = 2.0 * N * N * D * B                      # 2 N2 N1 D per pair
        unit_name = "mesh-pairs/s"
    elif args.workload == "icp":
        gen = torch.Generator(device=eng.device).manual_seed(1 + rank)
        C0 = torch.eye(k, dtype=torch.float64, device=eng.device).repeat(B, 1, 1) \
            + 0.01 * torch.randn(B, k, k, dtype=torch.float64, device=eng.device, generator=gen)

        def step():
            return eng.icp(dev["Phi1"], dev["Phi2"], C0, nit=10)
        split = os.environ.get("DM_KNN_SPLIT", "1") != "0"
        kernel, dtype = ("simnn_f16_mfma", "f16") if split else ("gred_f64", "f64")
        kd = max(96, -(-(3 * k + 8) // 32) * 32)                   # fp16 depth of the split features (dm_knnsplit.hip)
        flops_per_launch = 2.0 * N * N * (kd if split else ((k + 15) // 16) * 16) * B
        unit_name = "mesh-pairs/s"
    else:
        k0, nit = 50, 150
        C0 = torch.eye(k0, dtype=torch.float64, device=eng.device).repeat(B, 1, 1)

        def step():
            return eng.zoomout(dev["Phi1"], dev["Phi2"], dev["a2"], C0, nit=nit, step=1)
        # dominant kernel: the fp16-split first pass of the nearest-neighbour search (dm_knnsplit.hip); with
        # DM_KNN_SPLIT=0 the float64 G kernel does the same job
        split = os.environ.get("DM_KNN_SPLIT", "1") != "0"
        kernel, dtype = ("simnn_f16_mfma", "f16") if split else ("gred_f64", "f64")
        flops_per_launch = None                                     # varies with k: summed below
        unit_name = "mesh-pairs/s"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.profile_kernel(kernel)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    launches, kernel_ms = eng.profile_read()
    eng.profile_kernel("")
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=eng.device if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    pairs_total = B * world * args.steps
    value = pairs_total / elapsed
    avg_ms = kernel_ms / max(launches, 1)
    extra = {}
    if args.workload == "zoomout":
        ks = range(50, 200)
        alg = sum(2.0 * N * N * kk * B for kk in ks) / 150.0        # 2 N^2 k per pair and iteration
        if kernel == "simnn_f16_mfma":
            # what the fp16 matrix cores execute: three fp16 products per contraction index (hi*hi, hi*lo, lo*hi) plus
            # three bias entries, padded to the 32-wide stage
            flops_per_launch = sum(2.0 * N * N * max(96, -(-(3 * kk + 8) // 32) * 32) * B for kk in ks) / 150.0
            extra = {"algorithmic_f64_flops_per_launch": alg,
                     "note": "achieved/peak count the fp16 flops the split executes (3x the algorithmic 2N^2k + padding)"}
        else:
            flops_per_launch = sum(2.0 * N * N * (((kk + 15) // 16) * 16) * B for kk in ks) / 150.0
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if launches else None
    peak = PEAK_TFLOPS[dtype]
    traffic, traffic_file = (pmc_traffic_bytes(kernel, args.workload) if not args.batch else (None, None))

    out = {
        "metric": "mesh-pairs/sec at N=2048 D=768 k=128" if args.workload == "fmap" else f"mesh-pairs/sec ({args.workload})",
        "value": round(value, 2), "unit": unit_name, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "ms_per_pair": round(1e3 * elapsed / (B * args.steps), 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": w["cfg"], "pairs_per_gpu": B, "N": N, "D": D, "k": k, "parallelism": f"pairs sharded over {world} GPU(s), no collective"},
        "roofline": {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 3) if achieved else None, "peak": peak,
                     "unit": "TFLOP/s", "frac": round(achieved / peak, 4) if achieved else None, "traffic": traffic,
                     "traffic_source": f"HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/{traffic_file}: "
                                       "2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction)" if traffic else None,
                     "launches": launches, "avg_launch_ms": round(avg_ms, 4), "algorithmic_flops_per_launch": flops_per_launch,
                     **extra},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, host, k)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def pmc_traffic_bytes(kernel, workload):
    """HBM bytes per launch of the dominant kernel, measured in separate rocprofv3 --pmc passes of this same command
    (PMC collection cannot run inside the timed region); summaries committed under profiles/."""
    fname = f"r01_{workload}_hbm_traffic_pmc.csv"
    path = os.path.join(REPO, "profiles", fname)
    key = {"gred_f64": "gred_kernel", "simnn_f16_mfma": "simnn_pipe_kernel"}.get(kernel, kernel)
    try:
        import csv
        rows = [r for r in csv.reader(l for l in open(path) if not l.startswith("#"))]
        hdr = rows[0]
        for r in rows[1:]:
            if key in r[0]:
                d = dict(zip(hdr, r))
                return int((float(d["fetch_MB_corrected"]) + float(d["write_MB"])) * 1e6), fname
    except Exception:
        pass
    return None, None


def cpu_baseline(workload, host, k):
    """The NumPy float64 oracle (a port of the reference arithmetic) on the host cores, bounded sample."""
    from oracle import dm_oracle as orc
    ncores = os.cpu_count() or 1
    t0 = time.perf_counter()
    done = 0
    budget = 15.0
    if workload in ("fmap", "stress"):
        for i in range(host["F1"].shape[0]):
            orc.match_pair(host["Phi1"][i][:, :k], host["Phi2"][i][:, :k], host["lam1"][i][:k], host["lam2"][i][:k],
                           host["a1"][i], host["a2"][i], host["F1"][i], host["F2"][i])
            done += 1
            if time.perf_counter() - t0 > budget:
                break
        what = "project + closed-form solve + 4 maps (oracle.match_pair)"
    elif workload == "simnn":
        for i in range(host["F1"].shape[0]):
            orc.simnn(host["F2"][i], host["F1"][i])
            done += 1
            if time.perf_counter() - t0 > budget:
                break
        what = "float64 GEMM + argmax (oracle.simnn)"
    elif workload == "icp":
        orc.icp_refine(np.eye(k), host["Phi1"][0][:, :k], host["Phi2"][0][:, :k], nit=10)
        done = 1
        what = "spectral ICP, 10 iterations (oracle.icp_refine: brute-force NN, lstsq, SVD)"
    else:
        C0 = np.eye(50)
        orc.zoomout_refine(C0, host["Phi1"][0], host["Phi2"][0], nit=150, step=1, a2=host["a2"][0])
        done = 1
        what = "ZoomOut 50->200 step 1 (oracle.zoomout_refine)"
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 4), "unit": "mesh-pairs/s", "cores": ncores, "kind": "port",
            "sample": f"{done} pairs of the same workload, {what}, NumPy/BLAS threads on {ncores} host cores, {dt:.1f} s"}


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Per-kernel times of the eigenbasis stage of the notebook's compute_surface_map call (two meshes, one batched solve)."""
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from densematcher_amd import synth  # noqa: E402
from densematcher_amd.engine import default_engine  # noqa: E402
from densematcher_amd.pyFM.mesh import TriMesh  # noqa: E402

w = bench.WORKLOADS["surface_map"]
nu, nv, k = w["nu"], w["nv"], w["k"]
(v1, f1), (v2, f2) = synth.torus_mesh(nu, nv, perturb=0.03, seed=3), synth.torus_mesh(nu, nv, perturb=0.08, seed=1)
eng = default_engine()


def run():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        TriMesh.process_many([TriMesh(v1, f1), TriMesh(v2, f2)], [k, k], robust=True)
    torch.cuda.synchronize()


run()
run()
t0 = time.perf_counter()
run()
dt = time.perf_counter() - t0
eng.profile_kernel("*")
run()
rep = eng.profile_report()
eng.profile_kernel("")
tot = sum(ms for _, ms in rep.values())
print(f"process_many: {1e3 * dt:.1f} ms wall, {tot:.1f} ms of kernel time in {sum(n for n, _ in rep.values())} launches")
for name, (n, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:28s} {n:6d} x {1e3 * ms / n:8.2f} us = {ms:8.2f} ms")

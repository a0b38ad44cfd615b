"""
GPU (-m gpu): the N > 1 path with the real engine.  The GPU box has one device, so both ranks use cuda:0
(two processes, two HIP contexts, gloo for the gather / barrier): the partition, the per-rank engines and the gather
run exactly as on N GPUs; only the device index differs.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _batch(B):
    from densematcher_amd import synth
    return synth.make_pair_batch(B, 20, 12, 64, 32, sigma=0.3, n_distinct_meshes=2, seed0=3)


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from densematcher_amd import shard
    from densematcher_amd.engine import MatchEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = shard.match_sharded(_batch(B), lambda: MatchEngine(0), rank, world, gather=True, k=24)
    dist.barrier()
    if rank == 0:
        q.put({k: v.cpu().numpy() for k, v in res.items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 1])
def test_match_sharded_real_engine_two_ranks(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    from densematcher_amd.engine import MatchEngine
    eng = MatchEngine(0)
    ref = eng.match({k: torch.as_tensor(v).to(eng.device) for k, v in _batch(B).items()}, k=24)
    assert set(got) == {"C", "knn21", "knn12", "ind21", "ind12"}
    for name in got:
        assert np.array_equal(got[name], ref[name].cpu().numpy()), name       # a pair's result does not depend on its rank


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself and reports n_gpus = 2."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--single-device", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["pairs_per_gpu"] == 4 and out["value"] > 0


def test_bench_rccl_branch_on_one_device():
    """The `nccl` (= RCCL) branch of bench.py's timing code -- init_process_group with a device id, the barrier, the
    max-over-ranks all-reduce on a device tensor -- as far as ONE device allows: a process group of one rank (two ranks on one
    device are refused by RCCL; the 2-rank rehearsal above therefore carries its barrier over gloo)."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MASTER_PORT"] = str(_free_port())
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["process_group"] == "nccl" and out["value"] > 0


def _worker_refine(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from densematcher_amd import shard
    from densematcher_amd.engine import MatchEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = _batch(B)
    C0 = np.stack([np.eye(10)] * B)
    fac = lambda: MatchEngine(0)
    z = shard.run_sharded("zoomout", {"Phi1": b["Phi1"], "Phi2": b["Phi2"], "a2": b["a2"], "C0": C0}, fac, rank, world, nit=4, step=2)
    i = shard.run_sharded("icp", {"Phi1": b["Phi1"][:, :, :18], "Phi2": b["Phi2"][:, :, :18], "C0": np.stack([np.eye(18)] * B)}, fac, rank, world, nit=3)
    n = shard.run_sharded("simnn", {"F1": b["F1"], "F2": b["F2"]}, fac, rank, world)
    dist.barrier()
    if rank == 0:
        q.put({"zC": z["C"].cpu().numpy(), "zp": z["p2p21"].cpu().numpy(), "iC": i["C"].cpu().numpy(), "nn": n["nn21"].cpu().numpy()})
    dist.destroy_process_group()


def test_run_sharded_refinements_real_engine_two_ranks():
    """r06: ZoomOut, ICP and the feature NN through shard.run_sharded on two ranks (one device) equal the un-sharded
    engine calls bit for bit -- a pair's result does not depend on the rank that computed it"""
    B = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_refine, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    from densematcher_amd.engine import MatchEngine
    eng = MatchEngine(0)
    b = _batch(B)
    C, p21 = eng.zoomout(b["Phi1"], b["Phi2"], b["a2"], np.stack([np.eye(10)] * B), nit=4, step=2, return_p2p=True)
    assert np.array_equal(got["zC"], C.cpu().numpy()) and np.array_equal(got["zp"], p21.cpu().numpy())
    Ci = eng.icp(b["Phi1"][:, :, :18], b["Phi2"][:, :, :18], np.stack([np.eye(18)] * B), nit=3)
    assert np.array_equal(got["iC"], Ci.cpu().numpy())
    assert np.array_equal(got["nn"], eng.simnn(b["F2"], b["F1"]).cpu().numpy())

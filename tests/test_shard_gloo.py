"""CPU, world_size 2, gloo: the pairs -> ranks partition and the result gather of densematcher_amd.shard."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from densematcher_amd import shard


def test_block_range_covers_everything():
    for n in (0, 1, 7, 64, 255, 256):
        for world in (1, 2, 3, 8):
            blocks = [shard.block_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.block_range(4, 2, 2)


class _FakeEngine:
    """Stands in for MatchEngine on the CPU: a deterministic per-pair function of the inputs, so that the test
    checks the partition / ordering / gather plumbing (the kernels themselves are covered by the -m gpu tests)."""
    device = torch.device("cpu")

    def match(self, dev, k=None, **kw):
        F1, F2 = dev["F1"].float(), dev["F2"].float()
        return {"knn21": (F2 @ F1.transpose(1, 2)).argmax(dim=2).to(torch.int32),
                "C": (F1.sum(dim=(1, 2)) + 2 * F2.sum(dim=(1, 2)))[:, None, None].double().repeat(1, 2, 2),
                "ind12": None}

    def zoomout(self, Phi1, Phi2, a2, C0, nit, step=1, return_p2p=False):
        kf = C0.shape[1] + nit * step
        C = (Phi1.double().sum(dim=(1, 2)) - Phi2.double().sum(dim=(1, 2)) + a2.double().sum(dim=1))[:, None, None].repeat(1, kf, kf)
        p = (Phi2.double() @ Phi1.double().transpose(1, 2)).argmax(dim=2).to(torch.int32)
        return (C, p) if return_p2p else C

    def icp(self, Phi1, Phi2, C0, nit=10):
        return C0.double() * nit + Phi1.double().sum(dim=(1, 2))[:, None, None]

    def simnn(self, Ftgt, Fsrc):
        return (Ftgt.float() @ Fsrc.float().transpose(1, 2)).argmax(dim=2).to(torch.int32)


def _fake_surface_maps(eng, local, scale=1):
    """stands in for compute_surface_map_batch: one 14-tuple per pair, process-bound objects in slots 7-9"""
    out = []
    for m1, m2, c1, c2 in zip(local["meshes1"], local["meshes2"], local["c1s"], local["c2s"]):
        t = [np.asarray([scale * (m1 + m2) + q]) for q in range(14)]
        t[7], t[8], t[9] = object(), object(), object()
        t[2] = (np.arange(3), np.arange(3) + int(c1.sum()))
        out.append(tuple(t))
    return out


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    batch = {"F1": rng.standard_normal((B, 6, 4)).astype(np.float32), "F2": rng.standard_normal((B, 5, 4)).astype(np.float32)}
    res = shard.match_sharded(batch, _FakeEngine, rank, world, gather=True)
    dist.barrier()
    if rank == 0:
        q.put({k: v.numpy() for k, v in res.items()})
    else:
        assert res is None
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5, 1])      # 1: fewer pairs than ranks, rank 1 holds an empty block
def test_match_sharded_two_ranks(B):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    batch = {"F1": rng.standard_normal((B, 6, 4)).astype(np.float32), "F2": rng.standard_normal((B, 5, 4)).astype(np.float32)}
    ref = _FakeEngine().match({k: torch.as_tensor(v) for k, v in batch.items()})
    assert set(got) == {"knn21", "C"}
    assert np.array_equal(got["knn21"], ref["knn21"].numpy())
    assert np.array_equal(got["C"], ref["C"].numpy())


def test_match_sharded_keeps_one_engine_per_process_and_returns_local_tensors():
    """gather=False: the rank's block comes back as the engine's own tensors (no gather, no host copy); the engine is built once per
    process however many batches go through"""
    built = []
    shard.close_engines()

    class Counting(_FakeEngine):
        def __init__(self):
            built.append(1)
    rng = np.random.default_rng(1)
    for _ in range(3):
        batch = {"F1": torch.as_tensor(rng.standard_normal((5, 6, 4)).astype(np.float32)), "F2": torch.as_tensor(rng.standard_normal((5, 5, 4)).astype(np.float32))}
        res = shard.match_sharded(batch, Counting, rank=1, world=2, gather=False)
        lo, hi = shard.block_range(5, 1, 2)
        assert res["knn21"].shape[0] == hi - lo and isinstance(res["C"], torch.Tensor)
        want = _FakeEngine().match({k: v[lo:hi] for k, v in batch.items()})
        assert torch.equal(res["knn21"], want["knn21"])
    assert len(built) == 1


def _refine_batch(B):
    rng = np.random.default_rng(3)
    return {"Phi1": rng.standard_normal((B, 7, 6)), "Phi2": rng.standard_normal((B, 5, 6)), "a2": rng.random((B, 5)),
            "C0": rng.standard_normal((B, 2, 2)), "F1": rng.standard_normal((B, 7, 4)).astype(np.float32),
            "F2": rng.standard_normal((B, 5, 4)).astype(np.float32)}


def _worker_methods(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = _refine_batch(B)
    got = {}
    z = shard.run_sharded("zoomout", {k: batch[k] for k in ("Phi1", "Phi2", "a2", "C0")}, lambda: _FakeEngine(), rank, world, nit=3, step=1)
    i = shard.run_sharded("icp", {k: batch[k] for k in ("Phi1", "Phi2", "C0")}, lambda: _FakeEngine(), rank, world, nit=4)
    n = shard.run_sharded("simnn", {k: batch[k] for k in ("F1", "F2")}, lambda: _FakeEngine(), rank, world)
    # a block handed in already cut (weak scaling: no rank holds the whole batch)
    lo, hi = shard.block_range(B, rank, world)
    zb = shard.run_sharded("zoomout", {k: batch[k][lo:hi] for k in ("Phi1", "Phi2", "a2", "C0")}, lambda: _FakeEngine(), rank, world,
                           block=(lo, hi, B), nit=3, step=1)
    lists = {"meshes1": list(range(B)), "meshes2": list(range(10, 10 + B)), "c1s": [np.full(2, p) for p in range(B)], "c2s": [None] * B}
    sm = shard.run_sharded(_fake_surface_maps, lists, lambda: _FakeEngine(), rank, world, list_result=True, scale=2)
    assert len(shard._ENGINES) <= 1          # fresh lambdas every call: still one engine per process
    dist.barrier()
    if rank == 0:
        q.put({"zC": z["C"].numpy(), "zp": z["p2p21"].numpy(), "iC": i["C"].numpy(), "nn": n["nn21"].numpy(), "zbC": zb["C"].numpy(),
               "sm": [[None if x is None else (tuple(np.asarray(y) for y in x) if isinstance(x, tuple) else np.asarray(x)) for x in t] for t in sm]})
    else:
        assert z is None and i is None and n is None and sm is None and zb is None
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 1])
def test_run_sharded_every_batched_entry_point_two_ranks(B):
    """zoomout / icp / simnn / a list-valued method (compute_surface_map_batch's shape of result) through run_sharded on two gloo ranks:
    rank 0 receives every pair's results in pair order, equal to the un-sharded call"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_methods, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    batch = {k: torch.as_tensor(v) for k, v in _refine_batch(B).items()}
    e = _FakeEngine()
    C, p = e.zoomout(batch["Phi1"], batch["Phi2"], batch["a2"], batch["C0"], nit=3, step=1, return_p2p=True)
    assert np.array_equal(got["zC"], C.numpy()) and np.array_equal(got["zp"], p.numpy()) and np.array_equal(got["zbC"], C.numpy())
    assert np.array_equal(got["iC"], e.icp(batch["Phi1"], batch["Phi2"], batch["C0"], nit=4).numpy())
    assert np.array_equal(got["nn"], e.simnn(batch["F2"], batch["F1"]).numpy())
    lists = {"meshes1": list(range(B)), "meshes2": list(range(10, 10 + B)), "c1s": [np.full(2, p_) for p_ in range(B)], "c2s": [None] * B}
    want = _fake_surface_maps(None, lists, scale=2)
    assert len(got["sm"]) == B
    for t, w in zip(got["sm"], want):
        assert t[7] is None and t[8] is None and t[9] is None               # process-bound slots do not travel
        assert np.array_equal(t[0], w[0]) and np.array_equal(t[13], w[13])
        assert np.array_equal(t[2][1], w[2][1])


def test_engine_cache_is_keyed_on_the_key_not_on_the_factory_and_can_be_closed():
    shard.close_engines()
    closed = []

    class E(_FakeEngine):
        def close(self):
            closed.append(1)
    a = shard.engine_for(lambda: E())
    b = shard.engine_for(lambda: E())           # a fresh callable: the same engine
    assert a is b and len(shard._ENGINES) == 1
    c = shard.engine_for(lambda: E(), key="stream1")
    assert c is not a and len(shard._ENGINES) == 2
    shard.close_engines("stream1")
    assert len(shard._ENGINES) == 1 and len(closed) == 1
    shard.close_engines()
    assert not shard._ENGINES and len(closed) == 2

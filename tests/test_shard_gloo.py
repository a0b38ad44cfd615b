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

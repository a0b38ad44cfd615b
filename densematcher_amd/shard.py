"""
Pairs -> ranks.  The matching path shards embarrassingly: mesh pairs are independent, so a
batch of B pairs is split into contiguous blocks, one per rank (one process per GPU); every rank runs the whole
hot path on its block and owns a disjoint slice of the outputs.  There is NO data-path collective; the only
communication is an optional gather of the (small) results to rank 0 and the timing barrier of bench.py.

Works with any torch.distributed backend ("nccl" = RCCL on the GPUs, "gloo" in the CPU tests).
"""
import numpy as np
import torch


def block_range(n_items, rank, world):
    """Contiguous block [lo, hi) of rank `rank` out of `world`; the first (n_items % world) ranks get one more."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(batch, rank, world):
    """Slice every (B, ...) array / tensor of `batch` to this rank's block of pairs."""
    B = next(iter(batch.values())).shape[0]
    lo, hi = block_range(B, rank, world)
    return {k: v[lo:hi] for k, v in batch.items()}, (lo, hi)


def gather_results(local, n_items, rank=None, world=None, dst=0, group=None):
    """Gather per-rank result dicts (each value (b_local, ...)) to `dst` in pair order.  Returns the full dict on
    `dst`, None elsewhere.  Blocks may have different sizes, so the gather is padded to the largest block."""
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank(group)
    if world is None:
        world = dist.get_world_size(group)
    sizes = [block_range(n_items, r, world) for r in range(world)]
    maxb = max(hi - lo for lo, hi in sizes)
    out = {} if rank == dst else None
    for key in sorted(local):
        t = local[key]
        t = t if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t))
        pad = torch.zeros((maxb,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, bufs, dst=dst, group=group)
        if rank == dst:
            out[key] = torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
    return out


def match_sharded(batch_host, engine_factory, rank, world, gather=True, **match_kwargs):
    """Run the hot path on this rank's block of `batch_host` (dict of host arrays with leading pair axis).
    `engine_factory()` returns a MatchEngine for this rank's GPU.  With gather=True rank 0 receives every map."""
    local, (lo, hi) = shard_batch(batch_host, rank, world)
    eng = engine_factory()
    dev = {k: torch.as_tensor(v).to(eng.device) for k, v in local.items()}
    res = eng.match(dev, **match_kwargs) if hi > lo else {}
    if not gather or world == 1:
        return res
    B = next(iter(batch_host.values())).shape[0]
    return gather_results({k: v for k, v in res.items() if v is not None}, B, rank, world)

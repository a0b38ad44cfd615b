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
    `dst`, None elsewhere.  Blocks may have different sizes (or be empty: fewer pairs than ranks), so every rank first
    learns the key / dtype / trailing-shape list from the lowest rank that has results, then takes part in one padded
    gather per key -- the same collectives on every rank, whatever its block holds."""
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank(group)
    if world is None:
        world = dist.get_world_size(group)
    sizes = [block_range(n_items, r, world) for r in range(world)]
    maxb = max(hi - lo for lo, hi in sizes)
    local = {k: (v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))) for k, v in (local or {}).items()
             if v is not None}
    # schema = [(key, dtype name, trailing shape)], published by the first rank with a non-empty block
    owner = next((r for r, (lo, hi) in enumerate(sizes) if hi > lo), 0)
    schema = [[(k, str(local[k].dtype).replace("torch.", ""), tuple(local[k].shape[1:])) for k in sorted(local)]] \
        if rank == owner else [None]
    dist.broadcast_object_list(schema, src=owner, group=group)
    dev = next(iter(local.values())).device if local else torch.device("cpu")
    backend = dist.get_backend(group)
    if backend == "nccl" and dev.type != "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    out = {} if rank == dst else None
    for key, dtname, trail in schema[0]:
        dtype = getattr(torch, dtname)
        t = local.get(key)
        pad = torch.zeros((maxb,) + tuple(trail), dtype=dtype, device=dev)
        if t is not None and t.shape[0]:
            pad[: t.shape[0]] = t.to(dev)
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, bufs, dst=dst, group=group)
        if rank == dst:
            out[key] = torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
    return out


_ENGINES = {}


def engine_for(engine_factory):
    """the engine of this process for `engine_factory`, created on first use and kept: a rank calls match_sharded once per batch, and
    a MatchEngine owns a HIP context, a workspace arena (grown to the largest call so far) and the kernels' LDS grants -- building one
    per call threw all of that away"""
    if engine_factory not in _ENGINES:
        _ENGINES[engine_factory] = engine_factory()
    return _ENGINES[engine_factory]


def match_sharded(batch, engine_factory, rank, world, gather=True, **match_kwargs):
    """Run the hot path on this rank's block of `batch` (dict of arrays or tensors with leading pair axis; tensors already on the
    rank's GPU are used where they are).  `engine_factory()` returns a MatchEngine for this rank's GPU; it is called once per
    process (engine_for).  gather=False: this rank's results as DEVICE tensors (no copy, no collective); gather=True: rank 0 receives
    every map (one padded gather per output)."""
    local, (lo, hi) = shard_batch(batch, rank, world)
    res = {}
    if hi > lo:
        eng = engine_for(engine_factory)
        dev = {k: (v if isinstance(v, torch.Tensor) and v.device == eng.device else torch.as_tensor(v).to(eng.device)) for k, v in local.items()}
        res = eng.match(dev, **match_kwargs)
    if not gather or world == 1:
        return res
    B = next(iter(batch.values())).shape[0]
    return gather_results({k: v for k, v in res.items() if v is not None}, B, rank, world)

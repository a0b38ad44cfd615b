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


def engine_for(engine_factory, key=None):
    """The engine of this process under `key` (default: ONE engine per process = per rank), created by `engine_factory()` on first use
    and kept: a rank calls run_sharded once per batch, and a MatchEngine owns a HIP context, a workspace arena (grown to the largest
    call so far) and the kernels' LDS grants -- building one per call threw all of that away.  The cache is keyed on
    `key`, NOT on the factory object: a caller that passes a fresh lambda per call still gets the same engine (keyed on
    the callable, every call of such a caller added an engine that was never released).  Use distinct keys only for engines that
    must coexist (e.g. one per stream); close_engines() releases them."""
    if key not in _ENGINES:
        _ENGINES[key] = engine_factory()
    return _ENGINES[key]


def close_engines(key=None):
    """close and forget the cached engine under `key`, or all of them (key=None)"""
    keys = list(_ENGINES) if key is None else ([key] if key in _ENGINES else [])
    for k in keys:
        eng = _ENGINES.pop(k)
        close = getattr(eng, "close", None)
        if close is not None:
            close()


def _to_engine(eng, local):
    return {k: (v if isinstance(v, torch.Tensor) and v.device == eng.device else torch.as_tensor(v).to(eng.device)) for k, v in local.items()}


def _run_match(eng, local, **kw):
    return eng.match(_to_engine(eng, local), **kw)


def _run_zoomout(eng, local, nit, step=1, **kw):
    """local: Phi1, Phi2, a2, C0 -> {"C": refined maps, "p2p21": the last iteration's vertex maps} (pyFM/refine/zoomout.py:7-115)"""
    d = _to_engine(eng, local)
    C, p21 = eng.zoomout(d["Phi1"], d["Phi2"], d["a2"], d["C0"], nit=nit, step=step, return_p2p=True, **kw)
    return {"C": C, "p2p21": p21}


def _run_icp(eng, local, nit=10, **kw):
    """local: Phi1, Phi2, C0 -> {"C"} (pyFM/refine/icp.py:10-107)"""
    d = _to_engine(eng, local)
    return {"C": eng.icp(d["Phi1"], d["Phi2"], d["C0"], nit=nit, **kw)}


def _run_simnn(eng, local, **kw):
    """local: F2 (targets), F1 (sources) -> {"nn21"}"""
    d = _to_engine(eng, local)
    return {"nn21": eng.simnn(d["F2"], d["F1"], **kw)}


def _run_surface_map_batch(eng, local, **kw):
    """local: lists meshes1, meshes2, c1s, c2s (one entry per pair) -> the list of compute_surface_map's 14-tuples
    (functional_map.py:9-81).  `eng` only fixes the device: the call runs on default_engine()'s streams of that device."""
    from .functional_map import compute_surface_map_batch
    dev = getattr(eng, "device", None)
    if dev is not None and dev.type == "cuda":
        torch.cuda.set_device(dev)
    return compute_surface_map_batch(local["meshes1"], local["meshes2"], local["c1s"], local["c2s"], **kw)


METHODS = {"match": _run_match, "zoomout": _run_zoomout, "icp": _run_icp, "simnn": _run_simnn,
           "compute_surface_map_batch": _run_surface_map_batch}

# slots of compute_surface_map's 14-tuple that hold host objects tied to this process (model, mesh1, mesh2: they reference the rank's
# engine and device tensors); a gathered tuple carries None there
_TUPLE_LOCAL_SLOTS = (7, 8, 9)


def _n_items(batch):
    v = next(iter(batch.values()))
    return len(v) if isinstance(v, (list, tuple)) else v.shape[0]


def run_sharded(method, batch, engine_factory, rank, world, gather=True, block=None, engine_key=None, group=None, list_result=None, **kwargs):
    """Run one batched entry point of the hot path on this rank's block of pairs.

    method   "match" | "zoomout" | "icp" | "simnn" | "compute_surface_map_batch", or a callable (engine, local_batch, **kwargs) ->
             dict of (b_local, ...) tensors, or -- list_result=True -- a list with one entry per pair (every rank must know which, also
             a rank whose block is empty: list_result defaults to True for "compute_surface_map_batch" only)
    batch    dict of arrays / tensors / lists with a leading pair axis: the WHOLE batch (every rank passes the same one and takes its
             contiguous block, block_range), or -- block=(lo, hi, n_items) -- already this rank's block lo:hi of an n_items batch
             that no rank holds whole (weak scaling: bench.py builds 64 pairs per rank)
    gather   False: this rank's results where they are (device tensors / the local list), no collective;  True: rank 0 receives
             every pair's results in pair order (tensors: one padded gather per key; lists: gather_object, with the process-bound
             slots of a 14-tuple set to None), the other ranks None
    There is no data-path collective: pairs are independent."""
    fn = METHODS[method] if isinstance(method, str) else method
    if block is None:
        B = _n_items(batch)
        lo, hi = block_range(B, rank, world)
        local = {k: v[lo:hi] for k, v in batch.items()}
    else:
        lo, hi, B = block
        local = batch
        if _n_items(local) != hi - lo:
            raise ValueError("block=(lo, hi, n_items) does not match the batch handed in")
        if (lo, hi) != block_range(B, rank, world):
            raise ValueError("block is not this rank's block_range")
    if list_result is None:
        list_result = method == "compute_surface_map_batch"
    res = [] if list_result else {}
    if hi > lo:
        res = fn(engine_for(engine_factory, engine_key), local, **kwargs)
        if isinstance(res, list) != bool(list_result):
            raise TypeError("run_sharded: the method's result type does not match list_result")
    if not gather or world == 1:
        return res
    if list_result:
        import torch.distributed as dist
        mine = [tuple(None if (isinstance(t, tuple) and len(t) == 14 and q in _TUPLE_LOCAL_SLOTS) else x for q, x in enumerate(t))
                if isinstance(t, tuple) else t for t in res]
        bufs = [None] * world if rank == 0 else None
        dist.gather_object(mine, bufs, dst=0, group=group)
        return [t for part in bufs for t in part] if rank == 0 else None
    return gather_results({k: v for k, v in res.items() if v is not None}, B, rank, world, group=group)


def match_sharded(batch, engine_factory, rank, world, gather=True, **match_kwargs):
    """run_sharded("match", ...): Run the hot path on this rank's block of `batch` (dict of arrays or tensors with leading pair axis;
    tensors already on the rank's GPU are used where they are).  `engine_factory()` returns a MatchEngine for this rank's GPU; it is
    called once per process (engine_for).  gather=False: this rank's results as DEVICE tensors (no copy, no collective);
    gather=True: rank 0 receives every map (one padded gather per output)."""
    return run_sharded("match", batch, engine_factory, rank, world, gather=gather, **match_kwargs)

"""
GPU (-m gpu): the fused ZoomOut iteration (csrc/dm_zoomfuse.hip: embedding + split rows, biased-key search, merge, exact) and the
direct p2p_to_FM kernel (csrc/dm_zoomout.hip: p2pfm_direct_kernel) against the six-launch loop they replace (dm_set_option
"zoomout_fused" / "p2pfm_direct" = 0) and against the oracle (oracle/dm_oracle.py: zoomout_refine, p2p_to_fm; reference
pyFM/refine/zoomout.py:7-44, pyFM/spectral/convert.py:14-51).  Vertex maps bit-exact, C within 1e-11.
"""
import numpy as np
import pytest
import torch

from densematcher_amd import synth
from oracle import dm_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def _engine():
    from densematcher_amd.engine import MatchEngine
    return MatchEngine()


@pytest.fixture
def eng(_engine):
    yield _engine
    _engine.reset_options()


def _np(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("N1,N2,k1,k2,dt", [(300, 257, 5, 7, np.float32), (600, 900, 50, 50, np.float64), (1000, 777, 113, 96, np.float32),
                                            (2048, 2048, 200, 200, np.float64), (512, 512, 129, 17, np.float64), (640, 512, 208, 208, np.float32),
                                            (700, 640, 17, 130, np.float64)])
def test_p2p_to_fm_direct_kernel(eng, N1, N2, k1, k2, dt):
    """every tile shape of the direct kernel (1 x c and 2 x c blocks per wave, overlapping last groups, rectangular maps, row
    strides that are not multiples of 16) against the staged kernel and the oracle; a pair's result does not depend on its batch"""
    rng = np.random.default_rng(N1 + k1)
    B = 3
    Phi1 = (rng.standard_normal((B, N1, k1 + 3)) * 0.1).astype(dt)
    Phi2 = (rng.standard_normal((B, N2, k2 + 1)) * 0.1).astype(dt)
    a2 = rng.uniform(0.5, 1.5, (B, N2)).astype(dt)
    p = rng.integers(0, N1, (B, N2)).astype(np.int32)
    Cd = _np(eng.p2p_to_fm(p, Phi1, Phi2, a2, k1, k2))
    eng.set_option("p2pfm_direct", 0)
    Cs = _np(eng.p2p_to_fm(p, Phi1, Phi2, a2, k1, k2))
    eng.set_option("p2pfm_direct", 1)
    Co = np.stack([orc.p2p_to_fm(p[b], Phi1[b][:, :k1].astype(np.float64), Phi2[b][:, :k2].astype(np.float64), a2[b].astype(np.float64))
                   for b in range(B)])
    sc = max(1.0, np.abs(Co).max())
    assert Cd.shape == (B, k2, k1)
    assert np.abs(Cd - Co).max() <= 1e-12 * sc
    assert np.abs(Cd - Cs).max() <= 1e-12 * sc
    C1 = _np(eng.p2p_to_fm(p[1:2], Phi1[1:2], Phi2[1:2], a2[1:2], k1, k2))
    assert np.array_equal(C1[0], Cd[1])
    # out-of-range map entries are clamped like the staged kernel clamps them
    pb = p.copy()
    pb[0, :5] = [-3, N1, N1 + 7, -1, 2 ** 30]
    Cc = _np(eng.p2p_to_fm(pb, Phi1, Phi2, a2, k1, k2))
    Cr = _np(eng.p2p_to_fm(np.clip(pb, 0, N1 - 1), Phi1, Phi2, a2, k1, k2))
    assert np.array_equal(Cc, Cr)


@pytest.mark.parametrize("k", [64, 100, 144, 160, 176, 192, 200])
def test_p2p_to_fm_tile_shape_follows_the_batch_result_does_not(eng, k):
    """the direct kernel picks its tile (R x C blocks per wave: 1 x c, 2 x c, 3 x 5, 5 x 3, 6 x 3 ...) from a cost model of the batch:
    the same pair inside batches of 1, 2, 9, 16, 33 and 64 pairs (different shapes, different numbers of workgroups per CU) gives the
    same bits, equal to the oracle's to rounding"""
    rng = np.random.default_rng(k)
    N1, N2, Bmax = 384, 512, 64
    Phi1 = (rng.standard_normal((Bmax, N1, k)) * 0.1)
    Phi2 = (rng.standard_normal((Bmax, N2, k + 2)) * 0.1)
    a2 = rng.uniform(0.5, 1.5, (Bmax, N2))
    p = rng.integers(0, N1, (Bmax, N2)).astype(np.int32)
    Co = orc.p2p_to_fm(p[0], Phi1[0][:, :k], Phi2[0][:, :k], a2[0])
    ref = None
    for B in (1, 2, 9, 16, 33, 64):
        C = _np(eng.p2p_to_fm(p[:B], Phi1[:B], Phi2[:B], a2[:B], k, k))
        assert np.abs(C[0] - Co).max() <= 1e-12 * max(1.0, np.abs(Co).max())
        if ref is None:
            ref = C[0]
        assert np.array_equal(C[0], ref), B
        if B > 1:
            assert np.abs(C[B - 1] - orc.p2p_to_fm(p[B - 1], Phi1[B - 1][:, :k], Phi2[B - 1][:, :k], a2[B - 1])).max() <= 1e-12


@pytest.mark.parametrize("nu,nv,k0,nit,step,dt,B", [(32, 16, 10, 8, 3, np.float32, 3), (32, 16, 60, 6, 5, np.float64, 2),
                                                   (40, 25, 20, 12, 4, np.float64, 2), (64, 32, 50, 30, 5, np.float64, 2),
                                                   (64, 32, 190, 4, 4, np.float32, 2), (24, 11, 3, 5, 1, np.float64, 2)])
def test_zoomout_fused_equals_unfused_and_oracle(eng, nu, nv, k0, nit, step, dt, B):
    """the fused (five-launch) iteration against the six-launch one (same vertex maps; C to rounding: the embedding and p2p_to_FM sum in
    other orders) and the oracle, on aligned (N = 512, 2048) and padded (N = 1000, 264) sizes, depths below the tile kernel's
    minimum (k < 65: zero-padded rows) and up to 206, fp32 and float64 bases"""
    kmax = k0 + nit * step
    batch = synth.make_pair_batch(B, nu, nv, 8, kmax, sigma=0.1, n_distinct_meshes=2, seed0=7, basis="random", real_dtype=dt)
    C0 = np.stack([np.eye(k0) + 0.02 * np.random.default_rng(i).standard_normal((k0, k0)) for i in range(B)])
    res = {}
    for fused in (1, 0):
        eng.set_option("zoomout_fused", fused)
        C, p = eng.zoomout(batch["Phi1"], batch["Phi2"], batch["a2"], C0, nit=nit, step=step, return_p2p=True)
        res[fused] = (_np(C), _np(p))
    assert np.array_equal(res[1][1], res[0][1])
    assert np.abs(res[1][0] - res[0][0]).max() <= 1e-11
    for b in range(B):
        Co, po = orc.zoomout_refine(C0[b], batch["Phi1"][b].astype(np.float64), batch["Phi2"][b].astype(np.float64), nit=nit, step=step,
                                    a2=batch["a2"][b].astype(np.float64), return_p2p=True)
        assert np.array_equal(res[1][1][b], po), b
        assert np.abs(res[1][0][b] - Co).max() <= 1e-11, b
    # a pair's trajectory does not depend on the batch it is in
    C1, p1 = eng.zoomout(batch["Phi1"][1:2], batch["Phi2"][1:2], batch["a2"][1:2], C0[1:2], nit=nit, step=step, return_p2p=True)
    assert np.array_equal(_np(C1)[0], res[1][0][1]) and np.array_equal(_np(p1)[0], res[1][1][1])
    # nit = 0 returns the input map; without return_p2p the last search is skipped
    Cz = eng.zoomout(batch["Phi1"], batch["Phi2"], batch["a2"], C0, nit=0)
    assert np.array_equal(_np(Cz), C0)
    Cn = eng.zoomout(batch["Phi1"], batch["Phi2"], batch["a2"], C0, nit=nit, step=step)
    assert np.array_equal(_np(Cn), res[1][0])


def test_zoomout_fused_ragged_sizes(eng):
    """N1 != N2, neither a multiple of a tile (edge tiles mask the padding)"""
    rng = np.random.default_rng(3)
    N1, N2, k0, nit, step = 520, 700, 10, 4, 3
    kmax = k0 + nit * step
    x1, x2 = np.linspace(0, 1, N1)[:, None], np.linspace(0, 1, N2)[:, None]
    f = np.arange(1, kmax + 1)[None, :]
    Phi1 = np.cos(np.pi * f * x1 + rng.uniform(0, 6.28, (1, kmax))) * np.sqrt(2.0 / N1)
    Phi2 = np.cos(np.pi * f * x2 + rng.uniform(0, 6.28, (1, kmax))) * np.sqrt(2.0 / N2)
    a2 = rng.uniform(0.5, 1.5, N2) / N2
    C0 = np.eye(k0) + 0.05 * rng.standard_normal((k0, k0))
    C, p = eng.zoomout(Phi1[None], Phi2[None], a2[None], C0[None], nit=nit, step=step, return_p2p=True)
    Co, po = orc.zoomout_refine(C0, Phi1, Phi2, nit=nit, step=step, a2=a2, return_p2p=True)
    assert np.array_equal(_np(p)[0], po)
    assert np.abs(_np(C)[0] - Co).max() <= 1e-11


def test_zoomout_fused_scale_jump_takes_the_exact_path(eng):
    """The source rows of an iteration are scaled with the previous iteration's maximum.  A start map of size 1e-6 makes the
    embedding grow by six orders of magnitude between the first two iterations: the merge kernel must notice (ratio outside
    [1/4, 8)) and send every row of that pair through the exact float64 evaluation -- the maps stay those of the oracle."""
    rng = np.random.default_rng(5)
    batch = synth.make_pair_batch(2, 32, 16, 8, 40, sigma=0.1, n_distinct_meshes=2, seed0=3, basis="random", real_dtype=np.float64)
    C0 = np.stack([1e-6 * (np.eye(20) + 0.02 * rng.standard_normal((20, 20))), np.eye(20) + 0.02 * rng.standard_normal((20, 20))])
    C, p = eng.zoomout(batch["Phi1"], batch["Phi2"], batch["a2"], C0, nit=4, step=5, return_p2p=True)
    for b in range(2):
        Co, po = orc.zoomout_refine(C0[b], batch["Phi1"][b], batch["Phi2"][b], nit=4, step=5, a2=batch["a2"][b], return_p2p=True)
        assert np.array_equal(_np(p)[b], po), b
        assert np.abs(_np(C)[b] - Co).max() <= 1e-11, b
    # an all-zero start map: every score ties, the lowest index wins everywhere, like np.argmin
    Cz, pz = eng.zoomout(batch["Phi1"][:1], batch["Phi2"][:1], batch["a2"][:1], np.zeros((1, 20, 20)), nit=1, step=5, return_p2p=True)
    Co, po = orc.zoomout_refine(np.zeros((20, 20)), batch["Phi1"][0], batch["Phi2"][0], nit=1, step=5, a2=batch["a2"][0], return_p2p=True)
    assert np.array_equal(_np(pz)[0], po)


def test_requeued_rows_diagnostic(eng, fx_cfg2):
    """dm_last_requeued_rows: between 0 and N rows per map after a split-path fm_to_p2p, -1 after the float64 kernel"""
    fx = fx_cfg2
    b = lambda x: np.ascontiguousarray(x)[None]
    eng.fm_to_p2p(b(fx["Phi1"]), b(fx["Phi2"]), b(fx["a1"]), b(fx["C_f64"]))
    rows = eng.last_requeued_rows()
    n = fx["Phi1"].shape[0]
    assert all(0 <= r <= n for r in rows), rows
    eng.set_option("p2p_split", 0)
    eng.fm_to_p2p(b(fx["Phi1"]), b(fx["Phi2"]), b(fx["a1"]), b(fx["C_f64"]))
    assert eng.last_requeued_rows() == [-1, -1, -1, -1]


@pytest.mark.parametrize("N2,direct", [(768, 0), (15360, 1)])
def test_zoomout_fused_with_the_staged_p2p_to_fm_on_a_fresh_context(N2, direct):
    """the fused loop reserved the prescaled basis but not the split-K partials of the staged p2p_to_FM, which
    dm_launch_p2p_to_fm falls back to with p2pfm_direct = 0 and beyond 15000 target vertices -- DM_ENOMEM on a context whose arena had
    not been grown by an earlier call.  A fresh engine per case; result against the six-launch loop."""
    from densematcher_amd.engine import MatchEngine
    rng = np.random.default_rng(N2)
    N1, k0, nit, step = 512, 12, 3, 4
    kmax = k0 + nit * step
    Phi1 = (rng.standard_normal((1, N1, kmax)) / np.sqrt(N1)).astype(np.float64)
    Phi2 = (rng.standard_normal((1, N2, kmax)) / np.sqrt(N2)).astype(np.float64)
    a2 = (rng.uniform(0.5, 1.5, (1, N2)) / N2)
    C0 = (np.eye(k0) + 0.05 * rng.standard_normal((k0, k0)))[None]
    eng = MatchEngine()
    try:
        eng.set_option("p2pfm_direct", direct)
        C, p = eng.zoomout(Phi1, Phi2, a2, C0, nit=nit, step=step, return_p2p=True)          # (raised MemoryError before the fix)
        eng.set_option("zoomout_fused", 0)
        Cu, pu = eng.zoomout(Phi1, Phi2, a2, C0, nit=nit, step=step, return_p2p=True)
        assert np.array_equal(_np(p), _np(pu))
        assert np.abs(_np(C) - _np(Cu)).max() <= 1e-11
    finally:
        eng.close()

"""
GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the
reference-generated golden fixtures.  Integer maps must be bit-exact; C within 1e-4.
"""
import numpy as np
import pytest
import torch

from oracle import dm_oracle as orc

pytestmark = pytest.mark.gpu

C_TOL = 1e-4          # BASELINE.json north_star: float C matrix within 1e-4


@pytest.fixture(scope="module")
def _engine():
    from densematcher_amd.engine import MatchEngine
    return MatchEngine()


@pytest.fixture
def eng(_engine):
    """the module's engine; code-path options (dm_set_option) are back at their defaults after every test"""
    yield _engine
    _engine.reset_options()


def _np(t):
    return t.cpu().numpy()


def _b(x):
    """add the batch axis"""
    return np.ascontiguousarray(x)[None]


# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("fxname", ["fx_cfg1", "fx_cfg2", "fx_cfg5"])
def test_project_and_solve(eng, fxname, request):
    fx = request.getfixturevalue(fxname)
    k = int(fx["k"])
    wd, wl = float(fx["w_descr"]), float(fx["w_lap"])
    A64 = orc.project(fx["Phi1"][:, :k], fx["a1"], fx["F1"])
    B64 = orc.project(fx["Phi2"][:, :k], fx["a2"], fx["F2"])
    # float64 matrix-core path: fp32 output rounding only
    Ax = _np(eng.project(_b(fx["Phi1"]), _b(fx["a1"]), _b(fx["F1"]), k, exact=True))[0]
    assert np.abs(Ax - A64).max() <= 2e-7 * np.abs(A64).max() + 1e-12
    A32 = _np(eng.project(_b(fx["Phi1"]), _b(fx["a1"]), _b(fx["F1"].astype(np.float32)), k))[0]   # fp32 descriptors
    assert np.array_equal(A32, Ax)
    # default path for fp16 descriptors: fp16 matrix cores, basis split in two fp16 pieces
    A = _np(eng.project(_b(fx["Phi1"]), _b(fx["a1"]), _b(fx["F1"]), k))[0]
    Bm = _np(eng.project(_b(fx["Phi2"]), _b(fx["a2"]), _b(fx["F2"]), k))[0]
    ea, eb = np.abs(A - A64).max() / np.abs(A64).max(), np.abs(Bm - B64).max() / np.abs(B64).max()
    print(f"{fxname}: split-fp16 projection relative error {ea:.2e} {eb:.2e}")
    assert ea <= 3e-6 and eb <= 3e-6

    c00 = _np(eng.c00(_b(fx["Phi1"]), _b(fx["Phi2"]), _b(fx["a1"]), _b(fx["a2"])))[0]
    x0 = orc.get_x0(k, k, float(fx["Phi1"][0, 0]), float(fx["Phi2"][0, 0]),
                    float(fx["a1"].astype(np.float64).sum()), float(fx["a2"].astype(np.float64).sum()))
    assert abs(c00 - x0[0, 0]) <= 1e-14 * abs(x0[0, 0])

    C = _np(eng.fmap_solve(_b(A), _b(Bm), _b(fx["lam1"][:k]), _b(fx["lam2"][:k]), np.array([c00]), wd, wl))[0]
    err = np.abs(C - fx["C_f64"]).max()
    print(f"{fxname}: |C_gpu - C_f64| = {err:.3e}   |C_gpu - C_fit(reference fp32 L-BFGS)| = {np.abs(C - fx['C_fit']).max():.3e}")
    assert err <= C_TOL
    assert err <= 1e-5                     # what this design actually delivers on these fixtures
    assert np.array_equal(C[1:, 0], np.zeros(k - 1)) and C[0, 0] == c00
    # solve fed with the oracle's float64 projections (rounded to f32): isolates the solver
    C2 = _np(eng.fmap_solve(_b(A64.astype(np.float32)), _b(B64.astype(np.float32)), _b(fx["lam1"][:k]),
                            _b(fx["lam2"][:k]), np.array([x0[0, 0]]), wd, wl))[0]
    C2o = orc.fmap_solve(A64.astype(np.float32), B64.astype(np.float32), fx["lam1"][:k], fx["lam2"][:k], x0, wd, wl)
    assert np.abs(C2 - C2o).max() <= 1e-9


@pytest.mark.parametrize("fxname,pre,cname", [("fx_cfg1", "", "C_fit"), ("fx_cfg2", "", "C_fit"),
                                              ("fx_cfg2", "f64_", "C_f64"), ("fx_cfg1", "icp_", "C_icp"),
                                              ("fx_cfg5", "", "C_fit"), ("fx_cfg5", "f64_", "C_f64")])
def test_fm_to_p2p_bit_exact(eng, fxname, pre, cname, request):
    fx = request.getfixturevalue(fxname)
    k = int(fx["k"])
    out = eng.fm_to_p2p(_b(fx["Phi1"]), _b(fx["Phi2"]), _b(fx["a1"]), _b(fx[cname]))
    for name in ["knn21", "knn12", "ind21", "ind12"]:
        got = _np(out[name])[0].astype(np.int64)
        assert np.array_equal(got, fx[pre + name]), f"{fxname}/{name}: {(got != fx[pre + name]).sum()} mismatches"
    # sliced eigenvectors (ld == k) give the same answer as ld > k
    out2 = eng.fm_to_p2p(_b(fx["Phi1"][:, :k]), _b(fx["Phi2"][:, :k]), _b(fx["a1"]), _b(fx[cname]), knn=True, ind=False)
    assert out2["ind21"] is None
    assert np.array_equal(_np(out2["knn21"]), _np(out["knn21"]))


@pytest.mark.parametrize("B,N1,N2,k1,k2,ld1,ld2", [(2, 300, 517, 20, 33, 24, 33), (1, 129, 128, 17, 16, 17, 20),
                                                  (3, 1000, 777, 64, 50, 64, 64), (1, 257, 1030, 5, 9, 8, 12)])
def test_fm_to_p2p_ragged_rectangular(eng, B, N1, N2, k1, k2, ld1, ld2):
    """N1 != N2, k1 != k2, leading dimensions > k, sizes that are not multiples of any tile: all four maps bit-exact"""
    rng = np.random.default_rng(N1 * 31 + N2)
    Phi1 = (rng.standard_normal((B, N1, ld1)) * 0.05).astype(np.float32)
    Phi2 = (rng.standard_normal((B, N2, ld2)) * 0.05).astype(np.float32)
    a1 = (rng.uniform(0.5, 1.5, (B, N1)) / N1).astype(np.float32)
    C = rng.standard_normal((B, k2, k1))
    out = eng.fm_to_p2p(Phi1, Phi2, a1, C)
    for b in range(B):
        ref = orc.fm_to_p2p_all(C[b], Phi1[b], Phi2[b], a1[b])
        for name, r in zip(["knn21", "knn12", "ind21", "ind12"], ref):
            got = _np(out[name])[b].astype(np.int64)
            assert np.array_equal(got, r), f"pair {b} {name}: {(got != r).sum()} mismatches"


def test_ties_lowest_index(eng, fx_ties):
    fx = fx_ties
    out = eng.fm_to_p2p(_b(fx["Phi1"]), _b(fx["Phi2"]), _b(fx["a1"]), _b(fx["C"]))
    # indicator maps: np.argmax first-index rule, pinned by the reference itself
    assert np.array_equal(_np(out["ind21"])[0], fx["ind21"])
    assert np.array_equal(_np(out["ind12"])[0], fx["ind12"])
    # nearest-neighbour maps: equal to the lowest-index oracle (the kd-tree's own tie order is traversal dependent)
    p21, p12, _ = orc.fm_to_p2p(fx["C"], fx["Phi1"].astype(np.float64), fx["Phi2"].astype(np.float64), fx["a1"],
                                with_indicator=False)
    assert np.array_equal(_np(out["knn21"])[0], p21)
    assert np.array_equal(_np(out["knn12"])[0], p12)


def test_p2p_to_fm_config5_size_against_reference(eng, fx_cfg5):
    """N = 8192, k = 200: p2p_to_FM of the reference's own knn21 (convert.py:39-51) against the reference's result"""
    fx, k = fx_cfg5, int(fx_cfg5["k"])
    C = _np(eng.p2p_to_fm(_b(fx["knn21"].astype(np.int32)), _b(fx["Phi1"]), _b(fx["Phi2"]), _b(fx["a2"]), k, k))[0]
    assert np.abs(C - fx["C_from_p2p"]).max() <= 1e-12 * max(1.0, np.abs(fx["C_from_p2p"]).max())


def test_p2p_to_fm(eng, fx_cfg1, fx_cfg2):
    for fx, k in [(fx_cfg1, 30), (fx_cfg1, 48), (fx_cfg2, 128)]:
        C = _np(eng.p2p_to_fm(_b(fx["knn21"].astype(np.int32)), _b(fx["Phi1"]), _b(fx["Phi2"]), _b(fx["a2"]), k, k))[0]
        Co = orc.p2p_to_fm(fx["knn21"], fx["Phi1"][:, :k], fx["Phi2"][:, :k], fx["a2"])
        assert np.abs(C - Co).max() <= 1e-13 * max(1.0, np.abs(Co).max())
    assert np.abs(C[:30, :30] - 0).max() > 0
    C30 = _np(eng.p2p_to_fm(_b(fx_cfg1["knn21"].astype(np.int32)), _b(fx_cfg1["Phi1"]), _b(fx_cfg1["Phi2"]),
                            _b(fx_cfg1["a2"]), 30, 30))[0]
    assert np.abs(C30 - fx_cfg1["C_from_p2p"]).max() <= 1e-13
    # rectangular map
    C = _np(eng.p2p_to_fm(_b(fx_cfg1["knn21"].astype(np.int32)), _b(fx_cfg1["Phi1"]), _b(fx_cfg1["Phi2"]),
                          _b(fx_cfg1["a2"]), 20, 40))[0]
    Co = orc.p2p_to_fm(fx_cfg1["knn21"], fx_cfg1["Phi1"][:, :20], fx_cfg1["Phi2"][:, :40], fx_cfg1["a2"])
    assert C.shape == (40, 20) and np.abs(C - Co).max() <= 1e-13


def test_zoomout(eng, fx_cfg1, fx_cfg2):
    fx = fx_cfg1
    C, p = eng.zoomout(_b(fx["Phi1"]), _b(fx["Phi2"]), _b(fx["a2"]), _b(fx["C20"]), nit=20, step=1, return_p2p=True)
    assert np.array_equal(_np(p)[0], fx["p21_zo"])
    assert np.abs(_np(C)[0] - fx["C_zo"]).max() <= 1e-11
    C, p = eng.zoomout(_b(fx["Phi1"]), _b(fx["Phi2"]), _b(fx["a2"]), _b(fx["C20"]), nit=6, step=4, return_p2p=True)
    assert np.array_equal(_np(p)[0], fx["p21_zo4"])
    assert np.abs(_np(C)[0] - fx["C_zo4"]).max() <= 1e-11
    fx = fx_cfg2
    C, p = eng.zoomout(_b(fx["Phi1"]), _b(fx["Phi2"]), _b(fx["a2"]), _b(fx["C_fit"]), nit=3, step=4, return_p2p=True)
    assert np.array_equal(_np(p)[0], fx["p21_zo"])
    assert np.abs(_np(C)[0] - fx["C_zo"]).max() <= 1e-11
    C0 = eng.zoomout(_b(fx["Phi1"]), _b(fx["Phi2"]), _b(fx["a2"]), _b(fx["C_fit"]), nit=0)
    assert np.array_equal(_np(C0)[0], fx["C_fit"])
    with pytest.raises(AssertionError):
        eng.zoomout(_b(fx["Phi1"]), _b(fx["Phi2"]), _b(fx["a2"]), _b(fx["C_fit"]), nit=4, step=4)


@pytest.mark.parametrize("N1,N2,k1,k2,D", [(1, 1, 1, 1, 8), (2, 3, 1, 2, 8), (3, 2, 2, 2, 16), (5, 7, 3, 4, 8), (17, 16, 2, 2, 24)])
def test_tiny_shapes(eng, N1, N2, k1, k2, D):
    """degenerate sizes (a single vertex, a single eigenfunction): every entry point still equals the oracle"""
    rng = np.random.default_rng(N1 * 100 + N2)
    Phi1 = (rng.standard_normal((1, N1, k1)) * 0.3).astype(np.float32)
    Phi2 = (rng.standard_normal((1, N2, k2)) * 0.3).astype(np.float32)
    a1 = rng.uniform(0.5, 1.5, (1, N1)).astype(np.float32)
    a2 = rng.uniform(0.5, 1.5, (1, N2)).astype(np.float32)
    C = rng.standard_normal((1, k2, k1))
    out = eng.fm_to_p2p(Phi1, Phi2, a1, C)
    ref = orc.fm_to_p2p_all(C[0], Phi1[0], Phi2[0], a1[0])
    for name, r in zip(["knn21", "knn12", "ind21", "ind12"], ref):
        assert np.array_equal(_np(out[name])[0], r), name
    F1 = rng.standard_normal((1, N1, D)).astype(np.float16)
    F2 = rng.standard_normal((1, N2, D)).astype(np.float16)
    assert np.array_equal(_np(eng.simnn(F2, F1))[0], orc.simnn(F2[0], F1[0]))
    Ao = orc.project(Phi1[0], a1[0], F1[0])
    assert np.abs(_np(eng.project(Phi1, a1, F1, k1))[0] - Ao).max() <= 1e-5 * np.abs(Ao).max()
    p = rng.integers(0, N1, (1, N2)).astype(np.int32)
    Cpo = orc.p2p_to_fm(p[0], Phi1[0], Phi2[0], a2[0])
    assert np.abs(_np(eng.p2p_to_fm(p, Phi1, Phi2, a2, k1, k2))[0] - Cpo).max() <= 1e-12 * max(1.0, np.abs(Cpo).max())


def _smooth_basis(rng, N, k):
    """random low-frequency-looking columns (mass-orthonormal is not needed by the arithmetic under test)"""
    x = np.linspace(0.0, 1.0, N)[:, None]
    f = np.arange(1, k + 1)[None, :]
    return (np.cos(np.pi * f * x + rng.uniform(0, 6.28, (1, k))) * np.sqrt(2.0 / N)).astype(np.float32)


@pytest.mark.parametrize("N1,N2,k0,nit,step", [(400, 333, 6, 5, 2), (520, 700, 10, 4, 3)])
def test_refine_ragged(eng, N1, N2, k0, nit, step):
    """ZoomOut, p2p_to_FM and ICP on meshes of different sizes (N1 != N2, nothing a multiple of a tile)"""
    rng = np.random.default_rng(N1 + N2)
    kmax = k0 + nit * step
    Phi1, Phi2 = _smooth_basis(rng, N1, kmax), _smooth_basis(rng, N2, kmax)
    a2 = (rng.uniform(0.5, 1.5, N2) / N2).astype(np.float32)
    C0 = np.eye(k0) + 0.05 * rng.standard_normal((k0, k0))
    C, p = eng.zoomout(_b(Phi1), _b(Phi2), _b(a2), _b(C0), nit=nit, step=step, return_p2p=True)
    Co, po = orc.zoomout_refine(C0, Phi1, Phi2, nit=nit, step=step, a2=a2, return_p2p=True)
    assert np.array_equal(_np(p)[0].astype(np.int64), po)
    assert np.abs(_np(C)[0] - Co).max() <= 1e-11
    # rectangular p2p -> FM
    k1, k2 = k0 + 3, k0 + 1
    Cr = _np(eng.p2p_to_fm(_b(po.astype(np.int32)), _b(Phi1), _b(Phi2), _b(a2), k1, k2))[0]
    Cro = orc.p2p_to_fm(po, Phi1[:, :k1], Phi2[:, :k2], a2)
    assert Cr.shape == (k2, k1) and np.abs(Cr - Cro).max() <= 1e-13 * max(1.0, np.abs(Cro).max())
    # ICP from the same start: orthonormal columns, same map as the oracle
    Ci = _np(eng.icp(_b(Phi1[:, :k0]), _b(Phi2[:, :k0]), _b(C0), nit=3))[0]
    Cio = orc.icp_refine(C0, Phi1[:, :k0], Phi2[:, :k0], nit=3)
    assert np.abs(Ci - Cio).max() <= 1e-9


# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("B,N2,N1,D", [(1, 128, 128, 64), (2, 300, 517, 96), (3, 1000, 777, 384), (1, 2048, 2048, 768),
                                       (2, 256, 512, 96), (2, 512, 256, 64), (1, 768, 512, 160), (1, 512, 512, 136), (3, 512, 768, 128), (5, 256, 256, 96)])
@pytest.mark.parametrize("pipe", ["persist", "pertile", "edge", "big"])
def test_simnn_random(eng, B, N2, N1, D, pipe):
    # interior shapes (N % 256 == 0) take the ring-buffered LDS-DMA kernel when D % 32 == 0 and D >= 96 (one persistent
    # workgroup per CU when there are more tiles than CUs, else -- or with simnn_persist = 0 -- one workgroup per tile);
    # everything else, and everything with simnn_pipe = 0, the bounds-checked register-staged kernel
    eng.set_option("simnn_pipe", 0 if pipe == "edge" else 1)
    eng.set_option("simnn_persist", 8 if pipe == "persist" else 0)      # 8 workgroups walk all the tiles
    eng.set_option("simnn_big", 1 if pipe == "big" else 0)              # four waves of 128 x 128 instead of eight of 128 x 64
    rng = np.random.default_rng(B * 1000 + N2)
    S = rng.standard_normal((B, N1, D)).astype(np.float16)
    T = rng.standard_normal((B, N2, D)).astype(np.float16)
    nn, best, margin = eng.simnn(T, S, return_scores=True)
    nn = _np(nn)
    for b in range(B):
        ref = orc.simnn(T[b], S[b])
        assert np.array_equal(nn[b], ref), f"{(nn[b] != ref).sum()} mismatches"
        sc = (T[b].astype(np.float64) @ S[b].astype(np.float64).T)
        srt = np.sort(sc, axis=1)
        assert np.abs(_np(best)[b] - srt[:, -1]).max() <= 2e-4 * np.abs(srt).max()
        assert np.abs(_np(margin)[b] - (srt[:, -1] - srt[:, -2])).max() <= 4e-4 * np.abs(srt).max()


def test_simnn_ties_and_near_ties(eng):
    rng = np.random.default_rng(5)
    D, N1, N2 = 256, 640, 384
    S = rng.standard_normal((N1, D)).astype(np.float16)
    S[400:420] = S[10:30]                     # exact duplicates: lowest index must win
    T = S[rng.integers(0, N1, size=N2)].copy()
    # near ties: rows whose two best candidates differ in the last fp16 bit of one coordinate
    S[500] = S[100]
    S[500, 7] = np.nextafter(S[500, 7], np.float16(np.inf))
    T[0] = S[100]
    nn = _np(eng.simnn(T[None], S[None]))[0]
    assert np.array_equal(nn, orc.simnn(T, S))
    assert (nn[np.isin(nn, np.arange(10, 30))].size > 0) and not np.isin(nn, np.arange(400, 420)).any()


def test_simnn_unit_norm_hard(eng):
    """BASELINE config-3 recipe at reduced B: unit-norm rows, F2 = F1[perm] + sigma noise."""
    from densematcher_amd import synth
    for sigma in (0.1, 1.0):
        F1, F2, perm = synth.feature_pair(2048, 2048, 768, 1000, 2000, sigma=sigma)
        nn = _np(eng.simnn(F2[None], F1[None]))[0]
        assert np.array_equal(nn, orc.simnn(F2, F1))
        if sigma == 0.1:
            assert np.array_equal(nn, perm)


# --------------------------------------------------------------------------- #
def test_match_batch_end_to_end(eng):
    """project -> solve -> maps for a batch with distinct pairs; C within 1e-4 of the oracle,
    maps bit-exact when the oracle consumes the same C, end-to-end agreement reported."""
    from densematcher_amd import synth
    B, k = 3, 40
    batch = synth.make_pair_batch(B, 32, 16, 96, 56, sigma=0.5, n_distinct_meshes=2)
    dev = {n: torch.as_tensor(v).to(eng.device) for n, v in batch.items()}
    out = eng.match(dev, k=k, check=True)
    agree = []
    for b in range(B):
        Co, k21, k12, i21, i12 = orc.match_pair(batch["Phi1"][b][:, :k], batch["Phi2"][b][:, :k], batch["lam1"][b][:k],
                                                batch["lam2"][b][:k], batch["a1"][b], batch["a2"][b], batch["F1"][b],
                                                batch["F2"][b])
        Cg = _np(out["C"])[b]
        assert np.abs(Cg - Co).max() <= C_TOL
        same = orc.fm_to_p2p_all(Cg, batch["Phi1"][b][:, :k].astype(np.float64), batch["Phi2"][b][:, :k].astype(np.float64),
                                 batch["a1"][b])
        for got, name in zip(same, ["knn21", "knn12", "ind21", "ind12"]):
            assert np.array_equal(_np(out[name])[b], got), name
        agree.append(np.mean([(_np(out["knn21"])[b] == k21).mean(), (_np(out["ind21"])[b] == i21).mean()]))
    print("end-to-end map agreement with the float64 oracle:", agree)
    assert min(agree) >= 0.995


def test_errors(eng):
    z = np.zeros((1, 64, 8), np.float32)
    with pytest.raises(ValueError):
        eng.simnn(np.zeros((1, 8, 12), np.float16), np.zeros((1, 8, 12), np.float16))      # D % 8
    with pytest.raises(ValueError):
        eng.project(z, np.ones((1, 63), np.float32), np.zeros((1, 64, 4), np.float16), 8)
    with pytest.raises(AssertionError):
        eng.fm_to_p2p(z, z, np.ones((1, 64), np.float32), np.zeros((1, 9, 9)))
    # rank-deficient descriptors and w_lap = 0: the system is singular and must be reported
    from densematcher_amd._lib import DenseMatchError
    A = np.zeros((1, 6, 16), np.float32)
    with pytest.raises(DenseMatchError):
        eng.fmap_solve(A, A, np.arange(6.0)[None], np.arange(6.0)[None], np.ones(1), 1.0, 0.0)


@pytest.mark.parametrize("k1,k2,D", [(2, 3, 16), (17, 17, 40), (50, 40, 96), (128, 128, 256), (177, 60, 200), (178, 20, 256), (193, 9, 256), (194, 300, 208),
                                     (200, 24, 256)])
@pytest.mark.parametrize("packed", ["0", "1"])
def test_solver_shapes(eng, k1, k2, D, packed):
    """blocked-MFMA Cholesky (n <= 176), its two-phase form (177 <= n <= 199: 12 or 13 block rows, more systems than
    workgroups at k2 = 300) and the packed-storage rank-4 solver (forced for every shape by dm_set_option
    "solve_packed"), square and rectangular maps"""
    eng.set_option("solve_packed", int(packed))
    rng = np.random.default_rng(k1 * 7 + k2)
    Bn = 2
    A = rng.standard_normal((Bn, k1, D)).astype(np.float32) * 0.1
    Bm = rng.standard_normal((Bn, k2, D)).astype(np.float32) * 0.1
    lam1 = np.sort(rng.uniform(0, 50, (Bn, k1)), axis=1); lam1[:, 0] = 0
    lam2 = np.sort(rng.uniform(0, 60, (Bn, k2)), axis=1); lam2[:, 0] = 0
    c00 = np.array([1.0, -0.9])
    C = _np(eng.fmap_solve(A, Bm, lam1, lam2, c00, 1e4, 1e3))
    for b in range(Bn):
        x0 = np.zeros((k2, k1)); x0[0, 0] = c00[b]
        Co = orc.fmap_solve(A[b], Bm[b], lam1[b], lam2[b], x0, 1e4, 1e3)
        assert np.abs(C[b] - Co).max() <= 1e-9 * max(1.0, np.abs(Co).max()), np.abs(C[b] - Co).max()


@pytest.mark.parametrize("B,N,D,k,ld", [(1, 77, 24, 5, 7), (2, 300, 200, 33, 40), (1, 1000, 136, 130, 130)])
def test_project_ragged(eng, B, N, D, k, ld):
    """shapes that are not multiples of any tile, both projection paths"""
    rng = np.random.default_rng(N)
    Phi = rng.standard_normal((B, N, ld)).astype(np.float32) * 0.03
    a = rng.uniform(0.5, 1.5, (B, N)).astype(np.float32) / N
    F = rng.standard_normal((B, N, D)).astype(np.float16)
    fast = _np(eng.project(Phi, a, F, k))
    exact = _np(eng.project(Phi, a, F, k, exact=True))
    for b in range(B):
        ref = orc.project(Phi[b][:, :k], a[b], F[b])
        assert np.abs(exact[b] - ref).max() <= 2e-7 * np.abs(ref).max()
        assert np.abs(fast[b] - ref).max() <= 3e-6 * np.abs(ref).max()


# --------------------------------------------------------------------------- #
# randomised shapes (hypothesis): whatever the sizes, the integer outputs equal the oracle's
def _fuzz_settings():
    from hypothesis import HealthCheck, settings
    return settings(max_examples=25, deadline=None, derandomize=True,
                    suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


def test_fuzz_fm_to_p2p(eng):
    from hypothesis import given, strategies as st

    @_fuzz_settings()
    @given(st.integers(1, 3), st.integers(1, 400), st.integers(1, 400), st.integers(1, 40), st.integers(1, 40), st.integers(0, 9),
           st.integers(0, 2 ** 31 - 1))
    def run(B, N1, N2, k1, k2, extra_ld, seed):
        rng = np.random.default_rng(seed)
        Phi1 = (rng.standard_normal((B, N1, k1 + extra_ld)) * 0.1).astype(np.float32)
        Phi2 = (rng.standard_normal((B, N2, k2 + (extra_ld // 2))) * 0.1).astype(np.float32)
        a1 = rng.uniform(0.1, 2.0, (B, N1)).astype(np.float32)
        C = rng.standard_normal((B, k2, k1))
        out = eng.fm_to_p2p(Phi1, Phi2, a1, C)
        for b in range(B):
            ref = orc.fm_to_p2p_all(C[b], Phi1[b], Phi2[b], a1[b])
            for name, r in zip(["knn21", "knn12", "ind21", "ind12"], ref):
                assert np.array_equal(_np(out[name])[b], r), (name, B, N1, N2, k1, k2)
    run()


@pytest.mark.parametrize("B,N1,N2,D,k1,k2,dt", [(2, 2048, 2048, 256, 128, 128, np.float64), (3, 1500, 900, 128, 40, 33, np.float32),
                                                (1, 700, 2500, 64, 17, 50, np.float64), (2, 4100, 800, 96, 30, 30, np.float32)])
def test_fmap_fit_equals_the_three_calls(eng, B, N1, N2, D, k1, k2, dt):
    """dm_fmap_fit (one call; the Gram kernel adds up the projections' split-K chunks as it reads them) returns the same C,
    bit for bit, as dm_project x 2 + dm_fmap_c00 + dm_fmap_solve; one, two and more chunks per operand"""
    rng = np.random.default_rng(N1 + N2)
    Phi1 = (rng.standard_normal((B, N1, k1)) * 0.05).astype(dt)
    Phi2 = (rng.standard_normal((B, N2, k2)) * 0.05).astype(dt)
    a1 = (rng.uniform(0.5, 1.5, (B, N1)) / N1).astype(dt)
    a2 = (rng.uniform(0.5, 1.5, (B, N2)) / N2).astype(dt)
    F1 = rng.standard_normal((B, N1, D)).astype(np.float16)
    F2 = rng.standard_normal((B, N2, D)).astype(np.float16)
    lam1 = np.sort(rng.uniform(0, 50, (B, k1)), axis=1); lam1[:, 0] = 0
    lam2 = np.sort(rng.uniform(0, 60, (B, k2)), axis=1); lam2[:, 0] = 0
    A = eng.project(Phi1, a1, F1, k1)
    Bm = eng.project(Phi2, a2, F2, k2)
    c00 = eng.c00(Phi1, Phi2, a1, a2)
    want = _np(eng.fmap_solve(A, Bm, lam1, lam2, c00, 1e4, 1e3))
    got = _np(eng.fmap_fit(Phi1, Phi2, a1, a2, F1, F2, lam1, lam2, 1e4, 1e3))
    assert np.array_equal(got, want), float(np.abs(got - want).max())


def test_fuzz_fm_to_p2p_split_sizes(eng):
    """the fp16 split path on arbitrary mesh sizes (padded operands, masked edge strips, both tile shapes) against the float64 G
    kernel: identical maps"""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=int(__import__("os").environ.get("DM_FUZZ_EXAMPLES", "12")), deadline=None, derandomize=True)
    @given(st.integers(1, 2), st.integers(256, 1400), st.integers(256, 1400), st.integers(65, 140), st.integers(20, 140),
           st.sampled_from([2, 3, 4, 1]), st.integers(0, 2 ** 31 - 1))
    def run(B, N1, N2, k2, k1, split, seed):
        rng = np.random.default_rng(seed)
        Phi1 = (rng.standard_normal((B, N1, k1)) * 0.1).astype(np.float64)
        Phi2 = (rng.standard_normal((B, N2, k2)) * 0.1).astype(np.float64)
        a1 = rng.uniform(0.1, 2.0, (B, N1))
        a1[:, rng.integers(0, N1, 3)] = 0.0                  # a few zero masses: ind12 = 0 there
        C = rng.standard_normal((B, k2, k1)) / np.sqrt(k1)
        assert eng.p2p_split_active(N2, N1, k2)
        eng.set_option("p2p_split", split)
        got = {k: _np(v) for k, v in eng.fm_to_p2p(Phi1, Phi2, a1, C).items()}
        eng.set_option("p2p_split", 0)
        want = {k: _np(v) for k, v in eng.fm_to_p2p(Phi1, Phi2, a1, C).items()}
        eng.reset_options()
        for name in ("knn21", "knn12", "ind21", "ind12"):
            assert np.array_equal(got[name], want[name]), (name, B, N1, N2, k1, k2, split, int((got[name] != want[name]).sum()))
    run()


def test_fuzz_simnn(eng):
    from hypothesis import given, strategies as st

    @_fuzz_settings()
    @given(st.integers(1, 3), st.integers(1, 600), st.integers(1, 600), st.integers(1, 40), st.booleans(), st.integers(0, 2 ** 31 - 1))
    def run(B, N2, N1, d8, dup, seed):
        rng = np.random.default_rng(seed)
        D = 8 * d8
        S = rng.standard_normal((B, N1, D)).astype(np.float16)
        T = rng.standard_normal((B, N2, D)).astype(np.float16)
        if dup and N1 > 4:                                   # duplicated source rows: the lowest index has to win
            S[:, N1 // 2] = S[:, 1]
            T[:, 0] = S[:, 1]
        nn = _np(eng.simnn(T, S))
        for b in range(B):
            assert np.array_equal(nn[b], orc.simnn(T[b], S[b])), (B, N2, N1, D)
    run()


def test_fuzz_project_and_solve(eng):
    from hypothesis import given, strategies as st

    @_fuzz_settings()
    @given(st.integers(1, 2), st.integers(2, 300), st.integers(1, 48), st.integers(1, 48), st.integers(1, 12), st.integers(0, 2 ** 31 - 1))
    def run(B, N, k1, k2, d8, seed):
        rng = np.random.default_rng(seed)
        D = 8 * d8
        Phi1 = (rng.standard_normal((B, N, k1)) * 0.1).astype(np.float32)
        Phi2 = (rng.standard_normal((B, N, k2)) * 0.1).astype(np.float32)
        a = (rng.uniform(0.5, 1.5, (B, N)) / N).astype(np.float32)
        F1 = rng.standard_normal((B, N, D)).astype(np.float16)
        F2 = rng.standard_normal((B, N, D)).astype(np.float16)
        A = _np(eng.project(Phi1, a, F1, k1, exact=True))
        Bm = _np(eng.project(Phi2, a, F2, k2, exact=True))
        lam1 = np.sort(rng.uniform(0, 50, (B, k1)), axis=1); lam1[:, 0] = 0
        lam2 = np.sort(rng.uniform(0, 60, (B, k2)), axis=1); lam2[:, 0] = 0
        c00 = rng.choice([-1.0, 1.0], B) * rng.uniform(0.5, 1.5, B)
        C = _np(eng.fmap_solve(A, Bm, lam1, lam2, c00, 1e4, 1e3))
        for b in range(B):
            Ao = orc.project(Phi1[b], a[b], F1[b])
            assert np.abs(A[b] - Ao).max() <= 2e-7 * max(np.abs(Ao).max(), 1e-30)
            x0 = np.zeros((k2, k1)); x0[0, 0] = c00[b]
            Co = orc.fmap_solve(A[b], Bm[b], lam1[b], lam2[b], x0, 1e4, 1e3)
            assert np.abs(C[b] - Co).max() <= 1e-8 * max(1.0, np.abs(Co).max()), (k1, k2, D, np.abs(C[b] - Co).max())
    run()


# --------------------------------------------------------------------------- #
# nearest-neighbour search on the fp16-split first pass (dm_knnsplit.hip) -- the path of ZoomOut, ICP and knn_query
@pytest.mark.parametrize("split", ["1", "0"])
def test_knn_query_adversarial(eng, split):
    """exact duplicates (lowest index wins), last-bit near ties, wildly different operand scales (bias overflow ->
    every row takes the exact path), an all-zero operand, an offset cloud (every margin inside the bound)"""
    eng.set_option("knn_split", int(split))
    rng = np.random.default_rng(11)
    nx, ny, p = 700, 300, 24
    X = rng.standard_normal((nx, p))
    X[400:420] = X[10:30]                                       # duplicated tree points
    X[500] = X[100]; X[500, 3] = np.nextafter(X[500, 3], np.inf)
    Y = X[rng.integers(0, nx, ny)] + 1e-3 * rng.standard_normal((ny, p))
    Y[0] = X[100]; Y[1] = X[500]; Y[2] = X[15]
    cases = {"base": (X, Y), "scaled tree": (X * 3e7, Y * 3e7), "tiny": (X * 1e-12, Y * 1e-12),
             "mixed scale": (X * 1e6, Y * 1e-3), "offset": (X + 1000.0, Y + 1000.0), "zero query": (X, np.zeros_like(Y)),
             "zero tree": (np.zeros_like(X), Y)}
    for name, (Xc, Yc) in cases.items():
        got = _np(eng.knn_query(Xc[None], Yc[None]))[0]
        ref = orc.knn_query(Xc, Yc)
        assert np.array_equal(got, ref), f"{name}: {(got != ref).sum()} mismatches"
    assert not np.isin(_np(eng.knn_query(X[None], Y[None]))[0], np.arange(400, 420)).any()


def test_zoomout_split_equals_f64_kernel(eng):
    """the two nearest-neighbour implementations give the same ZoomOut trajectory bit for bit (config-4 shape, reduced)"""
    from densematcher_amd import synth
    batch = synth.make_pair_batch(2, 64, 32, 8, 80, sigma=0.1, n_distinct_meshes=2, seed0=5, basis="random")
    C0 = np.eye(40)[None].repeat(2, axis=0)
    res = {}
    for split in ("1", "0"):
        eng.set_option("knn_split", int(split))
        C, p = eng.zoomout(batch["Phi1"], batch["Phi2"], batch["a2"], C0, nit=10, step=4, return_p2p=True)
        res[split] = (_np(C), _np(p))
    assert np.array_equal(res["1"][1], res["0"][1])
    assert np.array_equal(res["1"][0], res["0"][0])


def _ulp_tie(C, P1, P2, a1, name, idx, got, want):
    """both candidates of a mismatching entry score within a few float64 ulps of each other (extended precision)"""
    ld = np.longdouble
    k2, k1 = C.shape
    e1, e2, Cl = P1[:, :k1].astype(ld), P2[:, :k2].astype(ld), C.astype(ld)

    def score(i, j):                                   # the value the reference compares for target i / candidate j
        g = e2[i] @ Cl @ e1[j]
        if name == "knn21":
            y = Cl @ e1[j]
            return float(y @ y - 2 * g), float(abs(y @ y) + 2 * abs(g))
        if name == "knn12":
            x = e2[i] @ Cl
            return float(x @ x - 2 * g), float(abs(x @ x) + 2 * abs(g))
        return float(g * ld(a1[j])), float(abs(g) * float(a1[j]))
    if name in ("knn21", "ind21"):
        (sg, sc), (sw, _) = score(idx, got), score(idx, want)
    else:
        (sg, sc), (sw, _) = score(got, idx), score(want, idx)
    return abs(sg - sw) <= 64 * np.finfo(np.float64).eps * max(sc, 1e-300)


def _split_case(rng, B, N1, N2, k1, k2, kind):
    """operands for dm_fm_to_p2p that stress the fp16-split path: near-delta maps, exact duplicates, zero / spread masses"""
    Phi1 = rng.standard_normal((B, N1, k1)).astype(np.float32)
    Phi2 = rng.standard_normal((B, N2, k2)).astype(np.float32)
    C = rng.standard_normal((B, k2, k1)) / np.sqrt(k1)
    a1 = rng.uniform(0.5, 1.5, (B, N1)).astype(np.float32)
    if kind == "permuted":                                  # a true correspondence: Phi2 = Phi1[perm] + noise, C ~ identity
        for b in range(B):
            perm = rng.permutation(N1)[:N2] if N2 <= N1 else rng.integers(0, N1, N2)
            km = min(k1, k2)
            Phi2[b][:, :km] = Phi1[b][perm][:, :km] + 1e-3 * rng.standard_normal((N2, km)).astype(np.float32)
        C = np.eye(k2, k1)[None].repeat(B, axis=0) + 1e-3 * rng.standard_normal((B, k2, k1))
    if kind == "duplicates":                                # exact ties in all four reductions
        Phi1[:, N1 // 2:N1 // 2 + 40] = Phi1[:, :40]
        Phi2[:, 100:130] = Phi2[:, 300:330]
        a1[:, N1 // 2:N1 // 2 + 40] = a1[:, :40]
    if kind == "masses":                                    # zero masses, five decades of spread
        a1 = (10.0 ** rng.uniform(-5, 0, (B, N1))).astype(np.float32)
        a1[:, ::97] = 0.0
    if kind == "scales":                                    # operand scales far apart, smooth decay over the spectrum
        Phi1 *= (1e-3 * 0.97 ** np.arange(k1)).astype(np.float32)
        Phi2 *= (2e2 * 0.95 ** np.arange(k2)).astype(np.float32)
    return Phi1, Phi2, a1, C


@pytest.mark.parametrize("kind", ["random", "permuted", "duplicates", "masses", "scales"])
def test_fm_to_p2p_split_equals_f64_kernel(eng, kind):
    """the four maps from the two-pass fp16 tile kernel + exact re-evaluation equal those of the float64 G kernel"""
    rng = np.random.default_rng({"random": 1, "permuted": 2, "duplicates": 3, "masses": 4, "scales": 5}[kind])
    # (sizes that are not whole 256-tiles run on padded operands: the tiles that reach into the padding mask it)
    for (B, N1, N2, k1, k2) in ((3, 512, 768, 64, 80), (2, 1024, 512, 72, 100), (2, 600, 900, 64, 80), (1, 1000, 400, 72, 100)):
        Phi1, Phi2, a1, C = _split_case(rng, B, N1, N2, k1, k2, kind)
        assert eng.p2p_split_active(N2, N1, k2)
        res = {}
        for split in (4, 3, 2, 1, 0):   # 2: one pass in both directions (shape by size), 3 / 4: its 4-wave / 8-wave shape; 1: two passes; 0: float64 kernel
            eng.set_option("p2p_split", split)
            res[split] = {k: _np(v) for k, v in eng.fm_to_p2p(Phi1, Phi2, a1, C).items()}
        # the one-pass kernel launched one workgroup per tile / with an odd number of persistent workgroups (tile walks of
        # different lengths, bias slots of both parities)
        for tag, persist in (("pertile", 0), ("odd", 37)):
            eng.set_option("p2p_split", 4)
            eng.set_option("simnn_persist", persist)
            res[tag] = {k: _np(v) for k, v in eng.fm_to_p2p(Phi1, Phi2, a1, C).items()}
        eng.reset_options()
        # four waves of 128 x 128 (accumulators in AGPRs) instead of eight of 128 x 64
        eng.set_option("simnn_big", 1)
        res["big"] = {k: _np(v) for k, v in eng.fm_to_p2p(Phi1, Phi2, a1, C).items()}
        eng.reset_options()
        for name in ("knn21", "knn12", "ind21", "ind12"):
            for split in (1, 2, 3, 4, "pertile", "odd", "big"):
                bad = int((res[split][name] != res[0][name]).sum())
                assert bad == 0, (kind, name, split, bad, (B, N1, N2, k1, k2))
        # ... and against the oracle, every kind, EXACTLY: a different entry is accepted only where the two candidates score
        # within a few float64 ulps of each other (two summation orders of the same float64 arithmetic), judged in extended
        # precision, and is counted
        ties = 0
        for b in range(B):
            want = orc.fm_to_p2p_all(C[b], Phi1[b].astype(np.float64), Phi2[b].astype(np.float64), a1[b].astype(np.float64))
            for name, w in zip(("knn21", "knn12", "ind21", "ind12"), want):
                got = res[2][name][b].astype(np.int64)
                for idx in np.nonzero(got != w)[0]:
                    assert _ulp_tie(C[b], Phi1[b], Phi2[b], a1[b], name, int(idx), int(got[idx]), int(w[idx])), \
                        (kind, name, b, int(idx), int(got[idx]), int(w[idx]))
                    ties += 1
        rows = 2 * B * (N1 + N2)
        print(f"{kind} {(B, N1, N2, k1, k2)}: equal to the oracle except {ties} few-ulp ties of {rows} map entries")
        # a ceiling on what the tie rule may forgive: a handful per case is two summation orders meeting a near-tie; a
        # regression that decides rows wrongly by a few ulps en masse must not pass as "ties"
        # ("duplicates" plants exactly equal rows: every planted pair is a candidate tie whose winner is a matter of summation order --
        #  39 of 7680 entries measured; elsewhere a tie is an accident)
        assert ties <= (rows // 50 if kind == "duplicates" else max(4, rows // 500)), (kind, ties, rows)


def test_fuzz_knn_query(eng):
    from hypothesis import given, strategies as st

    @_fuzz_settings()
    @given(st.integers(1, 3), st.integers(1, 700), st.integers(1, 700), st.integers(1, 70), st.sampled_from([1.0, 1e-6, 1e5]),
           st.booleans(), st.integers(0, 2 ** 31 - 1))
    def run(B, nx, ny, p, scale, clustered, seed):
        rng = np.random.default_rng(seed)
        X = rng.standard_normal((B, nx, p)) * scale
        Y = rng.standard_normal((B, ny, p)) * scale
        if clustered and nx > 3:                             # queries on top of tree points, some tree points duplicated
            Y[:, : min(ny, nx)] = X[:, : min(ny, nx)]
            X[:, nx // 2] = X[:, 0]
        got = _np(eng.knn_query(X, Y))
        for b in range(B):
            assert np.array_equal(got[b], orc.knn_query(X[b], Y[b])), (B, nx, ny, p, scale, clustered)
    run()


# --------------------------------------------------------------------------- #
# ICP at the benchmark's size, against the reference's own output (tools/make_golden_r02.py)
def test_icp_config2_size_against_reference(eng, fx_cfg2, fx_cfg2_icp):
    """N = 2048, k = 128, 10 iterations from the reference's C_fit: C within 1e-8 of the reference's icp_refine
    (lstsq + SVD on the host there, normal equations + polar iteration here), the four maps of the refined C bit-exact"""
    fx = fx_cfg2
    k = int(fx["k"])
    Phi1, Phi2 = fx["Phi1"][:, :k].copy(), fx["Phi2"][:, :k].copy()
    C, resid, info = eng.icp(_b(Phi1), _b(Phi2), _b(fx["C_fit"]), nit=10, return_resid=True)
    C = _np(C)[0]
    assert int(_np(info)[0]) == 0 and float(_np(resid)[0]) < 1e-12
    err = np.abs(C - fx_cfg2_icp["C_icp"]).max()
    print("config-2 ICP |C_gpu - C_icp(reference)| =", err)
    assert err < 1e-8
    maps = eng.fm_to_p2p(_b(Phi1), _b(Phi2), _b(fx["a1"]), _b(fx_cfg2_icp["C_icp"]))
    for name in ("knn21", "knn12", "ind21", "ind12"):
        assert np.array_equal(_np(maps[name])[0], fx_cfg2_icp["icp_" + name]), name


def test_icp_polar_schedule_follows_the_conditioning(eng, fx_cfg2):
    """r05: the polar iteration's schedule is per batch and per pair.  From a fitted map (the documented call) a pair needs at most a lift
    or two and a handful of Newton-Schulz steps; from a poor start the batch takes the
    twelve lifts.  Either way C^T C = I to rounding; from the fitted map C equals the oracle's SVD-based icp_refine."""
    fx = fx_cfg2
    k = 40
    P1, P2 = fx["Phi1"][:, :k].astype(np.float64), fx["Phi2"][:, :k].astype(np.float64)
    good = orc.p2p_to_fm(fx["knn21"], P1, P2, fx["a2"])                          # a map that already is close to a vertex map's
    rng = np.random.default_rng(5)
    poor = np.eye(k) + 0.5 * rng.standard_normal((k, k))
    launches = {}
    for name, C0 in (("good", good), ("poor", poor)):
        eng.profile_kernel("*")
        C, resid, info = eng.icp(_b(P1), _b(P2), _b(C0), nit=4, return_resid=True)
        torch.cuda.synchronize()
        rep = eng.profile_report()
        eng.profile_kernel("")
        launches[name] = {n: v[0] for n, v in rep.items() if n.startswith("polar_")}
        assert int(_np(info)[0]) == 0 and float(_np(resid)[0]) < 1e-13
        if name == "good":                                   # (from the poor start the iterates' vertex maps sit on near-ties: not a comparison)
            Co = orc.icp_refine(C0, P1, P2, nit=4)
            assert np.abs(_np(C)[0] - Co).max() < 1e-8
    print("polar launches of 4 ICP iterations:", launches)
    assert launches["good"]["polar_update_nt_f64"] <= 4 * 13                     # (r04: 18 steps per polar factor, always)
    assert launches["good"]["polar_update_nt_f64"] < launches["poor"]["polar_update_nt_f64"]


@pytest.mark.parametrize("k1,k2", [(12, 20), (15, 15), (32, 32), (7, 31), (1, 5)])
def test_icp_small_maps_one_launch_polar(eng, fx_cfg1, k1, k2):
    """maps up to 32 x 32 (the documented call's sizes) take their polar factors in one workgroup launch (polar_small_kernel), square
    and rectangular: C^T C = I to rounding and C equal to the oracle's SVD-based icp_refine; a batch of two gives each pair its own result"""
    fx = fx_cfg1
    P1, P2 = fx["Phi1"][:, :k1].astype(np.float64), fx["Phi2"][:, :k2].astype(np.float64)
    rng = np.random.default_rng(k1 * 37 + k2)
    C0 = np.eye(k2, k1) + 0.05 * rng.standard_normal((k2, k1))
    C1 = np.eye(k2, k1) + 0.3 * rng.standard_normal((k2, k1))
    C, resid, info = eng.icp(np.stack([P1, P1]), np.stack([P2, P2]), np.stack([C0, C1]), nit=3, return_resid=True)
    assert int(_np(info).max()) == 0 and float(_np(resid).max()) < 1e-13
    Co = orc.icp_refine(C0, P1, P2, nit=3)
    assert np.abs(_np(C)[0] - Co).max() < 1e-8
    Cs = eng.icp(_b(P1), _b(P2), _b(C1), nit=3)
    assert np.array_equal(_np(Cs)[0], _np(C)[1])


@pytest.mark.parametrize("k1,k2", [(200, 200), (180, 200), (177, 177), (40, 256)])
def test_icp_large_k(eng, k1, k2):
    """k2 > 176: the Gram matrix is inverted by Newton-Schulz on the matrix cores instead of the in-LDS Cholesky
    (compute_surface_map(n_ev = 200) runs ICP at k = 200, functional_map.py:71)"""
    from densematcher_amd import synth
    N = 1500
    lam1, phi1, a1 = synth.random_basis(N, max(k1, k2), 21)
    lam2, phi2, a2 = synth.random_basis(N, max(k1, k2), 22)
    rng = np.random.default_rng(k1 + k2)
    C0 = np.eye(k2, k1) + 0.02 * rng.standard_normal((k2, k1))
    P1, P2 = phi1[:, :k1].astype(np.float32), phi2[:, :k2].astype(np.float32)
    C, resid, info = eng.icp(_b(P1), _b(P2), _b(C0), nit=3, return_resid=True)
    assert int(_np(info)[0]) == 0
    Co = orc.icp_refine(C0, P1.astype(np.float64), P2.astype(np.float64), nit=3)
    assert np.abs(_np(C)[0] - Co).max() < 1e-8


def test_icp_rank_deficient_map_takes_the_reference_svd(eng, fx_cfg1):
    """A start whose vertex map has fewer distinct images than the map has columns gives a rank-deficient least-squares map: the
    reference's scipy.linalg.svd returns U I V^T for any rank (icp.py:38-40); the polar iteration cannot, so that pair re-runs with
    lstsq + SVD (pyFM/refine/icp.py: icp_host_svd) instead of raising.  The null-space completion is LAPACK's
    choice, so the checks are the properties that define U I V^T: orthonormal columns, and trace(C^T X) = the sum of X's singular
    values for the least-squares map X of the step; where the map has full rank the fallback equals the oracle."""
    import scipy.linalg
    from densematcher_amd.pyFM import refine
    fx = fx_cfg1
    k = 12
    P1, P2 = fx["Phi1"][:, :k].astype(np.float64), fx["Phi2"][:, :k].astype(np.float64)
    C0 = np.zeros((k, k))
    C0[1, 1] = 1e-6                                      # a one-dimensional, tiny embedding: every target picks one of its two extreme sources
    p21 = orc.knn_query(P1 @ C0.T, P2)
    X = scipy.linalg.lstsq(P2, P1[p21])[0]
    sv = scipy.linalg.svd(X, compute_uv=False)
    assert (sv > 1e-9 * sv[0]).sum() < k                 # the case is what it claims to be
    with pytest.warns(UserWarning, match="lstsq \\+ SVD"):
        C = refine.icp_iteration(C0, P1, P2)
    assert np.abs(C.T @ C - np.eye(k)).max() < 1e-10
    assert abs(np.trace(C.T @ X) - sv.sum()) < 1e-9 * max(1.0, sv.sum())
    # full rank: the host path is the oracle's arithmetic
    rng = np.random.default_rng(2)
    C1 = np.eye(k) + 0.05 * rng.standard_normal((k, k))
    assert np.abs(refine.icp.icp_host_svd(C1, P1, P2, 3) - orc.icp_refine(C1, P1, P2, nit=3)).max() < 1e-9


# --------------------------------------------------------------------------- #
# linear assignment (the Hungarian outputs of compute_surface_map)
def test_linear_sum_assignment_equals_scipy(eng):
    """identical assignment to scipy.optimize.linear_sum_assignment (not only the same objective): real costs, integer
    costs with many ties, sparse matrices as the precise map gives, rectangular both ways, both senses, batched"""
    import scipy.optimize
    rng = np.random.default_rng(3)
    for trial, (nr, nc) in enumerate([(1, 1), (5, 9), (9, 5), (64, 64), (200, 230), (230, 200), (500, 500), (700, 512)]):
        mats = []
        for q in range(5):
            c = rng.standard_normal((nr, nc))
            if q == 1:
                c = np.round(3 * c)
            if q == 2:
                c = c * (rng.random((nr, nc)) < 0.02)
            if q == 3 and nc >= 2:                    # two constant columns: interchangeable, the warm start steps aside
                c[:, [0, nc - 1]] = 0.25
            if q == 4 and nr >= 5:                    # a few exact ties in otherwise generic data (warm start, then the rerun)
                c[rng.integers(0, nr, 4), rng.integers(0, nc, 4)] = c[0, 0]
            mats.append(c)
        # every implementation (dm_set_option "lsa_reg"): 2 = register state + column-reduction start, kept only where the
        # optimum is provably unique (the integer / sparse matrices here have ties: they are redone in SciPy's order; a matrix
        # whose column minima are mostly tied, or with two constant columns, skips the warm start altogether),
        # 1 = register state in SciPy's order, 0 = LDS state
        for mode in (2, 1, 0):
            eng.set_option("lsa_reg", mode)
            for mx in (False, True):
                got = _np(eng.linear_sum_assignment(np.stack(mats), maximize=mx))
                for q, c in enumerate(mats):
                    r0, c0 = scipy.optimize.linear_sum_assignment(c, maximize=mx)
                    rows = np.nonzero(got[q] >= 0)[0]
                    assert np.array_equal(rows, r0) and np.array_equal(got[q][rows], c0), (nr, nc, q, mx, mode)
        eng.reset_options()


def test_linear_sum_assignment_rejects_what_scipy_rejects(eng):
    """NaN / wrong-signed infinity -> "invalid numeric entries"; no finite complete assignment -> "infeasible" (SciPy raises
    ValueError in both cases; the kernel reports them in its info array instead of returning a short assignment)"""
    import scipy.optimize
    rng = np.random.default_rng(9)
    good = rng.standard_normal((2, 40, 40))
    assert np.array_equal(_np(eng.linear_sum_assignment(good))[1], scipy.optimize.linear_sum_assignment(good[1])[1])
    for mx, bad_val in ((False, -np.inf), (True, np.inf), (False, np.nan)):
        c = good.copy()
        c[1, 3, 7] = bad_val
        with pytest.raises(ValueError, match="invalid numeric"):
            eng.linear_sum_assignment(c, maximize=mx)
        with pytest.raises(ValueError):
            scipy.optimize.linear_sum_assignment(c[1], maximize=mx)
    c = good.copy()
    c[0, :, :5] = np.inf                                     # ...only 35 usable columns for 40 rows
    c[0, 5:, 5:] = np.inf
    with pytest.raises(ValueError, match="infeasible"):
        eng.linear_sum_assignment(c)
    with pytest.raises(ValueError, match="infeasible"):
        scipy.optimize.linear_sum_assignment(c[0])
    finite_inf = good.copy()
    finite_inf[1, 2, :39] = np.inf                           # +inf entries are fine when minimising as long as an assignment exists
    assert np.array_equal(_np(eng.linear_sum_assignment(finite_inf))[1], scipy.optimize.linear_sum_assignment(finite_inf[1])[1])


def test_precise_map_with_more_candidates_than_the_list_holds(eng):
    """a point far from the embedded surface has every face as a candidate (> 4096 here): the kernel then re-tests all faces
    instead of cutting its candidate list -- same result as the oracle, run to run identical; bad face indices are refused"""
    from densematcher_amd import synth
    nu, nv, k = 72, 40, 6
    v, f = synth.torus_mesh(nu, nv)
    assert f.shape[0] > 4096
    rng = np.random.default_rng(4)
    N = nu * nv
    e1 = (rng.standard_normal((N, k)) * 0.05).astype(np.float32)
    e2 = (rng.standard_normal((200, k)) * 0.05).astype(np.float32)
    e2[:50] += 40.0                                           # far away: Deltamin exceeds every face's bound
    C = np.eye(k)
    fm1, bary1 = eng.precise_map(_b(e1), _b(e2), _b(C), _b(f.astype(np.int32)))
    fm2, bary2 = eng.precise_map(_b(e1), _b(e2), _b(C), _b(f.astype(np.int32)))
    assert np.array_equal(_np(fm1), _np(fm2)) and np.array_equal(_np(bary1), _np(bary2))
    P, fo, bo = orc.precise_map_dense(C, e1.astype(np.float64), e2.astype(np.float64), f)
    d_gpu = np.linalg.norm((_np(bary1)[0][:, :, None] * e1.astype(np.float64)[f[_np(fm1)[0]]]).sum(1) - e2, axis=1)
    d_orc = np.linalg.norm((bo[:, :, None] * e1.astype(np.float64)[f[fo]]).sum(1) - e2, axis=1)
    assert np.abs(d_gpu - d_orc).max() <= 1e-9 * max(1.0, d_orc.max())
    # (in a random embedding most points project onto a vertex or an edge that several faces share at the same distance up to the
    #  last bits: the face NAME is a tie there, the projected point is not)
    print("precise map, > 4096 candidates: same face named for", float((np.asarray(_np(fm1)[0]) == fo).mean()), "of the points; distances equal")
    bad = f.astype(np.int32).copy()
    bad[7, 1] = N
    with pytest.raises(ValueError):
        eng.precise_map(_b(e1), _b(e2), _b(C), _b(bad))


def test_hungarian_of_mapped_indicator(eng, fx_cfg1):
    """functional_map.py:57,78: the assignment of the mapped indicator of the fixture's map equals the reference's
    hungarian_icp output when computed from the reference's ICP map"""
    import scipy.optimize
    fx = fx_cfg1
    k = int(fx["k"])
    M = eng.mapped_indicator(_b(fx["Phi1"][:, :k].copy()), _b(fx["Phi2"][:, :k].copy()), _b(fx["a1"]), _b(fx["csm_FM"]))
    got = _np(eng.linear_sum_assignment(M, maximize=True))[0]
    r0, c0 = scipy.optimize.linear_sum_assignment(_np(M)[0], maximize=True)
    assert np.array_equal(got, c0)
    agree = (got == fx["csm_hungarian_icp_cols"]).mean()
    print("hungarian_icp agreement with the reference tuple:", agree)
    assert agree >= 0.999


def test_assign_many_sequential_equals_batched(eng, monkeypatch):
    """functional_map._assign_many: above its memory threshold the matrices are assigned one after the other (no stacked copy)
    -- same assignments as the batched call"""
    from densematcher_amd import functional_map as fmod
    rng = np.random.default_rng(11)
    mats = [torch.as_tensor(rng.random((90, 90))).to(eng.device) for _ in range(3)]
    batched = fmod._assign_many(mats)
    monkeypatch.setattr(fmod, "_ASSIGN_STACK_LIMIT", 0)
    seq = fmod._assign_many(mats)
    for (r0, c0), (r1, c1) in zip(batched, seq):
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1)


def test_precise_map_and_its_assignment(eng, fx_cfg1, fx_cfg1_precise):
    """dm_precise_map against the reference's get_precise_map (tests/golden/fx_cfg1_precise.npz) and the oracle; the
    assignment of the dense precise map (hungarian_precise, functional_map.py:62-66) against the reference's"""
    fx = fx_cfg1
    k = int(fx["k"])
    P1, P2 = fx["Phi1"][:, :k].copy(), fx["Phi2"][:, :k].copy()
    fm, bary, M = eng.precise_map(_b(P1), _b(P2), _b(fx["C_fit"]), _b(fx["faces1"].astype(np.int32)), dense=True)
    Mo, fmo, baryo = orc.precise_map_dense(fx["C_fit"], P1.astype(np.float64), P2.astype(np.float64), fx["faces1"])
    P = np.zeros_like(Mo)
    P[fx_cfg1_precise["precise_rows"], fx_cfg1_precise["precise_cols"]] = fx_cfg1_precise["precise_vals"]
    Mg = _np(M)[0]
    print("precise map: faces equal to the oracle's:", (_np(fm)[0] == fmo).mean(), " max |M - M_reference| =", np.abs(Mg - P).max())
    assert np.abs(Mg - P).max() < 1e-9 and np.abs(Mg - Mo).max() < 1e-9
    assert np.abs(_np(bary)[0] - baryo)[_np(fm)[0] == fmo].max() < 1e-9
    got = _np(eng.linear_sum_assignment(M, maximize=True))[0]
    import scipy.optimize
    assert np.array_equal(got, scipy.optimize.linear_sum_assignment(Mg, maximize=True)[1])
    agree = (got == fx_cfg1_precise["hungarian_precise_cols"]).mean()
    print("hungarian_precise agreement with the reference:", agree)
    assert agree >= 0.99


# --------------------------------------------------------------------------- #
# the eigenbasis on the GPU
@pytest.mark.parametrize("nu,nv,k", [(25, 20, 30), (64, 32, 128), (64, 32, 200)])
def test_eigenbasis_against_dense_eigh(eng, nu, nv, k):
    """dm_eigenbasis (Chebyshev-filtered subspace iteration) against the dense generalized eigensolver of SciPy on the
    same W, A: eigenvalues to 1e-9 relative, A-orthonormality, the invariant subspace (clusters of equal eigenvalues come
    in an arbitrary basis, so vectors are compared through the projector), residuals"""
    import scipy.linalg
    import scipy.sparse as sps
    from densematcher_amd import synth
    meshes = [synth.torus_mesh(nu, nv, perturb=p, seed=s) for p, s in ((0.0, 0), (0.08, 1))]
    Ws, masses = zip(*[synth.cotan_laplacian(v, f) for v, f in meshes])
    masses = [m_.astype(np.float32).astype(np.float64) for m_ in masses]          # masses cross the ABI as fp32
    lam, Phi, resid, rounds = eng.eigenbasis(list(Ws), np.stack(masses), k, tol=1e-10)
    lam, Phi = _np(lam), _np(Phi)
    print(f"N={nu * nv} k={k}: {rounds} rounds, residual {float(resid.max()):.2e}")
    for b in range(2):
        w, V = scipy.linalg.eigh(Ws[b].toarray(), np.diag(masses[b]))
        scale = w[k - 1]
        assert np.abs(lam[b] - w[:k]).max() <= 1e-9 * scale
        A = sps.diags(masses[b])
        G = Phi[b].T @ (A @ Phi[b])
        assert np.abs(G - np.eye(k)).max() <= 1e-9
        R = Ws[b] @ Phi[b] - (A @ Phi[b]) * lam[b][None, :]
        assert np.abs(R).max() <= 1e-7 * scale
        # the invariant subspace below the last cluster boundary inside the first k values
        cut = k
        while cut > 1 and (w[cut] - w[cut - 1]) <= 1e-6 * scale:
            cut -= 1                                           # (do not cut a cluster of equal eigenvalues)
        P = V[:, :cut].T @ (A @ Phi[b][:, :cut])
        assert np.abs(P.T @ P - np.eye(cut)).max() <= 1e-7


def test_eigenbasis_small_mesh_beside_a_large_one(eng):
    """TriMesh.process_many on meshes of different vertex counts.  N = 700 beside 1200, k = 128: one batched call, the
    small mesh padded with decoupled vertices at the Gershgorin bound of its own operator (the top of the damped interval; the
    largest diagonal entry used before ranks near the middle of the spectrum).  N = 192, k = 128 > N / 2: the subspace iteration
    cannot resolve the upper half of a spectrum (r04 refused such meshes) -- the meshes are then solved one by one and the small
    one takes the dense route (r05: ARPACK in the reference handles any k < N, laplacian.py:165)."""
    import scipy.linalg
    from densematcher_amd import synth
    from densematcher_amd.pyFM.mesh import TriMesh
    k = 128
    meshes = [TriMesh(*synth.torus_mesh(28, 25)), TriMesh(*synth.torus_mesh(40, 30, perturb=0.05, seed=2))]
    TriMesh.process_many(meshes, [k, k])
    for m in meshes:
        n = m.n_vertices
        a = m.A.diagonal()
        w = scipy.linalg.eigh(m.W.toarray(), np.diag(a), eigvals_only=True)
        assert m.eigenvectors.shape == (n, k)
        assert np.abs(m.eigenvalues - w[:k]).max() <= 1e-6 * w[k - 1], n
        G = m.eigenvectors.T @ (a[:, None] * m.eigenvectors)
        assert np.abs(G - np.eye(k)).max() <= 1e-6, n
        R = m.W @ m.eigenvectors - (a[:, None] * m.eigenvectors) * m.eigenvalues[None, :]
        assert np.abs(R).max() <= 1e-5 * w[k - 1], n
    pair = [TriMesh(*synth.torus_mesh(16, 12)), TriMesh(*synth.torus_mesh(40, 30, perturb=0.05, seed=2))]
    TriMesh.process_many(pair, [k, k])
    for m in pair:
        a = m.A.diagonal().astype(np.float32).astype(np.float64)
        w = scipy.linalg.eigh(m.W.toarray(), np.diag(a), eigvals_only=True)
        assert m.eigenvectors.shape == (m.n_vertices, k)
        assert np.abs(m.eigenvalues - w[:k]).max() <= 1e-6 * w[k - 1], m.n_vertices
    # r06: the dense route takes meshes up to 2048 vertices (600 vertices, k = 290: test_small_meshes_take_the_dense_route); beyond
    # that the refusal stays (k + guard vectors cannot sit in the lower half of the spectrum)
    with pytest.raises(ValueError, match="dense route"):
        TriMesh(*synth.torus_mesh(60, 40)).process(k=1190)


def test_maps_on_gpu_eigenbasis_match_maps_on_host_eigenbasis(eng):
    """end to end: a pair matched on the GPU-made bases gives the same vertex maps as on SciPy's dense bases (the functional
    map itself is basis dependent inside clusters of equal eigenvalues, the maps are not)"""
    import scipy.linalg
    from densematcher_amd import synth
    nu, nv, D = 32, 24, 96
    (v1, f1), (v2, f2) = synth.torus_mesh(nu, nv, perturb=0.05, seed=3), synth.torus_mesh(nu, nv, perturb=0.08, seed=1)
    (W1, m1), (W2, m2) = synth.cotan_laplacian(v1, f1), synth.cotan_laplacian(v2, f2)
    m1, m2 = m1.astype(np.float32).astype(np.float64), m2.astype(np.float32).astype(np.float64)
    # truncate where neither spectrum has a near-multiple eigenvalue across the cut (span of the first k is then unique)
    w1 = scipy.linalg.eigh(W1.toarray(), np.diag(m1), eigvals_only=True)
    w2 = scipy.linalg.eigh(W2.toarray(), np.diag(m2), eigvals_only=True)
    k = max(range(34, 46), key=lambda q: min(w1[q] - w1[q - 1], w2[q] - w2[q - 1]))
    lam, Phi, resid, _ = eng.eigenbasis([W1, W2], np.stack([m1, m2]), k, tol=1e-11)
    lam, Phi = _np(lam), _np(Phi)
    F1, F2, _ = synth.feature_pair(nu * nv, nu * nv, D, 5, 6, sigma=0.3, perm="identity")
    res = {}
    for name in ("gpu", "host"):
        if name == "gpu":
            b1, b2, l1, l2 = Phi[0], Phi[1], lam[0], lam[1]
        else:
            l1, b1 = scipy.linalg.eigh(W1.toarray(), np.diag(m1), subset_by_index=[0, k - 1])
            l2, b2 = scipy.linalg.eigh(W2.toarray(), np.diag(m2), subset_by_index=[0, k - 1])
        batch = {"Phi1": b1.astype(np.float32)[None], "Phi2": b2.astype(np.float32)[None], "lam1": l1[None], "lam2": l2[None],
                 "a1": m1.astype(np.float32)[None], "a2": m2.astype(np.float32)[None], "F1": F1[None], "F2": F2[None]}
        out = eng.match({n: torch.as_tensor(v_).to(eng.device) for n, v_ in batch.items()}, k=k)
        res[name] = {n: _np(out[n])[0] for n in ("knn21", "knn12", "ind21", "ind12")}
    agree = {n: float((res["gpu"][n] == res["host"][n]).mean()) for n in res["gpu"]}
    print("maps on GPU basis vs host basis:", agree)
    assert min(agree.values()) >= 0.99


def test_engine_refuses_a_foreign_stream(eng):
    """the context launches on the stream it was created on: calling it under another torch stream would let the
    caching allocator recycle temporaries while kernels still read them"""
    import torch
    X = np.random.default_rng(0).standard_normal((1, 300, 5))
    other = torch.cuda.Stream()
    with torch.cuda.stream(other):
        with pytest.raises(RuntimeError, match="bound to the stream"):
            eng.knn_query(X, X)
    assert np.array_equal(_np(eng.knn_query(X, X))[0], np.arange(300))


@pytest.mark.gpu
def test_eigenbasis_of_meshes_of_different_sizes_in_one_call(eng):
    """FunctionalMapping.preprocess solves its two meshes in one batched call (TriMesh.process_many); meshes of different
    vertex counts are padded with decoupled vertices: every mesh gets the eigenpairs of the dense solver on ITS OWN W, A
    (eigenvalues 1e-9 relative, A-orthonormal vectors, residuals, the same invariant subspace), and the padding rows are ~0"""
    import scipy.linalg
    import scipy.sparse as sps
    from densematcher_amd import synth
    from densematcher_amd.pyFM.mesh import TriMesh
    k = 30
    meshes = [synth.torus_mesh(32, 20, perturb=0.05, seed=2), synth.torus_mesh(25, 20, perturb=0.08, seed=1), synth.torus_mesh(30, 17, perturb=0.03, seed=5)]
    Ws, masses = zip(*[synth.cotan_laplacian(v, f) for v, f in meshes])
    masses = [m_.astype(np.float32).astype(np.float64) for m_ in masses]
    lam, Phi, resid, _ = eng.eigenbasis(list(Ws), list(masses), k, tol=1e-10)
    lam, Phi = _np(lam), _np(Phi)
    assert Phi.shape == (3, 640, k)
    for b in range(3):
        n = masses[b].shape[0]
        w, V = scipy.linalg.eigh(Ws[b].toarray(), np.diag(masses[b]))
        scale = w[k - 1]
        assert np.abs(lam[b] - w[:k]).max() <= 1e-9 * scale, b
        assert np.abs(Phi[b, n:]).max(initial=0.0) <= 1e-9, b
        A = sps.diags(masses[b])
        Pb = Phi[b, :n]
        assert np.abs(Pb.T @ (A @ Pb) - np.eye(k)).max() <= 1e-9
        assert np.abs(Ws[b] @ Pb - (A @ Pb) * lam[b][None, :]).max() <= 1e-7 * scale
        cut = k
        while cut > 1 and (w[cut] - w[cut - 1]) <= 1e-6 * scale:
            cut -= 1
        P = V[:, :cut].T @ (A @ Pb[:, :cut])
        assert np.abs(P.T @ P - np.eye(cut)).max() <= 1e-7
    # the model-level entry: both meshes processed by one call, each keeps its own k
    import warnings
    m1, m2 = TriMesh(*meshes[0]), TriMesh(*meshes[1])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        TriMesh.process_many([m1, m2], [12, 25], robust=False)
    assert m1.eigenvectors.shape == (640, 12) and m2.eigenvectors.shape == (500, 25)
    for mesh, b, kk in ((m1, 0, 12), (m2, 1, 25)):
        w = scipy.linalg.eigh(Ws[b].toarray(), np.diag(masses[b]), eigvals_only=True, subset_by_index=[0, kk - 1])
        assert np.abs(mesh.eigenvalues - w).max() <= 1e-8 * max(w[-1], 1.0)






def test_fm_to_p2p_basis_hint_is_checked_not_trusted(eng, fx_cfg2):
    """r06: the four-map path keeps, per basis tensor, the maxima of |Phi2| a call measured; the next call on the same
    tensor lets the second embedding write the split target rows as the basis streams by (no fm_split_build_rows pass).  The hint is
    CHECKED against what the call measures: a tensor rewritten in place with another scale (or another tensor at the same address)
    sends that pair's rows to the exact path -- the maps stay the float64 arg-reductions either way, and equal the hint-less path's."""
    fx = fx_cfg2
    k = int(fx["k"])
    P1 = torch.as_tensor(np.stack([fx["Phi1"][:, :k]] * 2)).to(eng.device)
    P2 = torch.as_tensor(np.stack([fx["Phi2"][:, :k]] * 2)).to(eng.device)
    a1 = torch.as_tensor(np.stack([fx["a1"]] * 2)).to(eng.device)
    C = torch.as_tensor(np.stack([fx["C_f64"], fx["C_f64"].T.copy()])).to(eng.device)
    names = ("knn21", "knn12", "ind21", "ind12")
    eng.set_option("basis_stats", 0)
    ref = {n: v.cpu().numpy() for n, v in eng.fm_to_p2p(P1, P2, a1, C).items()}
    eng.set_option("basis_stats", 1)                       # (also forgets every stored hint)
    launches = []
    for rep in range(3):
        eng.profile_kernel("*")
        out = eng.fm_to_p2p(P1, P2, a1, C)
        torch.cuda.synchronize()
        launches.append({n: v[0] for n, v in eng.profile_report().items()})
        eng.profile_kernel("")
        for n in names:
            assert np.array_equal(out[n].cpu().numpy(), ref[n]), (rep, n)
    assert launches[0].get("fm_split_build_rows", 0) == 1 and launches[1].get("fm_split_build_rows", 0) == 0 and launches[2].get("fm_split_build_rows", 0) == 0
    flagged_ok = eng.last_requeued_rows()
    # the SOURCE rows ride on the first embedding with the scale of max |Phi1 C^T| of the previous call on the same Phi1 tensor; a map of
    # another magnitude (pair 1: C x 64, another binade) gets that pair's rows rebuilt by knn_split_build -- same results as without hints
    C2 = C.clone()
    C2[1] *= 64.0
    out2 = eng.fm_to_p2p(P1, P2, a1, C2)
    eng.set_option("basis_stats", 0)
    want2 = {n: v.cpu().numpy() for n, v in eng.fm_to_p2p(P1, P2, a1, C2).items()}
    eng.set_option("basis_stats", 1)
    for n in names:
        assert np.array_equal(out2[n].cpu().numpy(), want2[n]), n
    for rep in range(2):                                   # (refill the hints for the part below)
        out = eng.fm_to_p2p(P1, P2, a1, C)
        for n in names:
            assert np.array_equal(out[n].cpu().numpy(), ref[n]), n
    # the same tensor, rewritten in place: pair 1's basis scaled by 2^-7 (another binade: the hint is wrong for it), pair 0 untouched
    P2[1] *= 2.0 ** -7
    out = eng.fm_to_p2p(P1, P2, a1, C)
    flagged_bad = eng.last_requeued_rows()
    eng.set_option("basis_stats", 0)
    want = {n: v.cpu().numpy() for n, v in eng.fm_to_p2p(P1, P2, a1, C).items()}
    eng.set_option("basis_stats", 1)
    for n in names:
        assert np.array_equal(out[n].cpu().numpy(), want[n]), n
    assert flagged_bad[0] >= P2.shape[1] and flagged_ok[0] < P2.shape[1] // 4          # pair 1 went through the exact path whole
    # ... and the call after that has the right hint again
    out = eng.fm_to_p2p(P1, P2, a1, C)
    assert eng.last_requeued_rows()[0] < P2.shape[1] // 4
    for n in names:
        assert np.array_equal(out[n].cpu().numpy(), want[n]), n


@pytest.mark.parametrize("k,N", [(128, 640), (100, 512), (68, 400), (200, 900), (150, 700)])
def test_solver_batched_pcg_equals_direct(eng, k, N):
    """r06: the k2 systems of a pair by the batched Jacobi-preconditioned conjugate-gradient iteration on the float64 matrix cores
    (csrc/dm_pcg.h; orders 65 .. 128 with the matrix in registers, 129 .. 199 with streamed fragments) against the direct solvers
    (dm_set_option solve_pcg = 0): the maps agree to 1e-9, the iteration's kernel ran, and rank-deficient descriptors -- which it cannot
    finish in its step budget -- come back from the direct solver bit for bit (the fall-back launch)"""
    from densematcher_amd import synth
    B, D = 3, 96
    bases = [synth.random_basis(N, k, 40 + q) for q in range(2 * B)]
    st = lambda q0, j: np.stack([bases[2 * b + q0][j] for b in range(B)])
    lam1, lam2, P1, P2, a1, a2 = st(0, 0), st(1, 0), st(0, 1).astype(np.float32), st(1, 1).astype(np.float32), st(0, 2).astype(np.float32), st(1, 2).astype(np.float32)
    F = [synth.feature_pair(N, N, D, 70 + b, 80 + b, sigma=0.3, perm="identity") for b in range(B)]
    F1, F2 = np.stack([f[0] for f in F]), np.stack([f[1] for f in F])

    def fit(pcg, F1_, F2_):
        eng.set_option("solve_pcg", pcg)               # (2: the iteration whatever the batch size; 1 leaves small batches to the direct solvers)
        eng.profile_kernel("*")
        C = eng.fmap_fit(P1, P2, a1, a2, F1_, F2_, lam1, lam2, 1e4, 1e3, k1=k, k2=k).cpu().numpy()
        names = set(eng.profile_report())
        eng.profile_kernel("")
        return C, names
    try:
        Cd, nd = fit(0, F1, F2)
        Cp, npcg = fit(1, F1, F2)
        assert "fmap_solve_pcg" in npcg and "fmap_solve_pcg" not in nd
        err = np.abs(Cp - Cd).max()
        print(f"k = {k}: max |C_pcg - C_direct| = {err:.2e}")
        assert err <= 1e-9 * max(1.0, np.abs(Cd).max())
        for b in range(B):
            Co = orc.fit(P1[b], P2[b], lam1[b], lam2[b], a1[b], a2[b], F1[b], F2[b], 1e4, 1e3)
            assert np.abs(Cp[b] - Co).max() <= 1e-4
        # 20 distinct channels: P = A A^T has rank 20 < k - 1, the systems are held up by the Laplacian term alone (cond ~ 1e6)
        Fr1 = np.tile(F1[:, :, :20], (1, 1, 5))[:, :, :D].copy()
        Fr2 = np.tile(F2[:, :, :20], (1, 1, 5))[:, :, :D].copy()
        Cd, _ = fit(0, Fr1, Fr2)
        Cp, npcg = fit(1, Fr1, Fr2)
        same = np.array_equal(Cp, Cd)
        print(f"k = {k}, rank-20 descriptors: bit-identical to the direct solver (fall-back): {same}; max diff {np.abs(Cp - Cd).max():.2e}")
        assert "fmap_solve_chol" in npcg                                    # the fall-back launch is always there
        assert same or np.abs(Cp - Cd).max() <= 1e-7 * max(1.0, np.abs(Cd).max())
    finally:
        eng.set_option("solve_pcg", 1)

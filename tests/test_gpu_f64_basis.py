"""
GPU (-m gpu): parity on the reference's REAL inputs -- float64 eigenvectors and masses (VERDICT r02 "next" #1).

The reference runs FM_to_p2p / p2p_to_FM / ICP / ZoomOut on float64 eigenvectors (pyFM/spectral/convert.py:134-144,
refine/icp.py:36-40).  The `*_f64` entry points of the C ABI take them unrounded; these tests pin their integer outputs
bit-exact to the reference's own outputs on an un-rounded spectrum (tests/golden/fx_cfg2_f64.npz, tools/make_golden_r03.py)
and, on adversarial operands whose float32 rounding creates or breaks ties, to the float64 oracle -- with the disagreement
rate of the float32 entry points printed beside them.
"""
import numpy as np
import pytest

from oracle import dm_oracle as orc

pytestmark = pytest.mark.gpu
MAPS = ("knn21", "knn12", "ind21", "ind12")


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


def _b(x):
    return np.ascontiguousarray(x)[None]


def test_reference_outputs_on_unrounded_float64_basis(eng, fx_cfg2_f64):
    fx = fx_cfg2_f64
    k = int(fx["k"])
    P1, P2, a1, a2 = fx["Phi1"], fx["Phi2"], fx["a1"], fx["a2"]
    assert P1.dtype == np.float64 and a1.dtype == np.float64
    assert np.abs(P1 - P1.astype(np.float32)).max() > 0            # the fixture really is un-rounded

    # ---- the four maps of the reference's own C_fit, every code path of dm_fm_to_p2p_f64
    for split in (2, 1, 0):
        eng.set_option("p2p_split", split)
        out = eng.fm_to_p2p(_b(P1[:, :k]), _b(P2[:, :k]), _b(a1), _b(fx["C_fit"]))
        for name in MAPS:
            got = _np(out[name])[0].astype(np.int64)
            assert np.array_equal(got, fx[name]), f"p2p_split={split} {name}: {(got != fx[name]).sum()} mismatches vs the reference"
    eng.reset_options()
    # leading dimension > k (the stored spectrum has 136 columns)
    out = eng.fm_to_p2p(_b(P1), _b(P2), _b(a1), _b(fx["C_fit"]))
    for name in MAPS:
        assert np.array_equal(_np(out[name])[0], fx[name]), name
    # the float32 entry points on the rounded basis: how often does the boundary dtype change a map entry?
    out32 = eng.fm_to_p2p(_b(P1.astype(np.float32)), _b(P2.astype(np.float32)), _b(a1.astype(np.float32)), _b(fx["C_fit"]))
    rate = {name: float((_np(out32[name])[0] != fx[name]).mean()) for name in MAPS}
    print("fx_cfg2_f64: disagreement of the fp32-basis path with the reference's float64 maps:", rate)
    for name in MAPS:                                              # ... and it equals what the reference gives on that rounded basis
        assert np.array_equal(_np(out32[name])[0], fx[name + "_r32"]), name

    # ---- p2p_to_FM, both forms
    C = _np(eng.p2p_to_fm(_b(fx["knn21"].astype(np.int32)), _b(P1[:, :k]), _b(P2[:, :k]), _b(a2), k, k))[0]
    assert np.abs(C - fx["C_from_p2p"]).max() <= 1e-12
    Cl = _np(eng.p2p_to_fm_lstsq(_b(fx["knn21"].astype(np.int32)), _b(P1[:, :k]), _b(P2[:, :k]), k, k))[0]
    assert np.abs(Cl - fx["C_from_p2p_lstsq"]).max() <= 1e-9

    # ---- ICP: C within 1e-8 of the reference's icp_refine, the maps of the refined C bit-exact
    Ci, resid, info = eng.icp(_b(P1[:, :k]), _b(P2[:, :k]), _b(fx["C_fit"]), nit=10, return_resid=True)
    assert int(_np(info)[0]) == 0 and float(_np(resid)[0]) < 1e-12
    err = np.abs(_np(Ci)[0] - fx["C_icp"]).max()
    print("fx_cfg2_f64: |C_icp(gpu, float64 basis) - C_icp(reference)| =", err)
    assert err < 1e-8
    out = eng.fm_to_p2p(_b(P1[:, :k]), _b(P2[:, :k]), _b(a1), _b(fx["C_icp"]))
    for name in MAPS:
        assert np.array_equal(_np(out[name])[0], fx["icp_" + name]), name

    # ---- ZoomOut 128 -> 136, step 4
    Cz, pz = eng.zoomout(_b(P1), _b(P2), _b(a2), _b(fx["C_fit"]), nit=2, step=4, return_p2p=True)
    assert np.array_equal(_np(pz)[0], fx["p21_zo"])
    assert np.abs(_np(Cz)[0] - fx["C_zo"]).max() <= 1e-10

    # ---- dense indicator (convert.py:144) and the pinned entry of x0 from float64 masses
    M = _np(eng.mapped_indicator(_b(P1[:, :k]), _b(P2[:, :k]), _b(a1), _b(fx["C_fit"])))[0]
    Mo = (P2[:, :k] @ fx["C_fit"] @ P1[:, :k].T) * a1[None, :]
    assert np.abs(M - Mo).max() <= 1e-12 * max(1.0, np.abs(Mo).max())
    c00 = float(_np(eng.c00(_b(P1), _b(P2), _b(a1), _b(a2)))[0])
    assert abs(c00 - fx["x0_col0"][0]) <= 1e-15 * abs(fx["x0_col0"][0])


def test_mirror_passes_float64_through(fx_cfg2_f64):
    """spectral.FM_to_p2p / refine.* called the way the reference is called (float64 eigenvectors, sparse float64 A1)"""
    import scipy.sparse as sp
    from densematcher_amd.pyFM import refine, spectral
    fx = fx_cfg2_f64
    k = int(fx["k"])
    A1, A2 = sp.diags(fx["a1"]).tocsr(), sp.diags(fx["a2"]).tocsr()
    p21, p12, ind = spectral.FM_to_p2p(fx["C_fit"], fx["Phi1"][:, :k], fx["Phi2"][:, :k], A1)
    assert np.array_equal(p21, fx["knn21"]) and np.array_equal(p12, fx["knn12"])
    assert np.array_equal(ind.argmax(axis=1), fx["ind21"]) and np.array_equal(ind.argmax(axis=0), fx["ind12"])
    assert np.abs(spectral.p2p_to_FM(fx["knn21"], fx["Phi1"][:, :k], fx["Phi2"][:, :k], A2=A2) - fx["C_from_p2p"]).max() <= 1e-12
    C_icp = refine.icp_refine(fx["C_fit"], fx["Phi1"][:, :k], fx["Phi2"][:, :k], A1, nit=10)
    assert np.abs(C_icp - fx["C_icp"]).max() < 1e-8
    C_zo, p_zo = refine.zoomout_refine(fx["C_fit"], fx["Phi1"], fx["Phi2"], nit=2, step=4, A2=A2, return_p2p=True)
    assert np.array_equal(p_zo, fx["p21_zo"]) and np.abs(C_zo - fx["C_zo"]).max() <= 1e-10


# --------------------------------------------------------------------------- #
def _adversarial(rng, B, N1, N2, k1, k2, kind):
    """float64 operands whose float32 rounding changes the answer: rows that differ by 1e-10 .. 1e-8 relative (distinct in
    float64, equal or re-ordered after rounding), masses that differ below the float32 resolution"""
    Phi1 = rng.standard_normal((B, N1, k1))
    Phi2 = rng.standard_normal((B, N2, k2))
    C = rng.standard_normal((B, k2, k1)) / np.sqrt(k1)
    a1 = rng.uniform(0.5, 1.5, (B, N1))
    if kind == "near_duplicates":          # copies of rows perturbed far below the fp32 resolution: ties only after rounding
        Phi1[:, N1 // 2:N1 // 2 + 60] = Phi1[:, :60] * (1.0 + 1e-10 * rng.standard_normal((B, 60, 1)))
        Phi2[:, 100:160] = Phi2[:, 300:360] + 1e-11 * rng.standard_normal((B, 60, k2))
        a1[:, N1 // 2:N1 // 2 + 60] = a1[:, :60]
    if kind == "near_masses":              # equal indicator rows up to the mass: masses 1e-9 apart
        Phi1[:, 200:260] = Phi1[:, :60]
        a1[:, 200:260] = a1[:, :60] * (1.0 + 1e-9 * rng.uniform(-1, 1, (B, 60)))
    if kind == "permuted":                 # a true correspondence with noise below the fp32 resolution of the entries
        for b in range(B):
            perm = rng.permutation(N1)[:N2] if N2 <= N1 else rng.integers(0, N1, N2)
            km = min(k1, k2)
            Phi2[b][:, :km] = Phi1[b][perm][:, :km] + 1e-9 * rng.standard_normal((N2, km))
        C = np.eye(k2, k1)[None].repeat(B, axis=0) + 1e-3 * rng.standard_normal((B, k2, k1))
    if kind == "scales":
        Phi1 *= 1e-3 * 0.97 ** np.arange(k1)
        Phi2 *= 2e2 * 0.95 ** np.arange(k2)
    if kind == "zero_masses":
        a1 = 10.0 ** rng.uniform(-5, 0, (B, N1))
        a1[:, ::97] = 0.0
    return Phi1, Phi2, a1, C


def _ulp_tie(C, P1, P2, a1, name, idx, got, want):
    """both candidates of a mismatching entry score within a few ulps of each other in float64 (a genuine rounding-order tie
    between two summation orders of the same float64 arithmetic), judged with extended precision"""
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
        return float(g * ld(a1[j])), float(abs(g) * a1[j])
    if name in ("knn21", "ind21"):
        (sg, sc), (sw, _) = score(idx, got), score(idx, want)
    else:
        (sg, sc), (sw, _) = score(got, idx), score(want, idx)
    return abs(sg - sw) <= 64 * np.finfo(np.float64).eps * max(sc, 1e-300)


@pytest.mark.parametrize("kind", ["random", "near_duplicates", "near_masses", "permuted", "scales", "zero_masses"])
def test_float64_basis_adversarial_equals_oracle(eng, kind):
    """float64 operands at split-path sizes: every code path of dm_fm_to_p2p_f64 equals the float64 oracle EXACTLY (a
    mismatch is tolerated only where both candidates tie to a few float64 ulps, and is printed); the float32 entry points
    on the rounded operands do not -- their disagreement rate is printed"""
    rng = np.random.default_rng({"random": 11, "near_duplicates": 12, "near_masses": 13, "permuted": 14, "scales": 15, "zero_masses": 16}[kind])
    for (B, N1, N2, k1, k2) in ((2, 512, 768, 64, 80), (1, 1024, 512, 72, 100), (1, 777, 1000, 72, 90)):
        Phi1, Phi2, a1, C = _adversarial(rng, B, N1, N2, k1, k2, kind)
        assert eng.p2p_split_active(N2, N1, k2)
        want = [orc.fm_to_p2p_all(C[b], Phi1[b], Phi2[b], a1[b]) for b in range(B)]
        res = {}
        for split in (2, 1, 0):
            eng.set_option("p2p_split", split)
            res[split] = {n: _np(v) for n, v in eng.fm_to_p2p(Phi1, Phi2, a1, C).items()}
        eng.reset_options()
        for name in MAPS:
            assert np.array_equal(res[2][name], res[0][name]) and np.array_equal(res[1][name], res[0][name]), (kind, name)
        ties = 0
        for b in range(B):
            for name, w in zip(MAPS, want[b]):
                got = res[2][name][b].astype(np.int64)
                for idx in np.nonzero(got != w)[0]:
                    assert _ulp_tie(C[b], Phi1[b], Phi2[b], a1[b], name, int(idx), int(got[idx]), int(w[idx])), \
                        (kind, name, b, int(idx), int(got[idx]), int(w[idx]))
                    ties += 1
        out32 = {n: _np(v) for n, v in eng.fm_to_p2p(Phi1.astype(np.float32), Phi2.astype(np.float32), a1.astype(np.float32), C).items()}
        rate = {name: float(np.mean([(out32[name][b] != want[b][i]).mean() for b in range(B)])) for i, name in enumerate(MAPS)}
        print(f"{kind} {(B, N1, N2, k1, k2)}: float64 path vs oracle: {ties} few-ulp ties, else equal; fp32-rounded path disagrees on", rate)


@pytest.mark.parametrize("N1,N2,k1,k2,ld", [(300, 517, 20, 33, 40), (129, 128, 17, 16, 17), (1000, 777, 64, 50, 65)])
def test_float64_basis_ragged(eng, N1, N2, k1, k2, ld):
    """odd sizes, odd leading dimensions (unaligned rows: the scalar load path), rectangular maps: float64 G kernel"""
    rng = np.random.default_rng(N1 + 7 * N2)
    B = 2
    Phi1 = rng.standard_normal((B, N1, ld)) * 0.05
    Phi2 = rng.standard_normal((B, N2, ld)) * 0.05
    a1 = rng.uniform(0.5, 1.5, (B, N1)) / N1
    a2 = rng.uniform(0.5, 1.5, (B, N2)) / N2
    C = rng.standard_normal((B, k2, k1))
    out = eng.fm_to_p2p(Phi1, Phi2, a1, C)
    for b in range(B):
        ref = orc.fm_to_p2p_all(C[b], Phi1[b], Phi2[b], a1[b])
        for name, r in zip(MAPS, ref):
            assert np.array_equal(_np(out[name])[b], r), (name, b)
        Cg = _np(eng.p2p_to_fm(out["knn21"], Phi1, Phi2, a2, k1, k2))[b]
        Co = orc.p2p_to_fm(ref[0], Phi1[b][:, :k1], Phi2[b][:, :k2], a2[b])
        assert np.abs(Cg - Co).max() <= 1e-13 * max(1.0, np.abs(Co).max())
    # ZoomOut and ICP on a float64 basis against the oracle
    k0 = min(k1, k2) // 2
    C0 = np.eye(k0)[None].repeat(B, axis=0)
    Cz, pz = eng.zoomout(Phi1, Phi2, a2, C0, nit=3, step=2, return_p2p=True)
    for b in range(B):
        Co, po = orc.zoomout_refine(C0[b], Phi1[b], Phi2[b], 3, step=2, a2=a2[b], return_p2p=True)
        assert np.array_equal(_np(pz)[b], po) and np.abs(_np(Cz)[b] - Co).max() <= 1e-11 * max(1.0, np.abs(Co).max())

"""
GPU (-m gpu): dm_fmap_fit_fused -- the iterative fit of maps up to 32 x 32 with one launch per evaluation (the notebook's call,
reference pyFM/functional.py:352-487 + pyFM/optimize/base_functions.py:296-428) against the oracle's energy / gradient, the
multi-launch path, and itself across batch sizes (unit mode, chunk mode).
"""
import numpy as np
import pytest

from oracle import dm_oracle as orc

pytestmark = pytest.mark.gpu

NOTEBOOK_W = dict(w_descr=1e4, w_lap=1e3, w_ent=1e-1, w_sumto1=1e1)


def _random_problem(rng, B, N1, N2, k1, k2, D=8, scale=3.0):
    e1 = (rng.standard_normal((B, N1, k1)) / np.sqrt(N1)).astype(np.float32)
    e2 = (rng.standard_normal((B, N2, k2)) / np.sqrt(N2)).astype(np.float32)
    a1 = (rng.uniform(0.5, 1.5, (B, N1)) / N1).astype(np.float32)
    C = rng.standard_normal((B, k2, k1)) * scale
    A = rng.standard_normal((B, k1, D)).astype(np.float32)
    Bm = rng.standard_normal((B, k2, D)).astype(np.float32)
    lam1, lam2 = np.sort(rng.uniform(0, 50, (B, k1)), axis=1), np.sort(rng.uniform(0, 50, (B, k2)), axis=1)
    return e1, e2, a1, C, A, Bm, lam1, lam2


@pytest.mark.parametrize("N1,N2,k1,k2", [(300, 517, 15, 13), (1000, 777, 20, 30), (129, 65, 32, 32), (640, 576, 7, 18), (2048, 2048, 15, 15)])
def test_fused_energy_and_gradient_against_oracle(N1, N2, k1, k2):
    """one evaluation of the fused kernel (rows of the indicator per lane, scalar-cache operand rows, in-line log, the sum-to-one term
    as a quadratic form of centred Gram matrices) against the oracle's float64 energy_grad_general: every supported term alone
    and together, sizes that are multiples of nothing, rectangular maps, all register-tile instantiations"""
    from densematcher_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(N1 + 3 * N2 + k1)
    B = 2
    e1, e2, a1, C, A, Bm, lam1, lam2 = _random_problem(rng, B, N1, N2, k1, k2)
    # entries of the indicator on both sides of the clamp's ends: scale a pair's map up
    C[1] *= 40.0
    for w in ({"w_ent": 0.3}, {"w_sumto1": 2.0}, {"w_p2p": 0.5, "w_range01": 1.5}, dict(NOTEBOOK_W),
              {"w_p2p": 0.5, "w_ent": 0.3, "w_range01": 1.5, "w_sumto1": 2.0, "w_descr": 1.0, "w_lap": 0.1}):
        assert eng.fit_fused_ok(k1, k2, w)
        E, G = eng.energy_grad_fused(C, A, Bm, lam1, lam2, w, e1, e2, a1)
        for b in range(B):
            ev = orc.ev_sqdiff(lam1[b], lam2[b])
            Eo, Go = orc.energy_grad_general(C[b], A[b].astype(np.float64), Bm[b].astype(np.float64), ev, e1[b], e2[b], a1[b], w)
            assert abs(float(E[b]) - Eo) <= 1e-10 * abs(Eo), (w, b, float(E[b]), Eo)
            assert np.abs(G[b].cpu().numpy() - Go).max() <= 1e-10 * np.abs(Go).max(), (w, b, np.abs(G[b].cpu().numpy() - Go).max(), np.abs(Go).max())
            assert np.all(G[b].cpu().numpy()[:, 0] == 0)


def test_fused_is_refused_for_other_terms_and_sizes():
    from densematcher_amd.engine import default_engine
    eng = default_engine()
    assert not eng.fit_fused_ok(33, 15, NOTEBOOK_W)
    assert not eng.fit_fused_ok(15, 40, NOTEBOOK_W)
    assert not eng.fit_fused_ok(15, 15, dict(NOTEBOOK_W, w_stochastic=1.0))
    assert not eng.fit_fused_ok(15, 15, dict(NOTEBOOK_W, w_area=1.0))
    assert not eng.fit_fused_ok(15, 15, dict(w_descr=1.0, w_lap=1.0))           # closed form: nothing to iterate on
    assert eng.fit_fused_ok(15, 15, dict(NOTEBOOK_W, w_dcomm=1.0), None)      # (no operators: the term is off)


def _fit_batch(fx, k, B, rng):
    F2s = []
    for b in range(B):
        F2 = fx["F2"].copy()
        if b % 3:
            F2 = F2[:, np.random.default_rng(10 + b % 3).permutation(F2.shape[1])]
        F2s.append(F2)
    st = lambda x: np.stack([x] * B)
    return {"Phi1": st(fx["Phi1"][:, :k]), "Phi2": st(fx["Phi2"][:, :k]), "lam1": st(fx["lam1"][:k]), "lam2": st(fx["lam2"][:k]),
            "a1": st(fx["a1"]), "a2": st(fx["a2"]), "F1": st(fx["F1"]), "F2": np.stack(F2s)}


def test_fused_fit_against_the_multi_launch_path_and_across_batch_sizes(fx_cfg1, oracle_cfg1_fits):
    """the same fit through dm_fmap_fit_fused and through dm_fmap_fit_steps (six launches per evaluation): both end within 1e-4 of
    the oracle's tight float64 minimiser and within each other's flat-direction noise; the fused result of a pair is bit-identical
    in a batch of 1 (one unit per workgroup), 3, and 140 (whole chunks per workgroup)"""
    from densematcher_amd.engine import default_engine
    from densematcher_amd.pyFM.functional import LBFGS_OPTIONS
    eng = default_engine()
    fx = fx_cfg1
    k = 15
    rng = np.random.default_rng(0)
    x0 = orc.get_x0(k, k, float(fx["Phi1"][0, 0]), float(fx["Phi2"][0, 0]), float(fx["a1"].astype(np.float64).sum()),
                    float(fx["a2"].astype(np.float64).sum()))
    tight = dict(LBFGS_OPTIONS)
    b3 = _fit_batch(fx, k, 3, rng)
    C3, r3 = eng.fit_general(b3, NOTEBOOK_W, np.stack([x0] * 3), lbfgs_options=tight)
    assert getattr(r3, "path", "") == "fused"
    Cm, rm = eng.fit_general(b3, NOTEBOOK_W, np.stack([x0] * 3), lbfgs_options=tight, fused=False)
    assert getattr(rm, "path", "") != "fused"
    print("fused: iterations", r3.nit, "evaluations", r3.nfev, "launches", r3.evaluations, "| multi-launch: iterations", rm.nit,
          "| max |C_fused - C_multi| =", np.abs(C3 - Cm).max(), " energies", r3.fun, rm.fun)
    assert np.all((r3.status == 1) | (r3.status == 2))
    assert np.abs(C3 - Cm).max() <= 2e-4
    assert np.all(np.abs(r3.fun - rm.fun) <= 1e-9 * np.abs(rm.fun))
    for b in range(3):
        assert np.array_equal(C3[b][:, 0], x0[:, 0])
    # the oracle's energy at both results: the fused minimiser is as good
    e1, e2 = fx["Phi1"][:, :k], fx["Phi2"][:, :k]
    A, Bq = orc.project(e1, fx["a1"], fx["F1"]), orc.project(e2, fx["a2"], fx["F2"])
    ev = orc.ev_sqdiff(fx["lam1"][:k], fx["lam2"][:k])
    Ef, Gf = orc.energy_grad_general(C3[0], A, Bq, ev, e1, e2, fx["a1"], NOTEBOOK_W)
    Em, _ = orc.energy_grad_general(Cm[0], A, Bq, ev, e1, e2, fx["a1"], NOTEBOOK_W)
    assert Ef <= Em + 1e-9 * abs(Em) and np.abs(Gf).max() <= 1e-4 * max(1.0, abs(Ef))
    # batch invariance across the two decompositions
    one = {n: v[:1] for n, v in b3.items()}
    C1, r1 = eng.fit_general(one, NOTEBOOK_W, x0[None], lbfgs_options=tight)
    assert np.array_equal(C1[0], C3[0]) and r1.nit[0] == r3.nit[0] and r1.nfev[0] == r3.nfev[0]
    big = _fit_batch(fx, k, 140, rng)
    Cb, rb = eng.fit_general(big, NOTEBOOK_W, np.stack([x0] * 140), lbfgs_options=tight)
    for b in (0, 1, 2, 137, 139):
        assert np.array_equal(Cb[b], C3[b % 3]), b
        assert rb.nit[b] == r3.nit[b % 3]
    # SciPy's stopping rule (the reference's): stops earlier, inside the reference's own noise floor of the tight result
    Cr, rr = eng.fit_general(one, NOTEBOOK_W, x0[None])
    print("reference stopping rule: iterations", rr.nit, "evaluations", rr.nfev, "| |C - C_tight| =", np.abs(Cr[0] - C3[0]).max())
    assert np.abs(Cr[0] - C3[0]).max() <= 2e-3 and rr.nfev[0] < r3.nfev[0]


def test_fused_fit_k30_against_oracle(fx_cfg1, oracle_cfg1_fits):
    """the fixture's own size (30 x 30: the 32-wide register tiles, the generic two-loop recursion of n = 900 unknowns)"""
    from densematcher_amd.engine import default_engine
    from densematcher_amd.pyFM.functional import LBFGS_OPTIONS
    eng = default_engine()
    fx = fx_cfg1
    k = int(fx["k"])
    x0 = orc.get_x0(k, k, float(fx["Phi1"][0, 0]), float(fx["Phi2"][0, 0]), float(fx["a1"].astype(np.float64).sum()),
                    float(fx["a2"].astype(np.float64).sum()))
    one = {n: v[:1] for n, v in _fit_batch(fx, k, 1, None).items()}
    C1, r1 = eng.fit_general(one, NOTEBOOK_W, x0[None], lbfgs_options=dict(LBFGS_OPTIONS))
    assert r1.path == "fused"
    print("k = 30 fused fit: iterations", r1.nit, "evaluations", r1.nfev, "|C - C_oracle| =", np.abs(C1[0] - oracle_cfg1_fits["C_nb"]).max())
    assert np.abs(C1[0] - oracle_cfg1_fits["C_nb"]).max() <= 1e-4


@pytest.mark.parametrize("N1,N2,k1,k2", [(300, 517, 15, 13), (1000, 777, 20, 30), (2048, 2048, 15, 15)])
def test_fused_fp32_element_loop_against_oracle(N1, N2, k1, k2):
    """r06: the element loop of the fused evaluation in the REFERENCE's precision (fp32: pyFM/functional.py:379-383 moves every tensor
    to float32; products on v_pk_fma_f32, v_log_f32 / v_rcp_f32) -- energy and gradient within 1e-6 (relative to |E|, max |G|) of the
    oracle's float64 values on meshes of the path's sizes.  (On a 65-vertex mesh with a 32 x 32 map the bound is not met -- 1e-5,
    tools/fit_f32_gate.py: an entry within fp32 rounding of the clamp's corner m = 0 flips the derivative's 23 w jump, and 8 k entries do
    not average one flip away; the reference's fp32 evaluation has the same corner.)"""
    from densematcher_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(N1 + 3 * N2 + k1)
    B = 2
    e1, e2, a1, C, A, Bm, lam1, lam2 = _random_problem(rng, B, N1, N2, k1, k2)
    C[1] *= 40.0
    for w in ({"w_ent": 0.3}, dict(NOTEBOOK_W), {"w_p2p": 0.5, "w_ent": 0.3, "w_range01": 1.5, "w_sumto1": 2.0, "w_descr": 1.0, "w_lap": 0.1}):
        E, G = eng.energy_grad_fused(C, A, Bm, lam1, lam2, w, e1, e2, a1, precision="f32")
        for b in range(B):
            ev = orc.ev_sqdiff(lam1[b], lam2[b])
            Eo, Go = orc.energy_grad_general(C[b], A[b].astype(np.float64), Bm[b].astype(np.float64), ev, e1[b], e2[b], a1[b], w)
            assert abs(float(E[b]) - Eo) <= 1e-6 * abs(Eo), (w, b, float(E[b]), Eo)
            assert np.abs(G[b].cpu().numpy() - Go).max() <= 1e-6 * np.abs(Go).max(), (w, b)


@pytest.mark.parametrize("N1,N2,k1,k2", [(300, 517, 15, 13), (1000, 777, 16, 16), (129, 65, 7, 9), (200, 4100, 12, 16)])
def test_fused_fp32_loop_matrix_and_vector_forms(N1, N2, k1, k2):
    """r06: maps up to 16 x 16 run the fp32 element loop's two 16-deep products on v_mfma_f32_16x16x4_f32 (dm_set_option fit_mfma = 1, the
    default); the packed-vector form (0) stays for wider maps.  Same fp32 arithmetic in another summation order: both forms within 2e-6
    (relative to |E|, max |G|) of the oracle's float64 values, on ragged sizes (rows and columns past the last 64 x 128 unit) too."""
    from densematcher_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(N1 + 3 * N2 + k1)
    e1, e2, a1, C, A, Bm, lam1, lam2 = _random_problem(rng, 2, N1, N2, k1, k2)
    C[1] *= 40.0
    try:
        for w in ({"w_ent": 0.3}, dict(NOTEBOOK_W), {"w_p2p": 0.5, "w_ent": 0.3, "w_range01": 1.5, "w_sumto1": 2.0, "w_descr": 1.0, "w_lap": 0.1}):
            for mf in (0, 1):
                eng.set_option("fit_mfma", mf)
                E, G = eng.energy_grad_fused(C, A, Bm, lam1, lam2, w, e1, e2, a1, precision="f32")
                for b in range(2):
                    ev = orc.ev_sqdiff(lam1[b], lam2[b])
                    Eo, Go = orc.energy_grad_general(C[b], A[b].astype(np.float64), Bm[b].astype(np.float64), ev, e1[b], e2[b], a1[b], w)
                    assert abs(float(E[b]) - Eo) <= 2e-6 * abs(Eo), (w, mf, b, float(E[b]), Eo)
                    assert np.abs(G[b].cpu().numpy() - Go).max() <= 2e-6 * np.abs(Go).max(), (w, mf, b)
    finally:
        eng.set_option("fit_mfma", 1)


def test_fused_fit_matrix_form_same_iterations_batch_invariant(fx_cfg1):
    """the notebook's fit (15 x 15, SciPy's stopping rule: fp32 loop) with the products on the matrix instruction: the iteration count of the vector
    form to within a step (the rule stops on an energy at its fp32 noise floor), the map within 1e-4 of it (7e-7 where the counts agree), a
    pair's bits independent of its batch (1, 3, 140)"""
    from densematcher_amd.engine import default_engine
    eng = default_engine()
    fx = fx_cfg1
    k = 15
    x0 = orc.get_x0(k, k, float(fx["Phi1"][0, 0]), float(fx["Phi2"][0, 0]), float(fx["a1"].astype(np.float64).sum()),
                    float(fx["a2"].astype(np.float64).sum()))
    b3 = _fit_batch(fx, k, 3, None)
    try:
        eng.set_option("fit_mfma", 0)
        Cv, rv = eng.fit_general(b3, NOTEBOOK_W, np.stack([x0] * 3))
        eng.set_option("fit_mfma", 1)
        Cm, rm = eng.fit_general(b3, NOTEBOOK_W, np.stack([x0] * 3))
        assert rm.element_loop == "f32" and rv.element_loop == "f32"
        assert np.abs(rm.nit.astype(int) - rv.nit.astype(int)).max() <= 2          # (SciPy's rule stops on an fp32-noisy energy: 39 / 40 / 37 against 39 / 39 / 37)
        print("fused fit, matrix against vector form: |C_m - C_v| =", np.abs(Cm - Cv).max(), "iterations", rm.nit, rv.nit)
        assert np.abs(Cm - Cv).max() < 1e-4                                         # (one more step of a fit that stops ~5e-4 short of the minimiser)
        C1, _ = eng.fit_general({n: v[:1] for n, v in b3.items()}, NOTEBOOK_W, x0[None])
        assert np.array_equal(C1[0], Cm[0])
        big = _fit_batch(fx, k, 140, None)
        Cb, _ = eng.fit_general(big, NOTEBOOK_W, np.stack([x0] * 140))
        assert all(np.array_equal(Cb[b], Cm[b % 3]) for b in (0, 1, 2, 137, 139))
    finally:
        eng.set_option("fit_mfma", 1)


def test_fused_fit_precision_follows_the_stopping_rule(fx_cfg1):
    """SciPy's stopping rule (the reference's call, the default) runs the element loop in fp32 like the reference: the same iterations as
    the float64 loop, the map within 1e-4 of it (measured 3e-6), bit-identical in batches of 1, 3 and 140; a tighter rule (ftol 1e-12 is
    below the fp32 noise floor of the energy) keeps float64"""
    from densematcher_amd.engine import default_engine
    from densematcher_amd.pyFM.functional import LBFGS_OPTIONS
    eng = default_engine()
    fx = fx_cfg1
    k = 15
    x0 = orc.get_x0(k, k, float(fx["Phi1"][0, 0]), float(fx["Phi2"][0, 0]), float(fx["a1"].astype(np.float64).sum()),
                    float(fx["a2"].astype(np.float64).sum()))
    b3 = _fit_batch(fx, k, 3, None)
    C3, r3 = eng.fit_general(b3, NOTEBOOK_W, np.stack([x0] * 3))
    assert r3.path == "fused" and r3.element_loop == "f32"
    C64, r64 = eng.fit_general(b3, NOTEBOOK_W, np.stack([x0] * 3), precision="f64")
    assert r64.element_loop == "f64"
    print("reference stopping: fp32 loop nit", r3.nit, "nfev", r3.nfev, "| f64 loop nit", r64.nit, "nfev", r64.nfev, "| max |C_f32 - C_f64| =", np.abs(C3 - C64).max())
    assert np.all((r3.status == 1) | (r3.status == 2))
    assert np.abs(C3 - C64).max() <= 1e-4
    assert np.all(np.abs(r3.nit.astype(int) - r64.nit.astype(int)) <= 5)
    C1, r1 = eng.fit_general({n: v[:1] for n, v in b3.items()}, NOTEBOOK_W, x0[None])
    assert np.array_equal(C1[0], C3[0]) and r1.nit[0] == r3.nit[0]
    big = _fit_batch(fx, k, 140, None)
    Cb, rb = eng.fit_general(big, NOTEBOOK_W, np.stack([x0] * 140))
    for b in (0, 1, 2, 137, 139):
        assert np.array_equal(Cb[b], C3[b % 3]), b
    Ct, rt = eng.fit_general({n: v[:1] for n, v in b3.items()}, NOTEBOOK_W, x0[None], lbfgs_options=dict(LBFGS_OPTIONS))
    assert rt.element_loop == "f64"

"""
CPU: pin the oracle (oracle/dm_oracle.py) against the golden vectors that
tools/make_golden.py produced by running the reference itself.
"""
import numpy as np
import pytest

from oracle import dm_oracle as orc


def _f64(fx, *names):
    return [np.asarray(fx[n], dtype=np.float64) for n in names]


@pytest.mark.parametrize("fxname", ["fx_cfg1", "fx_cfg2", "fx_cfg5"])
def test_fit_matches_reference(fxname, request):
    fx = request.getfixturevalue(fxname)
    phi1, phi2, a1, a2 = _f64(fx, "Phi1", "Phi2", "a1", "a2")
    k = int(fx["k"])
    phi1, phi2 = phi1[:, :k], phi2[:, :k]
    lam1, lam2 = fx["lam1"][:k], fx["lam2"][:k]
    C = orc.fit(phi1, phi2, lam1, lam2, a1, a2, fx["F1"], fx["F2"], float(fx["w_descr"]), float(fx["w_lap"]))
    # closed form == float64 L-BFGS-B over the reference's analytic gradients
    assert np.abs(C - fx["C_f64"]).max() < 2e-6
    # distance to the reference's fp32-autograd fit(): its own noise floor
    assert np.abs(C - fx["C_fit"]).max() < 2e-3
    # pinned first column (functional.py:654-658, base_functions.py:759)
    assert np.array_equal(C[:, 0], fx["C_fit"][:, 0])


def test_pieces_cfg1(fx_cfg1):
    fx = fx_cfg1
    k = int(fx["k"])
    phi1, phi2, a1, a2 = _f64(fx, "Phi1", "Phi2", "a1", "a2")
    phi1k, phi2k = phi1[:, :k], phi2[:, :k]
    A = orc.project(phi1k, a1, fx["F1"])
    B = orc.project(phi2k, a2, fx["F2"])
    assert np.abs(A - fx["A_f64"]).max() < 1e-12
    assert np.abs(B - fx["B_f64"]).max() < 1e-12
    assert np.abs(A - fx["A_f32"]).max() < 1e-5 * np.abs(A).max() + 1e-7      # reference computes these in fp32
    ev = orc.ev_sqdiff(fx["lam1"][:k], fx["lam2"][:k])
    assert np.array_equal(ev, fx["ev_sqdiff"])
    x0 = orc.get_x0(k, k, phi1[0, 0], phi2[0, 0], a1.sum(), a2.sum())
    assert np.allclose(x0, fx["x0"], rtol=1e-15, atol=0)
    # energy/gradient: the minimiser has zero free gradient
    g = orc.grad_energy(fx["C_f64"], A, B, ev, float(fx["w_descr"]), float(fx["w_lap"]))
    assert np.abs(g).max() < 1e-4 * float(fx["w_descr"]) * np.abs(A @ A.T).max()
    Cl, res = orc.fmap_fit_lbfgs(A, B, ev, x0, float(fx["w_descr"]), float(fx["w_lap"]))
    assert np.abs(Cl - fx["C_f64"]).max() < 1e-5


@pytest.mark.parametrize("fxname,pre", [("fx_cfg1", ""), ("fx_cfg2", ""), ("fx_cfg2", "f64_"), ("fx_cfg5", ""), ("fx_cfg5", "f64_")])
def test_maps_bit_exact(fxname, pre, request):
    fx = request.getfixturevalue(fxname)
    k = int(fx["k"])
    phi1, phi2, a1 = _f64(fx, "Phi1", "Phi2", "a1")
    C = fx["C_f64"] if pre else fx["C_fit"]
    p21, p12, ind = orc.fm_to_p2p(C, phi1[:, :k], phi2[:, :k], a1)
    i21, i12 = orc.indicator_argmax(ind, np.ones(ind.shape[0]))
    assert np.array_equal(p21, fx[pre + "knn21"])
    assert np.array_equal(p12, fx[pre + "knn12"])
    assert np.array_equal(i21, fx[pre + "ind21"])
    assert np.array_equal(i12, fx[pre + "ind12"])
    q = orc.fm_to_p2p_all(C, phi1[:, :k], phi2[:, :k], a1, chunk=300)
    for got, name in zip(q, ["knn21", "knn12", "ind21", "ind12"]):
        assert np.array_equal(got, fx[pre + name])
    if fxname == "fx_cfg1":
        assert np.allclose(ind[fx["ind_row_ids"]], fx["ind_rows"], rtol=1e-12, atol=1e-14)


def test_ties(fx_ties):
    fx = fx_ties
    phi1, phi2, a1 = _f64(fx, "Phi1", "Phi2", "a1")
    p21, p12, ind = orc.fm_to_p2p(fx["C"], phi1, phi2, a1)
    i21, i12 = orc.indicator_argmax(ind)
    # np.argmax is first-index: the indicator maps are pinned bit-exactly, ties included
    assert np.array_equal(i21, fx["ind21"])
    assert np.array_equal(i12, fx["ind12"])
    # the kd-tree breaks exact ties by traversal order, not by index: where the
    # reference differs from the lowest-index rule it must be an exact tie.
    e1 = phi1 @ fx["C"].T
    for got, ref, tree, query in [(p21, fx["knn21"], e1, phi2), (p12, fx["knn12"], phi2 @ fx["C"], phi1)]:
        diff = np.nonzero(got != ref)[0]
        for i in diff:
            assert np.array_equal(tree[got[i]], tree[ref[i]]), "mismatch that is not an exact duplicate"
            assert got[i] < ref[i]
        assert len(diff) <= 80


def test_p2p_to_fm(fx_cfg1):
    fx = fx_cfg1
    k = int(fx["k"])
    phi1, phi2, a2 = _f64(fx, "Phi1", "Phi2", "a2")
    C = orc.p2p_to_fm(fx["knn21"], phi1[:, :k], phi2[:, :k], a2)
    assert np.abs(C - fx["C_from_p2p"]).max() < 1e-13
    C = orc.p2p_to_fm(fx["knn21"], phi1[:, :k], phi2[:, :k])
    assert np.abs(C - fx["C_from_p2p_lstsq"]).max() < 1e-10


def test_zoomout(fx_cfg1, fx_cfg2):
    fx = fx_cfg1
    phi1, phi2, a2 = _f64(fx, "Phi1", "Phi2", "a2")
    C, p = orc.zoomout_refine(fx["C20"], phi1, phi2, nit=20, step=1, a2=a2, return_p2p=True)
    assert C.shape == (40, 40)
    assert np.array_equal(p, fx["p21_zo"])
    assert np.abs(C - fx["C_zo"]).max() < 1e-12
    C, p = orc.zoomout_refine(fx["C20"], phi1, phi2, nit=6, step=4, a2=a2, return_p2p=True)
    assert C.shape == (44, 44)
    assert np.array_equal(p, fx["p21_zo4"])
    assert np.abs(C - fx["C_zo4"]).max() < 1e-12
    fx = fx_cfg2
    phi1, phi2, a2 = _f64(fx, "Phi1", "Phi2", "a2")
    C, p = orc.zoomout_refine(fx["C_fit"], phi1, phi2, nit=3, step=4, a2=a2, return_p2p=True)
    assert np.array_equal(p, fx["p21_zo"])
    assert np.abs(C - fx["C_zo"]).max() < 1e-12
    with pytest.raises(AssertionError):
        orc.zoomout_refine(fx["C_fit"], phi1, phi2, nit=4, step=4, a2=a2)


def test_icp(fx_cfg1):
    fx = fx_cfg1
    k = int(fx["k"])
    phi1, phi2, a1 = _f64(fx, "Phi1", "Phi2", "a1")
    C = orc.icp_refine(fx["C_fit"], phi1[:, :k], phi2[:, :k], nit=10)
    assert np.abs(C - fx["C_icp"]).max() < 1e-9
    q = orc.fm_to_p2p_all(fx["C_icp"], phi1[:, :k], phi2[:, :k], a1)
    for got, name in zip(q, ["icp_knn21", "icp_knn12", "icp_ind21", "icp_ind12"]):
        assert np.array_equal(got, fx[name])


def test_compute_surface_map_tuple(fx_cfg1):
    """The 14-tuple's integer slots equal the piecewise reference outputs
    (functional_map.py:48-50,75-77,81)."""
    fx = fx_cfg1
    assert np.array_equal(fx["csm_p2p_21"], fx["ind21"])
    assert np.array_equal(fx["csm_p2p_12"], fx["ind12"])
    assert np.array_equal(fx["csm_p2p_21_adjoint"], fx["knn21"])
    assert np.array_equal(fx["csm_p2p_12_adjoint"], fx["knn12"])
    assert np.array_equal(fx["csm_p2p_21_icp"], fx["icp_ind21"])
    assert np.array_equal(fx["csm_p2p_21_icp_adjoint"], fx["icp_knn21"])
    assert np.abs(fx["csm_FM_base"] - fx["C_fit"]).max() == 0.0


def test_simnn_definition():
    rng = np.random.default_rng(0)
    S = rng.standard_normal((64, 32)).astype(np.float16)
    T = S[rng.permutation(64)][:40]
    nn = orc.simnn(T, S)
    assert np.array_equal(S[nn], T)
    S2 = np.concatenate([S, S[:5]])            # duplicates: lowest index wins
    assert np.array_equal(orc.simnn(T, S2), nn)


def test_general_energy_terms_against_reference(fx_cfg1, fx_cfg1_terms):
    """every further energy term of the restatement equals the reference's torch function (value and autograd gradient,
    evaluated in float64 by tools/make_golden_r02.py) at a fixed C: base_functions.py:124-226, 296-428"""
    fx, ft = fx_cfg1, fx_cfg1_terms
    k = int(fx["k"])
    e1, e2 = fx["Phi1"][:, :k].astype(np.float64), fx["Phi2"][:, :k].astype(np.float64)
    C = ft["C_test"]
    for name, w in (("p2p", "w_p2p"), ("stochastic", "w_stochastic"), ("ent", "w_ent"), ("range01", "w_range01"), ("sumto1", "w_sumto1")):
        E, G = orc.m_terms_energy_grad(C, e1, e2, fx["a1"], **{w: 1.0})
        assert abs(E - float(ft["E_" + name])) <= 1e-12 * abs(float(ft["E_" + name])), name
        assert np.abs(G - ft["G_" + name]).max() <= 1e-12 * np.abs(ft["G_" + name]).max(), name
    nd = int(ft["dcomm_ndescr"])
    o1 = orc.descr_ops(e1, fx["a1"], fx["F1"][:, :nd])
    o2 = orc.descr_ops(e2, fx["a2"], fx["F2"][:, :nd])
    assert np.abs(o1 - ft["ops1"]).max() < 1e-14 and np.abs(o2 - ft["ops2"]).max() < 1e-14
    E, G = orc.dcomm_energy_grad(C, o1, o2)
    assert abs(E - float(ft["E_dcomm"])) <= 1e-12 * float(ft["E_dcomm"])
    assert np.abs(G - ft["G_dcomm"]).max() <= 1e-12 * np.abs(ft["G_dcomm"]).max()


def test_fit_with_descriptor_commutativity_against_reference(fx_cfg1, fx_cfg1_terms):
    """the pyFM default w_dcomm = 1 (quadratic): the float64 L-BFGS-B of the restatement lands within the reference's fp32
    noise floor of its fit() output"""
    fx = fx_cfg1
    k = int(fx["k"])
    C, res = orc.fit_general(fx["Phi1"][:, :k], fx["Phi2"][:, :k], fx["lam1"][:k], fx["lam2"][:k], fx["a1"], fx["a2"], fx["F1"], fx["F2"],
                             dict(w_descr=1e4, w_lap=1e3, w_dcomm=1.0))
    assert np.abs(C - fx_cfg1_terms["C_fit_dcomm"]).max() < 2e-3
    assert np.array_equal(C[:, 0], fx_cfg1_terms["C_fit_dcomm"][:, 0])


def test_linear_sum_assignment_restatement_equals_scipy():
    """the step-for-step restatement of SciPy's rectangular LSAP (third-party arithmetic of functional_map.py:57,66,78)
    returns SciPy's assignment exactly: random real costs, integer costs with many ties, sparse 0/1-like matrices as
    the precise map produces, rectangular both ways, maximize"""
    import scipy.optimize
    rng = np.random.default_rng(0)
    for trial in range(40):
        n = int(rng.integers(2, 45))
        m = int(n + rng.integers(-6, 7))
        m = max(m, 1)
        c = rng.standard_normal((n, m))
        if trial % 4 == 1:
            c = np.round(2 * c)
        if trial % 4 == 2:
            c = c * (rng.random((n, m)) < 0.1)
        mx = bool(trial % 2)
        r0, c0 = scipy.optimize.linear_sum_assignment(c, maximize=mx)
        r1, c1 = orc.linear_sum_assignment(c, maximize=mx)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1), (trial, n, m)


def test_precise_map_against_reference(fx_cfg1, fx_cfg1_precise):
    """get_precise_map: face choice and barycentric weights of every vertex equal the reference's sparse matrix"""
    fx = fx_cfg1
    M, fm, bary = orc.precise_map_dense(fx["C_fit"], fx["Phi1"].astype(np.float64), fx["Phi2"].astype(np.float64), fx["faces1"])
    P = np.zeros_like(M)
    P[fx_cfg1_precise["precise_rows"], fx_cfg1_precise["precise_cols"]] = fx_cfg1_precise["precise_vals"]
    assert np.abs(M - P).max() < 1e-12
    assert np.allclose(bary.sum(axis=1), 1.0)


def test_shape_and_orientation_terms_against_reference(fx_cfg1, fx_cfg1_shape_terms):
    """area / conformal energies and gradients (base_functions.py:228-294) against the reference's values and torch autograd
    gradients; the orientation operators against BOTH forms the reference builds (compute_orientation_op, NumPy float64:
    functional.py:686-728; orientation_op_torch inside energy_func_std, float32 inside: base_functions.py:430-478) and the
    commutation energy / gradient on them (base_functions.py:176-203)."""
    fx, ft = fx_cfg1, fx_cfg1_shape_terms
    k = int(fx["k"])
    C = ft["C"]
    e, g = orc.area_energy_grad(C)
    assert abs(e - float(ft["area_E"])) <= 1e-12 * abs(e) and np.abs(g - ft["area_G"]).max() <= 1e-11
    e, g = orc.conformal_energy_grad(C, fx["lam1"][:k], fx["lam2"][:k])
    assert abs(e - float(ft["conf_E"])) <= 1e-12 * abs(e) and np.abs(g - ft["conf_G"]).max() <= 1e-11
    nd = int(ft["ndesc"])
    ops = []
    for which in (1, 2):
        phi, a = fx[f"Phi{which}"][:, :k].astype(np.float64), fx[f"a{which}"].astype(np.float64)
        F = fx[f"F{which}"][:, :nd].astype(np.float64)
        o_np = orc.orientation_ops(phi, a, fx[f"verts{which}"], fx[f"faces{which}"], F, vertex_areas=ft[f"vertex_areas{which}"])
        o_t = orc.orientation_ops(phi, a, fx[f"verts{which}"], fx[f"faces{which}"], F)
        sc = np.abs(ft[f"orient_np_op{which}"]).max()
        assert np.abs(o_np - ft[f"orient_np_op{which}"]).max() <= 1e-11 * sc
        assert np.abs(o_t - ft[f"orient_t_op{which}"]).max() <= 5e-6 * sc              # (the reference's gradients are float32 there)
        ops.append(ft[f"orient_t_op{which}"])
    e, g = orc.dcomm_energy_grad(C, ops[0], ops[1])
    assert abs(e - float(ft["orient_E"])) <= 1e-11 * abs(e) and np.abs(g - ft["orient_G"]).max() <= 1e-10 * np.abs(g).max()


def test_zoomout_config4_full_length_against_reference(fx_cfg4):
    """r05: the reference's own zoomout_refine 50 -> 200 (150 iterations, N = 2048) -- the oracle follows it to the end: same final
    vertex map, C to 1e-9"""
    fx = fx_cfg4
    phi1, phi2, a2 = _f64(fx, "Phi1", "Phi2", "a2")
    C, p21 = orc.zoomout_refine(fx["C0"], phi1, phi2, nit=int(fx["nit"]), step=1, a2=a2, return_p2p=True)
    assert np.array_equal(p21, fx["p21_zo"])
    assert np.abs(C - fx["C_zo"]).max() < 1e-9

"""
GPU (-m gpu): the reference's Python call surface (compute_surface_map, FunctionalMapping, spectral.*, refine.*)
served by the HIP library, against the golden fixtures / the oracle.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import dm_oracle as orc

pytestmark = pytest.mark.gpu


class _Duck:
    """what compute_surface_map needs from a pytorch3d Meshes (reference functional_map.py:17-18)"""
    def __init__(self, v, f):
        import torch
        self.v, self.f = torch.tensor(v), torch.tensor(f)

    def verts_list(self):
        return [self.v]

    def faces_list(self):
        return [self.f]


def _mesh(fx, which, k=None):
    from densematcher_amd.pyFM.mesh import TriMesh
    m = TriMesh(fx[f"verts{which}"], fx[f"faces{which}"])
    kk = fx[f"Phi{which}"].shape[1] if k is None else k
    m.A = sp.diags(fx[f"a{which}"].astype(np.float64)).tocsr()
    m.W = sp.identity(m.n_vertices).tocsr()            # not used by the matching path
    m.eigenvalues = fx[f"lam{which}"][:kk].copy()
    m.eigenvectors = fx[f"Phi{which}"][:, :kk].astype(np.float64)
    return m


def test_spectral_functions(fx_cfg1):
    from densematcher_amd.pyFM import spectral
    fx = fx_cfg1
    k = int(fx["k"])
    A1 = sp.diags(fx["a1"].astype(np.float64)).tocsr()
    A2 = sp.diags(fx["a2"].astype(np.float64)).tocsr()
    e1, e2 = fx["Phi1"].astype(np.float64), fx["Phi2"].astype(np.float64)
    p21, p12, ind = spectral.FM_to_p2p(fx["C_fit"], e1, e2, A1)
    assert p21.dtype == np.int64 and np.array_equal(p21, fx["knn21"]) and np.array_equal(p12, fx["knn12"])
    eta = np.ones(e2.shape[0])
    assert np.array_equal((ind * eta[..., None]).argmax(axis=1), fx["ind21"])      # functional_map.py:49
    assert np.array_equal((ind * eta[..., None]).argmax(axis=0), fx["ind12"])
    dense = np.asarray(ind)                                                          # materialised on the GPU
    assert dense.shape == (500, 500)
    assert np.allclose(dense[fx["ind_row_ids"]], fx["ind_rows"], rtol=1e-12, atol=1e-15)
    assert np.array_equal(dense.argmax(axis=1), fx["ind21"])
    with pytest.raises(AssertionError):
        spectral.FM_to_p2p(np.zeros((60, 60)), e1, e2, A1)
    C = spectral.p2p_to_FM(fx["knn21"], e1[:, :k], e2[:, :k], A2=A2)
    assert np.abs(C - fx["C_from_p2p"]).max() < 1e-13
    C = spectral.p2p_to_FM(fx["knn21"], e1[:, :k], e2[:, :k], A2=fx["a2"].astype(np.float64))
    assert np.abs(C - fx["C_from_p2p"]).max() < 1e-13
    with pytest.raises(ValueError):
        spectral.p2p_to_FM(fx["knn21"], e1[:, :k], e2[:100, :k], A2=A2)
    # knn_query: arbitrary point sets
    rng = np.random.default_rng(0)
    X, Y = rng.standard_normal((333, 7)), rng.standard_normal((129, 7))
    assert np.array_equal(spectral.knn_query(X, Y), orc.knn_query(X, Y))
    d, m = spectral.knn_query(X, Y, return_distance=True)
    assert np.allclose(d, np.linalg.norm(X[m] - Y, axis=1))


def test_refine_functions(fx_cfg1):
    from densematcher_amd.pyFM import refine
    fx = fx_cfg1
    k = int(fx["k"])
    A2 = sp.diags(fx["a2"].astype(np.float64)).tocsr()
    e1, e2 = fx["Phi1"].astype(np.float64), fx["Phi2"].astype(np.float64)
    C, p = refine.zoomout_refine(fx["C20"], e1, e2, nit=20, step=1, A2=A2, return_p2p=True)
    assert C.shape == (40, 40) and np.array_equal(p, fx["p21_zo"]) and np.abs(C - fx["C_zo"]).max() < 1e-11
    C1 = refine.zoomout_iteration(fx["C20"], e1, e2, step=1, A2=A2)
    assert C1.shape == (21, 21)
    with pytest.raises(AssertionError):
        refine.zoomout_refine(fx["C20"], e1, e2, nit=40, step=1, A2=A2)
    # ICP: normal equations + Newton-Schulz polar factor against the reference's lstsq + SVD
    C = refine.icp_refine(fx["C_fit"], e1[:, :k], e2[:, :k], None, nit=10)
    err = np.abs(C - fx["C_icp"]).max()
    print("ICP |C_gpu - C_ref| =", err, " orthogonality", np.abs(C.T @ C - np.eye(k)).max())
    assert err < 1e-8
    C1 = refine.icp_iteration(fx["C_fit"], e1[:, :k], e2[:, :k])
    C1o = orc.icp_refine(fx["C_fit"], e1[:, :k], e2[:, :k], nit=1)
    assert np.abs(C1 - C1o).max() < 1e-9


def test_refine_variants(fx_cfg1):
    """the forms of ICP / ZoomOut / p2p_to_FM beyond the fused GPU loops: tolerance-driven ICP (icp.py:84-96), least-squares
    and sparse-map p2p_to_FM (convert.py:39,51), rectangular maps with two step sizes and subsampled vertices in ZoomOut
    (zoomout.py:80-105)"""
    from densematcher_amd.pyFM import refine, spectral
    fx = fx_cfg1
    k = int(fx["k"])
    e1, e2 = fx["Phi1"].astype(np.float64), fx["Phi2"].astype(np.float64)
    A2 = sp.diags(fx["a2"].astype(np.float64)).tocsr()
    # least squares against the reference's own output, sparse map form against the index form
    Cl = spectral.p2p_to_FM(fx["knn21"], e1[:, :k], e2[:, :k])
    assert np.abs(Cl - fx["C_from_p2p_lstsq"]).max() < 1e-9
    n2, n1 = e2.shape[0], e1.shape[0]
    P = sp.csr_matrix((np.ones(n2), (np.arange(n2), fx["knn21"])), shape=(n2, n1))
    assert np.abs(spectral.p2p_to_FM(P, e1[:, :k], e2[:, :k], A2=A2) - fx["C_from_p2p"]).max() < 1e-12
    assert np.abs(spectral.p2p_to_FM(P, e1[:, :k], e2[:, :k]) - Cl).max() < 1e-12
    # tolerance-driven ICP: same fixed point as the oracle's loop with the same stopping rule
    C = refine.icp_refine(fx["C_fit"], e1[:, :k], e2[:, :k], None, nit=None, tol=1e-7)
    Co = np.array(fx["C_fit"])
    for _ in range(10000):
        Cn = orc.icp_refine(Co, e1[:, :k], e2[:, :k], nit=1)
        done = np.max(np.abs(Cn - Co)) <= 1e-7
        Co = Cn
        if done:
            break
    assert np.abs(C - Co).max() < 1e-8
    # rectangular map, two step sizes
    C0 = fx["C20"][:18, :20]
    Cz, pz = refine.zoomout_refine(C0, e1, e2, nit=5, step=(2, 3), A2=A2, return_p2p=True)
    Czo, pzo = orc.zoomout_refine(C0, e1, e2, nit=5, step=(2, 3), a2=fx["a2"], return_p2p=True)
    assert Cz.shape == (33, 30) and np.array_equal(pz, pzo) and np.abs(Cz - Czo).max() < 1e-11
    # subsampled vertices (least-squares p2p_to_FM inside, final map on all vertices)
    rng = np.random.default_rng(0)
    sub = (np.sort(rng.choice(n1, 300, replace=False)), np.sort(rng.choice(n2, 320, replace=False)))
    Cs, ps = refine.zoomout_refine(fx["C20"], e1, e2, nit=6, step=2, A2=A2, subsample=sub, return_p2p=True)
    Cso, pso = orc.zoomout_refine(fx["C20"], e1, e2, nit=6, step=2, a2=fx["a2"], subsample=sub, return_p2p=True)
    assert Cs.shape == (32, 32) and np.abs(Cs - Cso).max() < 1e-8 and np.array_equal(ps, pso)


def test_functional_mapping_and_surface_map(fx_cfg1, monkeypatch):
    """whole reference call surface on the fixture's (float32-rounded) spectrum"""
    from densematcher_amd.functional_map import compute_surface_map
    from densematcher_amd.pyFM import FunctionalMapping
    from densematcher_amd.pyFM.mesh import TriMesh
    fx = fx_cfg1
    k = int(fx["k"])
    fit_params = dict(w_descr=float(fx["w_descr"]), w_lap=float(fx["w_lap"]), w_dcomm=0, optinit="zeros", maxiter=5000)

    model = FunctionalMapping(_mesh(fx, 1, k), _mesh(fx, 2, k), partial=False, optimizer="L-BFGS-B")
    model.preprocess(n_ev=(k, k), n_descr=128, descr1=fx["F1"], descr2=fx["F2"], subsample_step=1)
    with pytest.raises(NotImplementedError):
        model.fit(w_descr=1e4, w_lap=1e3, w_dcomm=0, w_mumford_shah=1.0)    # Mumford-Shah / area-difference / eta-entropy are off the path
    model.fit(**fit_params)
    assert model.FM.shape == (k, k) and model.FM.dtype == np.float64
    assert np.abs(model.FM - fx["C_f64"]).max() < 1e-4
    assert np.abs(model.get_x0() - fx["x0"]).max() < 1e-15
    p21a, p12a = model.get_p2p()
    q = orc.fm_to_p2p_all(model.FM, fx["Phi1"][:, :k].astype(np.float64), fx["Phi2"][:, :k].astype(np.float64), fx["a1"])
    assert np.array_equal(p21a, q[0]) and np.array_equal(p12a, q[1])
    assert np.array_equal(model.mapped_indicator.argmax(axis=1), q[2])
    model.zoomout_refine(nit=2, step=1) if False else None     # (k == stored width: nothing to zoom into)
    model.icp_refine(nit=3)
    assert model.FM_type == "icp" and np.abs(model.FM.T @ model.FM - np.eye(k)).max() < 1e-9

    # compute_surface_map: fresh TriMesh objects are built inside; give them the fixture's spectrum
    by_verts = [(fx["verts1"], 1), (fx["verts2"], 2)]

    def process(self, k=200, **kw):
        for vv, which in by_verts:
            if np.array_equal(self.vertlist, vv):
                src = _mesh(fx, which, k)
                self.W, self.A, self.eigenvalues, self.eigenvectors = src.W, src.A, src.eigenvalues, src.eigenvectors
                return self
        raise RuntimeError("unknown mesh")

    monkeypatch.setattr(TriMesh, "process", process)
    res = compute_surface_map(_Duck(fx["verts1"], fx["faces1"]), _Duck(fx["verts2"], fx["faces2"]), fx["F1"], fx["F2"], n_ev=k,
                              optimizer="L-BFGS-B", fit_params=fit_params)
    assert len(res) == 14
    # slots 0,1 / 10,11: indicator arg-max and kd-tree maps of the plain map, equal to the oracle on the same C
    Cg = res[7]._FM_base
    assert np.abs(Cg - fx["C_f64"]).max() < 1e-4
    q = orc.fm_to_p2p_all(Cg, fx["Phi1"][:, :k].astype(np.float64), fx["Phi2"][:, :k].astype(np.float64), fx["a1"])
    for got, ref in zip([res[10], res[11], res[0], res[1]], q):
        assert np.array_equal(got, ref)
    # against the reference's own tuple (its C comes from fp32 L-BFGS: maps agree except for a few near-ties)
    agree = [(res[0] == fx["csm_p2p_21"]).mean(), (res[1] == fx["csm_p2p_12"]).mean(),
             (res[10] == fx["csm_p2p_21_adjoint"]).mean(), (res[11] == fx["csm_p2p_12_adjoint"]).mean(),
             (res[4] == fx["csm_p2p_21_icp"]).mean(), (res[12] == fx["csm_p2p_21_icp_adjoint"]).mean()]
    print("compute_surface_map agreement with the reference tuple:", [round(float(a), 4) for a in agree])
    assert min(agree[:4]) >= 0.98
    # slots 4,5 / 12,13: the ICP maps, bit-exact against the oracle's ICP + maps started from the SAME plain map
    # (functional_map.py:71-77; the agreement with the reference tuple above is lower only because ICP amplifies the
    # 5e-4 distance between the closed-form C and the reference's fp32 L-BFGS C_fit)
    e1, e2 = fx["Phi1"][:, :k].astype(np.float64), fx["Phi2"][:, :k].astype(np.float64)
    C_icp_o = orc.icp_refine(Cg, e1, e2, nit=10)
    assert np.abs(res[7].FM - C_icp_o).max() < 1e-8
    qi = orc.fm_to_p2p_all(res[7].FM, e1, e2, fx["a1"])
    for got, ref in zip([res[12], res[13], res[4], res[5]], qi):
        assert np.array_equal(got, ref)
    assert res[6] is not None and len(res[6]) == 2          # hungarian_icp: the GPU assignment kernel (SciPy's algorithm)
    assert np.array_equal(res[6][0], np.arange(500))
    assert res[2] is None and res[3] is None                # compute_extra=False


def test_fit_on_spectral_signatures():
    """descr_type='HKS' (row a-3 of the scope table): descriptors from the host mirror, fit on the GPU; float64
    descriptors go through the float64 projection path.  Against the oracle on the same descriptors, and (loosely)
    against the map the reference's own fp32 L-BFGS-B produced (tests/golden/fx_sig.npz)."""
    import os
    import types
    from densematcher_amd.pyFM import FunctionalMapping
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "fx_sig.npz"))
    k = int(fx["k"])

    def mesh(which):
        m = types.SimpleNamespace(eigenvalues=fx[f"lam{which}"][:k].copy(), eigenvectors=fx[f"Phi{which}"][:, :k].astype(np.float64),
                                  A=sp.diags(fx[f"a{which}"].astype(np.float64)).tocsr())
        m.process = lambda *a, **kw: m
        m.area = float(fx[f"a{which}"].astype(np.float64).sum())
        return m

    model = FunctionalMapping(mesh(1), mesh(2), partial=False, optimizer="L-BFGS-B")
    model.preprocess(n_ev=(k, k), n_descr=16, descr_type="HKS", landmarks=fx["landmarks2"], subsample_step=2)
    assert np.abs(model.descr1 - fx["pre_descr1"]).max() <= 1e-12 * np.abs(fx["pre_descr1"]).max()
    model.fit(w_descr=1e4, w_lap=1e3, w_dcomm=0, optinit="zeros")
    Co = orc.fit(fx["Phi1"][:, :k], fx["Phi2"][:, :k], fx["lam1"][:k], fx["lam2"][:k], fx["a1"], fx["a2"],
                 model.descr1.astype(np.float32), model.descr2.astype(np.float32), 1e4, 1e3)
    assert np.abs(model.FM - Co).max() <= 1e-4
    # The HKS problem is ill-conditioned (32 nearly collinear descriptors): the reference's fp32 L-BFGS-B stops in a flat
    # valley (|grad| = 2e-2, 0.5 away in C) at an energy 6.5e-5 (relative) ABOVE the minimum.  Parity is therefore
    # stated on the energy both minimise: the GPU map is at least as good and within 1e-3 of the reference's value.
    A = orc.project(fx["Phi1"][:, :k], fx["a1"], model.descr1.astype(np.float32))
    B = orc.project(fx["Phi2"][:, :k], fx["a2"], model.descr2.astype(np.float32))
    ev = orc.ev_sqdiff(fx["lam1"][:k], fx["lam2"][:k])
    e_gpu, e_ref = orc.energy(model.FM, A, B, ev, 1e4, 1e3), orc.energy(fx["C_fit_hks"], A, B, ev, 1e4, 1e3)
    print("energy: GPU", e_gpu, " reference fit", e_ref, " |C_gpu - C_fit| =", np.abs(model.FM - fx["C_fit_hks"]).max())
    assert e_gpu <= e_ref * (1 + 1e-12) and (e_ref - e_gpu) <= 1e-3 * e_ref
    assert np.array_equal(model.FM[:, 0], fx["C_fit_hks"][:, 0])                 # the pinned column (get_x0) is identical


# --------------------------------------------------------------------------- #
# the energy terms beyond w_descr / w_lap
MIX = dict(w_descr=1e4, w_lap=1e3, w_dcomm=0.5, w_p2p=0.05, w_stochastic=0.02, w_ent=0.1, w_range01=1.0, w_sumto1=2.0)
NOTEBOOK = dict(w_descr=1e4, w_lap=1e3, w_dcomm=0, w_ent=1e-1, w_sumto1=1e1, optinit="zeros", maxiter=5000)   # example.ipynb cell 11


def _terms_setup(fx, nd=None):
    from densematcher_amd.engine import default_engine
    eng = default_engine()
    k = int(fx["k"])
    F1, F2 = (fx["F1"], fx["F2"]) if nd is None else (fx["F1"][:, :nd].copy(), fx["F2"][:, :nd].copy())
    e1, e2 = fx["Phi1"][:, :k].copy(), fx["Phi2"][:, :k].copy()
    A = eng.project(e1[None], fx["a1"][None], F1[None], exact=True)
    B = eng.project(e2[None], fx["a2"][None], F2[None], exact=True)
    return eng, k, e1, e2, F1, F2, A, B


def test_energy_terms_against_oracle(fx_cfg1, fx_cfg1_terms):
    """dm_fmap_energy_grad: every term alone and all together, value and gradient, against the oracle (itself pinned to the
    reference's torch functions at this very C, tests/test_oracle_golden.py) -- base_functions.py:480-763"""
    fx, ft = fx_cfg1, fx_cfg1_terms
    nd = 12
    eng, k, e1, e2, F1, F2, A, B = _terms_setup(fx, nd)
    C = ft["C_test"]
    A64, B64 = A[0].cpu().numpy().astype(np.float64), B[0].cpu().numpy().astype(np.float64)
    ev = orc.ev_sqdiff(fx["lam1"][:k], fx["lam2"][:k])
    o1, o2 = orc.descr_ops(e1, fx["a1"], F1), orc.descr_ops(e2, fx["a2"], F2)
    ops1 = eng.descr_ops(e1[None], fx["a1"][None], F1[None])
    ops2 = eng.descr_ops(e2[None], fx["a2"][None], F2[None])
    assert np.abs(ops1[0].cpu().numpy() - o1).max() <= 1e-13 * np.abs(o1).max()
    assert np.abs(ops2[0].cpu().numpy() - o2).max() <= 1e-13 * np.abs(o2).max()
    cases = [{n: v} for n, v in MIX.items()] + [MIX, {n: v for n, v in NOTEBOOK.items() if n.startswith("w_")}]
    for w in cases:
        E, G = eng.energy_grad(C[None], A, B, fx["lam1"][None, :k], fx["lam2"][None, :k], w, e1[None], e2[None], fx["a1"][None], ops1, ops2)
        Eo, Go = orc.energy_grad_general(C, A64, B64, ev, e1, e2, fx["a1"], w, o1, o2)
        E, G = float(E[0]), G[0].cpu().numpy()
        assert abs(E - Eo) <= 1e-11 * abs(Eo), (w, E, Eo)
        assert np.abs(G - Go).max() <= 1e-11 * max(np.abs(Go).max(), 1e-300), (w, np.abs(G - Go).max(), np.abs(Go).max())
        assert np.all(G[:, 0] == 0)
    # a batch evaluates every pair independently
    E2, G2 = eng.energy_grad(np.stack([C, 0.5 * C]), A.repeat(2, 1, 1), B.repeat(2, 1, 1), np.stack([fx["lam1"][:k]] * 2),
                             np.stack([fx["lam2"][:k]] * 2), MIX, np.stack([e1] * 2), np.stack([e2] * 2), np.stack([fx["a1"]] * 2),
                             ops1.repeat(2, 1, 1, 1), ops2.repeat(2, 1, 1, 1))
    E1, G1 = eng.energy_grad(C[None], A, B, fx["lam1"][None, :k], fx["lam2"][None, :k], MIX, e1[None], e2[None], fx["a1"][None], ops1, ops2)
    assert float(E2[0]) == float(E1[0]) and np.array_equal(G2[0].cpu().numpy(), G1[0].cpu().numpy())


def test_fit_with_notebook_params(fx_cfg1, fx_cfg1_terms, oracle_cfg1_fits):
    """FunctionalMapping.fit with the reference notebook's fit_params (example.ipynb cell 11: w_ent = 0.1, w_sumto1 = 10):
    within 1e-4 of the float64 minimiser of the reference energy, within the reference's fp32 noise floor of its own fit()
    output, stationary and not worse than it on the oracle's energy"""
    from densematcher_amd.pyFM import FunctionalMapping
    fx = fx_cfg1
    k = int(fx["k"])
    model = FunctionalMapping(_mesh(fx, 1, k), _mesh(fx, 2, k), partial=False, optimizer="L-BFGS-B")
    model.preprocess(n_ev=(k, k), n_descr=128, descr1=fx["F1"], descr2=fx["F2"], subsample_step=1)
    model.fit(**NOTEBOOK, stopping="tight")
    C = model.FM
    print("notebook fit:", model.fit_result.nit, "iterations,", model.fit_result.nfev, "evaluations;",
          "|C - C_oracle| =", np.abs(C - oracle_cfg1_fits["C_nb"]).max(), " |C - C_fit(reference)| =", np.abs(C - fx_cfg1_terms["C_fit_nb"]).max())
    assert np.abs(C - oracle_cfg1_fits["C_nb"]).max() <= 1e-4
    assert np.abs(C - fx_cfg1_terms["C_fit_nb"]).max() <= 2e-3
    assert np.array_equal(C[:, 0], model.get_x0()[:, 0])
    e1, e2 = fx["Phi1"][:, :k], fx["Phi2"][:, :k]
    A, B = orc.project(e1, fx["a1"], fx["F1"]), orc.project(e2, fx["a2"], fx["F2"])
    ev = orc.ev_sqdiff(fx["lam1"][:k], fx["lam2"][:k])
    w = {n: v for n, v in NOTEBOOK.items() if n.startswith("w_")}
    Eg, Gg = orc.energy_grad_general(C, A, B, ev, e1, e2, fx["a1"], w)
    Er, _ = orc.energy_grad_general(fx_cfg1_terms["C_fit_nb"], A, B, ev, e1, e2, fx["a1"], w)
    # (the GPU objective uses the fp32-rounded projections of dm_project: the float64 oracle gradient at its minimiser is ~1e-5 relative)
    assert Eg <= Er and np.abs(Gg).max() <= 1e-4 * max(1.0, abs(Eg))


def test_fit_verbose_prints_the_reference_terms(fx_cfg1, monkeypatch, capsys):
    """VERBOSE in the environment (reference base_functions.py:27-29, 538-636): every live term's weighted loss under the
    reference's labels, at the start point and at the solution; the printed terms add up to the objective"""
    from densematcher_amd.pyFM import FunctionalMapping
    from densematcher_amd.engine import default_engine
    fx = fx_cfg1
    k = int(fx["k"])
    monkeypatch.setenv("VERBOSE", "1")
    model = FunctionalMapping(_mesh(fx, 1, k), _mesh(fx, 2, k), partial=False, optimizer="L-BFGS-B")
    model.preprocess(n_ev=(k, k), n_descr=128, descr1=fx["F1"], descr2=fx["F2"], subsample_step=1)
    model.fit(**NOTEBOOK)
    out = capsys.readouterr().out
    live = {n: v for n, v in NOTEBOOK.items() if n.startswith("w_") and v > 0}
    labels = dict(model._VERBOSE_LABELS)
    sol = out.split("energy terms at the solution:")[1]
    vals = {}
    for n in live:
        assert out.count(labels[n]) == 2, (n, out)
        vals[n] = float(sol.split(labels[n])[1].split()[0])
    total = float(default_engine().fit_energy(model._dev, live, model.FM[None])[0])
    print("VERBOSE terms at the solution:", vals, "sum", sum(vals.values()), "objective", total)
    assert abs(sum(vals.values()) - total) <= 1e-9 * abs(total)
    monkeypatch.delenv("VERBOSE")
    model.fit(**NOTEBOOK)
    assert "loss:" not in capsys.readouterr().out


def test_fit_with_descriptor_commutativity(fx_cfg1, fx_cfg1_terms):
    """the pyFM default w_dcomm = 1 (all 128 descriptor operators): the bare model.fit(w_descr, w_lap) call of the reference"""
    from densematcher_amd.pyFM import FunctionalMapping
    fx = fx_cfg1
    k = int(fx["k"])
    model = FunctionalMapping(_mesh(fx, 1, k), _mesh(fx, 2, k), partial=False, optimizer="L-BFGS-B")
    model.preprocess(n_ev=(k, k), n_descr=128, descr1=fx["F1"], descr2=fx["F2"], subsample_step=1)
    model.fit(w_descr=1e4, w_lap=1e3, stopping="tight")         # w_dcomm defaults to 1
    Co, _ = orc.fit_general(fx["Phi1"][:, :k], fx["Phi2"][:, :k], fx["lam1"][:k], fx["lam2"][:k], fx["a1"], fx["a2"], fx["F1"], fx["F2"],
                            dict(w_descr=1e4, w_lap=1e3, w_dcomm=1.0))
    print("w_dcomm fit: |C - C_oracle| =", np.abs(model.FM - Co).max(), " |C - C_fit(reference)| =", np.abs(model.FM - fx_cfg1_terms["C_fit_dcomm"]).max())
    assert np.abs(model.FM - Co).max() <= 1e-4
    assert np.abs(model.FM - fx_cfg1_terms["C_fit_dcomm"]).max() <= 2e-3


def test_fit_with_orientation_area_conformal_terms(fx_cfg1, fx_cfg1_shape_terms):
    """The last part of fit()'s surface (VERDICT r03 missing #1): w_area, w_conformal (dm_fmap_energy_grad weights 8, 9) and w_orient
    (orientation operators on the host like the reference, the commutation energy on the GPU through the operator lists).
    Energy and gradient of each term against the oracle (pinned to the reference's autograd in fx_cfg1_shape_terms.npz); the
    orientation operators of the mirror against the reference's; the rescaled orientation weight and the fitted map against the
    reference's own fit (its float32 L-BFGS-B noise floor)."""
    from densematcher_amd.engine import default_engine
    from densematcher_amd.pyFM.functional import FunctionalMapping
    fx, ft = fx_cfg1, fx_cfg1_shape_terms
    k, nd = int(fx["k"]), int(ft["ndesc"])
    eng = default_engine()
    model = FunctionalMapping(_mesh(fx, 1, k), _mesh(fx, 2, k), partial=False, optimizer="L-BFGS-B")
    model.preprocess(n_ev=(k, k), n_descr=nd, descr1=fx["F1"][:, :nd].astype(np.float64), descr2=fx["F2"][:, :nd].astype(np.float64), subsample_step=1)
    ops = model.compute_orientation_op()
    o1, o2 = np.stack([a for a, _ in ops]), np.stack([b for _, b in ops])
    assert np.abs(o1 - ft["orient_np_op1"]).max() <= 1e-10 * np.abs(o1).max() and np.abs(o2 - ft["orient_np_op2"]).max() <= 1e-10 * np.abs(o2).max()
    opsr = model.compute_orientation_op(reversing=True)
    assert np.array_equal(opsr[0][0], ops[0][0]) and np.array_equal(opsr[0][1], -ops[0][1])
    # energy + gradient of the three terms on the GPU at the fixture's map
    e1, e2 = fx["Phi1"][:, :k].astype(np.float32), fx["Phi2"][:, :k].astype(np.float32)
    batch = {"Phi1": e1[None], "Phi2": e2[None], "a1": fx["a1"][None], "a2": fx["a2"][None], "lam1": fx["lam1"][None, :k], "lam2": fx["lam2"][None, :k],
             "F1": fx["F1"][None, :, :nd].astype(np.float32), "F2": fx["F2"][None, :, :nd].astype(np.float32)}
    C = ft["C"]
    A, Bm, lam1, lam2, w, P1, P2, a1, ops1, ops2, _, _ = eng._fit_inputs(batch, dict(w_orient=0.7, w_area=2.0, w_conformal=3.0), None,
                                                                        (ft["orient_t_op1"][None], ft["orient_t_op2"][None]))
    w.setdefault("w_descr", 0.0)
    e, g = eng.energy_grad(C[None], A, Bm, lam1, lam2, w, P1, P2, a1, ops1, ops2)
    ea, ga = orc.area_energy_grad(C)
    ec, gc = orc.conformal_energy_grad(C, fx["lam1"][:k], fx["lam2"][:k])
    eo, go = orc.dcomm_energy_grad(C, ft["orient_t_op1"], ft["orient_t_op2"])
    Eo, Go = 2.0 * ea + 3.0 * ec + 0.7 * eo, 2.0 * ga + 3.0 * gc + 0.7 * go
    Go[:, 0] = 0
    assert abs(float(e[0]) - Eo) <= 1e-11 * abs(Eo)
    assert np.abs(g[0].cpu().numpy() - Go).max() <= 1e-11 * np.abs(Go).max()
    # the reference's fit with the three terms switched on
    model.fit(w_descr=1e4, w_lap=1e3, w_dcomm=0, w_orient=1, w_area=1e2, w_conformal=1e2, optinit="zeros", stopping="tight")
    d = np.abs(model.FM - ft["fit_orient_C"]).max()
    print("fit with w_orient / w_area / w_conformal: |C - C_reference| = %.2e, rescaled w_orient = %.4e" % (d, model.w_orient_rescaled))
    assert d <= 5e-3


def test_compute_surface_map_notebook_call(fx_cfg1, fx_cfg1_notebook_call, monkeypatch):
    """The reference's one documented call (example.ipynb cell 11) runs unchanged: notebook fit_params (w_ent, w_sumto1),
    compute_extra=True (Hungarian on the plain map, precise map + Hungarian, ICP, Hungarian on the ICP map).  Every output is
    pinned to the oracle on the SAME functional maps; against the reference's own tuple the agreement is reported
    (its C comes from a float32 L-BFGS-B that stops 7e-4 away from the minimiser)."""
    import scipy.optimize
    from densematcher_amd.functional_map import compute_surface_map
    from densematcher_amd.pyFM.mesh import TriMesh
    fx, ref = fx_cfg1, fx_cfg1_notebook_call
    k = int(fx["k"])
    by_verts = [(fx["verts1"], 1), (fx["verts2"], 2)]

    def process(self, k=200, **kw):
        for vv, which in by_verts:
            if np.array_equal(self.vertlist, vv):
                src = _mesh(fx, which, k)
                self.W, self.A, self.eigenvalues, self.eigenvectors = src.W, src.A, src.eigenvalues, src.eigenvectors
                return self
        raise RuntimeError("unknown mesh")

    monkeypatch.setattr(TriMesh, "process", process)
    res = compute_surface_map(_Duck(fx["verts1"], fx["faces1"]), _Duck(fx["verts2"], fx["faces2"]), fx["F1"], fx["F2"], n_ev=k,
                              compute_extra=True, optimizer="L-BFGS-B", fit_params=dict(NOTEBOOK))
    assert len(res) == 14 and all(r is not None for r in res)
    model = res[7]
    e1, e2 = fx["Phi1"][:, :k].astype(np.float64), fx["Phi2"][:, :k].astype(np.float64)
    C0, Ci = model._FM_base, model.FM
    assert np.abs(C0 - ref["FM_base"]).max() <= 2e-3                       # the reference's fp32 noise floor
    # plain map: maps, Hungarian, precise map + Hungarian -- oracle on the same C
    q = orc.fm_to_p2p_all(C0, e1, e2, fx["a1"])
    for got, want in zip([res[10], res[11], res[0], res[1]], q):
        assert np.array_equal(got, want)
    M0 = orc.mapped_indicator(C0, e1, e2, fx["a1"])
    h0 = scipy.optimize.linear_sum_assignment(M0, maximize=True)
    assert np.array_equal(res[2][0], h0[0])
    assert abs(M0[res[2]].sum() - M0[h0].sum()) <= 1e-9 * abs(M0[h0].sum())    # (same objective; the GPU indicator differs in the last bits)
    # the matrix the GPU assignment ran on: every entry within the summation-order bound of the oracle's (2 k + 2 roundings on the
    # sum of absolute terms), the assignment IS SciPy's on that matrix, and it is counted against SciPy's on the oracle's matrix
    from densematcher_amd.engine import default_engine
    Mg = default_engine().mapped_indicator(fx["Phi1"][None, :, :k], fx["Phi2"][None, :, :k], fx["a1"][None], C0[None])[0].cpu().numpy()
    S = ((np.abs(e2) @ np.abs(C0)) @ np.abs(e1).T) * fx["a1"].astype(np.float64)[None, :]
    ulp_bound = (2 * k + 2) * np.finfo(np.float64).eps * S
    assert (np.abs(Mg - M0) <= ulp_bound).all(), float((np.abs(Mg - M0) / S).max() / np.finfo(np.float64).eps)
    hg = scipy.optimize.linear_sum_assignment(Mg, maximize=True)
    assert np.array_equal(res[2][1], hg[1])                                    # identical to SciPy on the GPU's own matrix
    same_as_oracle_matrix = float((res[2][1] == h0[1]).mean())
    print("hungarian on the GPU indicator vs on the oracle's indicator: %.4f of the rows equal" % same_as_oracle_matrix)
    assert same_as_oracle_matrix >= 0.99
    P0, _, _ = orc.precise_map_dense(C0, e1, e2, fx["faces1"])
    hp = scipy.optimize.linear_sum_assignment(P0, maximize=True)
    assert abs(P0[res[3]].sum() - P0[hp].sum()) <= 1e-9 * abs(P0[hp].sum())
    # ICP map
    assert np.abs(Ci - orc.icp_refine(C0, e1, e2, nit=10)).max() < 1e-8
    qi = orc.fm_to_p2p_all(Ci, e1, e2, fx["a1"])
    for got, want in zip([res[12], res[13], res[4], res[5]], qi):
        assert np.array_equal(got, want)
    Mi = orc.mapped_indicator(Ci, e1, e2, fx["a1"])
    hi = scipy.optimize.linear_sum_assignment(Mi, maximize=True)
    assert abs(Mi[res[6]].sum() - Mi[hi].sum()) <= 1e-9 * abs(Mi[hi].sum())
    agree = {n: round(float((np.asarray(a) == ref[n]).mean()), 4) for n, a in
             dict(p2p_21=res[0], p2p_12=res[1], hungarian_cols=res[2][1], hungarian_precise_cols=res[3][1], p2p_21_icp=res[4],
                  p2p_12_icp=res[5], hungarian_icp_cols=res[6][1], p2p_21_adjoint=res[10], p2p_12_adjoint=res[11]).items()}
    print("notebook call, agreement with the reference's tuple:", agree)
    assert min(agree[n] for n in ("p2p_21", "p2p_12", "p2p_21_adjoint", "p2p_12_adjoint", "hungarian_cols")) >= 0.95
    # the ICP slots ABSOLUTELY (a relative statement alone lets 0.97 -> 0.80 pass): measured 0.97 / 0.98 / 0.996
    assert min(agree[n] for n in ("p2p_21_icp", "p2p_12_icp", "hungarian_icp_cols")) >= 0.95, agree
    assert agree["hungarian_precise_cols"] >= 0.95, agree
    # the same call with the package's tight stopping rule (ftol 1e-12: the float64 minimiser; the reference's own fit stops ~5e-4 short
    # of it, so the tuple moves AWAY from the reference's -- the reason the default is SciPy's rule)
    res_t = compute_surface_map(_Duck(fx["verts1"], fx["faces1"]), _Duck(fx["verts2"], fx["faces2"]), fx["F1"], fx["F2"], n_ev=k,
                                compute_extra=True, optimizer="L-BFGS-B", fit_params=dict(NOTEBOOK, stopping="tight"))
    agree_t = {n: round(float((np.asarray(a) == ref[n]).mean()), 4) for n, a in
               dict(p2p_21=res_t[0], p2p_12=res_t[1], hungarian_cols=res_t[2][1], hungarian_precise_cols=res_t[3][1], p2p_21_icp=res_t[4],
                    p2p_12_icp=res_t[5], hungarian_icp_cols=res_t[6][1], p2p_21_adjoint=res_t[10], p2p_12_adjoint=res_t[11]).items()}
    print("notebook call with stopping='tight', agreement with the reference's tuple:", agree_t,
          "max |C - C_ref| = %.2e (default rule: %.2e)" % (np.abs(res_t[7]._FM_base - ref["FM_base"]).max(), np.abs(C0 - ref["FM_base"]).max()))
    assert min(agree_t[n] for n in ("p2p_21", "p2p_12", "p2p_21_adjoint", "p2p_12_adjoint")) >= 0.90
    # the default must not be the worse of the two on the slots the fit's stopping point moves most (the ICP maps)
    assert agree["p2p_21_icp"] + agree["p2p_12_icp"] >= agree_t["p2p_21_icp"] + agree_t["p2p_12_icp"] - 0.02


def test_compute_surface_map_batch_equals_single_calls(fx_cfg1, monkeypatch):
    """compute_surface_map_batch (no reference counterpart: the documented call with a batch dimension) returns, pair by pair, what
    compute_surface_map returns: three pairs -- two of one size (batched kernels: device L-BFGS over both maps, 2 x 4 vertex maps,
    precise maps, ICP, six assignments in one launch), one with mesh 2 subsampled (its own group) -- on injected spectra, so
    that both paths see the same eigenbases bit for bit."""
    from densematcher_amd.functional_map import compute_surface_map, compute_surface_map_batch
    from densematcher_amd.pyFM.mesh import TriMesh
    fx = fx_cfg1
    k = int(fx["k"])
    by_verts = [(fx["verts1"], 1), (fx["verts2"], 2)]

    def process(self, k=200, **kw):
        for vv, which in by_verts:
            if np.array_equal(self.vertlist, vv):
                src = _mesh(fx, which, k)
                self.W, self.A, self.eigenvalues, self.eigenvectors = src.W, src.A, src.eigenvalues, src.eigenvectors
                return self
        raise RuntimeError("unknown mesh")

    monkeypatch.setattr(TriMesh, "process", process)
    rng = np.random.default_rng(0)
    F1b = (fx["F1"].astype(np.float32) + 0.05 * rng.standard_normal(fx["F1"].shape)).astype(np.float16)
    pairs = [(_Duck(fx["verts1"], fx["faces1"]), _Duck(fx["verts2"], fx["faces2"]), fx["F1"], fx["F2"]),
             (_Duck(fx["verts1"], fx["faces1"]), _Duck(fx["verts2"], fx["faces2"]), F1b, fx["F2"]),
             (_Duck(fx["verts2"], fx["faces2"]), _Duck(fx["verts1"], fx["faces1"]), fx["F2"], fx["F1"])]
    kw = dict(n_ev=k, compute_extra=True, optimizer="L-BFGS-B", fit_params=dict(NOTEBOOK))
    got = compute_surface_map_batch([p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs], [p[3] for p in pairs], **kw)
    assert len(got) == 3
    for q, p in enumerate(pairs):
        want = compute_surface_map(*p, **kw)
        assert np.array_equal(got[q][7]._FM_base, want[7]._FM_base), q        # a pair's fit does not depend on its batch
        assert np.array_equal(got[q][7].FM, want[7].FM), q                    # (the ICP map)
        for slot in (0, 1, 4, 5, 10, 11, 12, 13):
            assert np.array_equal(got[q][slot], want[slot]), (q, slot)
        for slot in (2, 3, 6):
            assert np.array_equal(got[q][slot][0], want[slot][0]) and np.array_equal(got[q][slot][1], want[slot][1]), (q, slot)
    # r05: the same results whichever way the work is laid over streams -- two chunk streams (helper threads, upload streams), and the
    # single call with its three assignments in one launch instead of on side streams as soon as their matrices exist
    import densematcher_amd.functional_map as fmod
    got2 = compute_surface_map_batch([p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs], [p[3] for p in pairs], streams=2, **kw)
    monkeypatch.setattr(fmod, "EARLY_ASSIGNMENTS", False)
    want0 = compute_surface_map(*pairs[0], **kw)
    got0 = compute_surface_map_batch([pairs[0][0]], [pairs[0][1]], [pairs[0][2]], [pairs[0][3]], **kw)[0]
    for q in range(3):
        for other, mine in ((got2[q], got[q]),) + (((want0, got[0]), (got0, got[0])) if q == 0 else ()):
            for slot in (0, 1, 4, 5, 10, 11, 12, 13):
                assert np.array_equal(other[slot], mine[slot]), (q, slot)
            for slot in (2, 3, 6):
                assert np.array_equal(other[slot][0], mine[slot][0]) and np.array_equal(other[slot][1], mine[slot][1]), (q, slot)


def test_compute_surface_map_from_raw_meshes():
    """no injected spectrum: TriMesh.process assembles the (robust) Laplacian on the host and computes the eigenbasis on
    the GPU (dm_eigenbasis); the vertex maps equal those of the oracle pipeline run on SciPy's dense eigenbasis of the same
    W, A (functional maps are basis dependent -- signs, rotations inside clusters -- vertex maps are not)"""
    import warnings
    import scipy.linalg
    from densematcher_amd import synth
    from densematcher_amd.functional_map import compute_surface_map
    nu, nv, D = 32, 24, 96
    (v1, f1), (v2, f2) = synth.torus_mesh(nu, nv, perturb=0.05, seed=3), synth.torus_mesh(nu, nv, perturb=0.08, seed=1)
    from densematcher_amd.pyFM.mesh import laplacian as lap
    (W1, M1), (W2, M2) = lap.robust_mesh_laplacian(v1, f1), lap.robust_mesh_laplacian(v2, f2)     # what process(robust=True) assembles
    m1, m2 = M1.diagonal().astype(np.float32).astype(np.float64), M2.diagonal().astype(np.float32).astype(np.float64)
    w1, V1 = scipy.linalg.eigh(W1.toarray(), np.diag(m1))
    w2, V2 = scipy.linalg.eigh(W2.toarray(), np.diag(m2))
    k = max(range(34, 46), key=lambda q: min(w1[q] - w1[q - 1], w2[q] - w2[q - 1]))
    F1, F2, _ = synth.feature_pair(nu * nv, nu * nv, D, 5, 6, sigma=0.3, perm="identity")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        res = compute_surface_map(_Duck(v1, f1), _Duck(v2, f2), F1, F2, n_ev=k, optimizer="L-BFGS-B",
                                  fit_params=dict(w_descr=1e4, w_lap=1e3, w_dcomm=0, optinit="zeros"))
    assert any("robust_laplacian" in str(w_.message) for w_ in caught)       # the wheel is absent: own implementation, said loudly
    model = res[7]
    assert np.abs(model.mesh1.eigenvalues - w1[:k]).max() <= 1e-8 * w1[k - 1]
    # oracle on the host basis (the fp32 rounding of the basis at the ABI included)
    e1, e2 = V1[:, :k].astype(np.float32).astype(np.float64), V2[:, :k].astype(np.float32).astype(np.float64)
    Co = orc.fit(e1, e2, w1[:k], w2[:k], m1, m2, F1, F2, 1e4, 1e3)
    q = orc.fm_to_p2p_all(Co, e1, e2, m1)
    agree = [float((got == want).mean()) for got, want in zip([res[10], res[11], res[0], res[1]], q)]
    print("raw-mesh compute_surface_map vs oracle on SciPy's eigenbasis:", agree)
    assert min(agree) >= 0.99


def test_knn_query_top_k():
    """nn_utils.knn_query with k > 1 (sklearn kneighbors semantics: (n2, k), nearest first)"""
    from densematcher_amd.pyFM.spectral.nn_utils import knn_query
    rng = np.random.default_rng(5)
    for nx, ny, p, k in ((700, 333, 3, 5), (1500, 64, 40, 8), (97, 211, 7, 97)):
        X, Y = rng.standard_normal((nx, p)), rng.standard_normal((ny, p))
        d_ref, i_ref = orc.knn_query_topk(X, Y, k)
        d, i = knn_query(X, Y, k=k, return_distance=True)
        assert i.shape == (ny, k) and i.dtype == np.int64 and d.shape == (ny, k)
        assert np.abs(d - d_ref).max() < 1e-12 * max(1.0, d_ref.max())
        gap_ok = np.ones((ny, k), dtype=bool)            # positions whose rank is decided by more than rounding
        dn = np.sqrt(((Y[:, None, :] - X[None]) ** 2).sum(-1))
        dn.sort(axis=1)
        gaps = np.diff(dn, axis=1)
        for r in range(k):
            lo = gaps[:, r - 1] if r > 0 else np.inf
            hi = gaps[:, r] if r < nx - 1 else np.inf
            gap_ok[:, r] = np.minimum(lo, hi) > 1e-10
        assert np.array_equal(i[gap_ok], i_ref[gap_ok]) and gap_ok.mean() > 0.99
        assert np.array_equal(knn_query(X, Y, k=k), i)
        assert all(len(set(row)) == k for row in i)
    # exact duplicates: the lowest index comes first, like a stable sort on the distances
    X = np.repeat(rng.standard_normal((50, 3)), 3, axis=0)
    Y = rng.standard_normal((40, 3))
    _, i_ref = orc.knn_query_topk(X, Y, 6)
    assert np.array_equal(knn_query(X, Y, k=6), i_ref)
    # k = 1 keeps the reference's squeezed shapes
    d1, i1 = knn_query(X, Y, k=1, return_distance=True)
    assert i1.shape == (40,) and d1.shape == (40,) and np.array_equal(i1, i_ref[:, 0])


def test_function_space_helpers(fx_cfg1):
    """TriMesh.project / decode / l2_* / integrate and FunctionalMapping.project / decode / transport / transfer
    (reference pyFM/mesh/trimesh.py:533-640, pyFM/functional.py:730-831) against their closed forms in float64."""
    from densematcher_amd.pyFM.functional import FunctionalMapping
    fx = fx_cfg1
    k = int(fx["k"])
    m1, m2 = _mesh(fx, 1), _mesh(fx, 2)
    rng = np.random.default_rng(11)
    f = rng.standard_normal((m1.n_vertices, 4))
    a1 = m1.A.diagonal()
    want = m1.eigenvectors.T @ (a1[:, None] * f)
    scale = np.abs(want).max()
    assert np.abs(m1.project(f) - want).max() < 2e-6 * scale                      # fp32 inputs, exact accumulation
    assert np.abs(m1.project(f[:, 0], k=20) - want[:20, 0]).max() < 2e-6 * scale
    assert m1.project(f[:, 0]).shape == (m1.eigenvectors.shape[1],)
    with pytest.raises(ValueError):
        m1.project(f, k=m1.eigenvectors.shape[1] + 1)
    c = rng.standard_normal((k, 3))
    assert np.array_equal(m1.decode(c), m1.eigenvectors[:, :k] @ c)
    with pytest.raises(ValueError):
        m1.decode(np.zeros((m1.eigenvectors.shape[1] + 1, 2)))
    assert np.allclose(m1.l2_inner(f, 2 * f), 2 * (f * a1[:, None] * f).sum(0), rtol=1e-13)
    assert np.allclose(m1.l2_sqnorm(f[:, 1]), (f[:, 1] ** 2 * a1).sum(), rtol=1e-13)
    assert np.allclose(m1.integrate(f), a1 @ f, rtol=1e-13)
    assert np.isclose(m1.integrate(np.ones(m1.n_vertices)), a1.sum(), rtol=1e-13)

    model = FunctionalMapping(m1, m2)
    model.k1 = model.k2 = k
    with pytest.raises(ValueError):
        model.transport(c)
    model.descr1 = model.descr2 = np.zeros((1, 1))                                # "preprocessed"
    model.FM = fx["C_fit"].astype(np.float64)
    assert np.abs(model.project(f) - want[:k]).max() < 2e-6 * scale
    a2 = m2.A.diagonal()
    g = rng.standard_normal((m2.n_vertices, 2))
    want2 = m2.eigenvectors[:, :k].T @ (a2[:, None] * g)
    assert np.abs(model.project(g, mesh_ind=2) - want2).max() < 2e-6 * np.abs(want2).max()
    with pytest.raises(ValueError):
        model.project(f, mesh_ind=3)
    assert np.array_equal(model.decode(c), m2.eigenvectors[:, :k] @ c)
    assert np.array_equal(model.decode(c, mesh_ind=1), m1.eigenvectors[:, :k] @ c)
    assert np.allclose(model.transport(c), model.FM @ c, rtol=1e-13)
    assert np.allclose(model.transport(c, reverse=True), np.linalg.pinv(model.FM) @ c, rtol=1e-12)
    t = model.transfer(f)
    assert t.shape == (m2.n_vertices, 4)
    assert np.abs(t - m2.eigenvectors[:, :k] @ (model.FM @ want[:k])).max() < 1e-5 * np.abs(t).max()
    tr = model.transfer(g, reverse=True)
    assert tr.shape == (m1.n_vertices, 2)
    assert np.abs(tr - m1.eigenvectors[:, :k] @ (np.linalg.pinv(model.FM) @ want2)).max() < 1e-5 * np.abs(tr).max()


@pytest.mark.parametrize("N1,N2,k1,k2", [(300, 517, 20, 33), (1000, 777, 70, 50), (640, 576, 200, 190), (129, 65, 15, 15)])
def test_energy_indicator_terms_ragged(N1, N2, k1, k2):
    """the tile-fused indicator terms (the N2 x N1 matrix is never stored) on sizes that are multiples of no tile, rectangular
    maps, k1 from 15 to 200 (one to four 16-column tiles of Y per wave): value and gradient against the oracle"""
    from densematcher_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(N1 + 3 * N2 + k1)
    B, D = 2, 8
    e1 = (rng.standard_normal((B, N1, k1)) / np.sqrt(N1)).astype(np.float32)
    e2 = (rng.standard_normal((B, N2, k2)) / np.sqrt(N2)).astype(np.float32)
    a1 = (rng.uniform(0.5, 1.5, (B, N1)) / N1).astype(np.float32)
    C = rng.standard_normal((B, k2, k1)) * 3.0
    A = rng.standard_normal((B, k1, D)).astype(np.float32)
    Bm = rng.standard_normal((B, k2, D)).astype(np.float32)
    lam1, lam2 = np.sort(rng.uniform(0, 50, (B, k1)), axis=1), np.sort(rng.uniform(0, 50, (B, k2)), axis=1)
    for w in ({"w_ent": 0.3}, {"w_sumto1": 2.0}, {"w_stochastic": 0.7}, {"w_p2p": 0.5, "w_range01": 1.5},
              {"w_p2p": 0.5, "w_stochastic": 0.7, "w_ent": 0.3, "w_range01": 1.5, "w_sumto1": 2.0, "w_descr": 1.0, "w_lap": 0.1}):
        E, G = eng.energy_grad(C, A, Bm, lam1, lam2, w, e1, e2, a1)
        for b in range(B):
            ev = orc.ev_sqdiff(lam1[b], lam2[b])
            Eo, Go = orc.energy_grad_general(C[b], A[b].astype(np.float64), Bm[b].astype(np.float64), ev, e1[b], e2[b], a1[b], w)
            assert abs(float(E[b]) - Eo) <= 1e-11 * abs(Eo), (w, b, float(E[b]), Eo)
            assert np.abs(G[b].cpu().numpy() - Go).max() <= 1e-11 * np.abs(Go).max(), (w, b)


def test_energy_indicator_terms_never_materialise_the_indicator():
    """N = 8192, four pairs: the old form needed B N2 N1 doubles (2.1 GB here, 34 GB for a batch of 64); the tile-fused one
    runs in O(N k) -- and pair 0 equals the oracle (which does build the 537 MB matrix, once)"""
    import torch
    from densematcher_amd.engine import MatchEngine
    eng = MatchEngine()                                   # a fresh context: its arena shows what this call alone needs
    rng = np.random.default_rng(8192)
    B, N, k, D = 4, 8192, 50, 8
    e1 = (rng.standard_normal((B, N, k)) / np.sqrt(N)).astype(np.float32)
    e2 = (rng.standard_normal((B, N, k)) / np.sqrt(N)).astype(np.float32)
    a1 = (rng.uniform(0.5, 1.5, (B, N)) / N).astype(np.float32)
    C = rng.standard_normal((B, k, k))
    A = rng.standard_normal((B, k, D)).astype(np.float32)
    Bm = rng.standard_normal((B, k, D)).astype(np.float32)
    lam = np.sort(rng.uniform(0, 50, (B, k)), axis=1)
    w = {"w_ent": 0.1, "w_sumto1": 10.0, "w_descr": 1e4, "w_lap": 1e3}
    E, G = eng.energy_grad(C, A, Bm, lam, lam, w, e1, e2, a1)
    torch.cuda.synchronize()
    assert eng.workspace_bytes() < (1 << 30) // 8, eng.workspace_bytes()        # 128 MiB for four pairs
    Eo, Go = orc.energy_grad_general(C[0], A[0].astype(np.float64), Bm[0].astype(np.float64), orc.ev_sqdiff(lam[0], lam[0]), e1[0], e2[0],
                                     a1[0], w)
    assert abs(float(E[0]) - Eo) <= 1e-11 * abs(Eo)
    assert np.abs(G[0].cpu().numpy() - Go).max() <= 1e-10 * np.abs(Go).max()
    eng.close()


def test_fit_general_batched_device_lbfgs(fx_cfg1, oracle_cfg1_fits):
    """The iterative fit on the device-resident batched L-BFGS (dm_lbfgs_*): every pair of a batch has its own optimiser
    state, so a pair's C is bit-identical whatever batch it is in; the result is the minimiser the
    oracle's tight float64 L-BFGS-B finds (1e-4), also with SciPy driving the same GPU evaluations, and the reference's
    default stopping rule lands inside the reference's own noise floor of it."""
    import torch
    from densematcher_amd.engine import default_engine
    eng = default_engine()
    fx = fx_cfg1
    k = int(fx["k"])
    w = {n: v for n, v in NOTEBOOK.items() if n.startswith("w_")}
    x0 = orc.get_x0(k, k, float(fx["Phi1"][0, 0]), float(fx["Phi2"][0, 0]), float(fx["a1"].astype(np.float64).sum()),
                    float(fx["a2"].astype(np.float64).sum()))
    rng = np.random.default_rng(3)

    def pair(scale, perm_seed):
        F2 = fx["F2"].copy()
        if perm_seed:
            F2 = F2[:, np.random.default_rng(perm_seed).permutation(F2.shape[1])]
        return {"Phi1": fx["Phi1"][:, :k], "Phi2": fx["Phi2"][:, :k], "lam1": fx["lam1"][:k], "lam2": fx["lam2"][:k], "a1": fx["a1"], "a2": fx["a2"],
                "F1": (fx["F1"].astype(np.float32) * scale).astype(np.float16), "F2": (F2.astype(np.float32) * scale).astype(np.float16)}
    pairs = [pair(1.0, 0), pair(0.5, 11), pair(1.0, 12)]        # pair 0 = the fixture; the others converge after different iteration counts
    batch = {n: np.stack([p[n] for p in pairs]) for n in pairs[0]}
    # the package's own tight rule (pyFM/functional.py: a relative decrease of a few machine epsilons, 1e-15, is decided by rounding
    # noise -- one of these pairs once needed more than 15000 evaluations to see it)
    from densematcher_amd.pyFM.functional import LBFGS_OPTIONS
    tight = dict(LBFGS_OPTIONS)
    C3, r3 = eng.fit_general(batch, w, np.stack([x0] * 3), lbfgs_options=tight)
    print("batched device L-BFGS: iterations", r3.nit, "evaluations", r3.nfev, "status", r3.status, "host loop", r3.evaluations)
    assert np.all((r3.status == 1) | (r3.status == 2)) and len(set(r3.nit.tolist())) > 1
    for b in range(3):
        one = {n: v[b:b + 1] for n, v in batch.items()}
        C1, r1 = eng.fit_general(one, w, x0[None], lbfgs_options=tight)
        assert np.array_equal(C1[0], C3[b]), f"pair {b}: result depends on the batch"
        assert r1.nit[0] == r3.nit[b] and r1.nfev[0] == r3.nfev[b]
    assert np.abs(C3[0] - oracle_cfg1_fits["C_nb"]).max() <= 1e-4
    assert np.array_equal(C3[0][:, 0], x0[:, 0])                  # pinned column
    # SciPy driving the same GPU evaluations (the reference's own optimiser), one pair
    one = {n: v[0:1] for n, v in batch.items()}
    Cs, rs = eng.fit_general(one, w, x0[None], lbfgs_options=tight, driver="scipy")
    print("scipy driver:", rs.nit, "iterations,", rs.nfev, "evaluations; |C_device - C_scipy| =", np.abs(Cs[0] - C3[0]).max())
    assert np.abs(Cs[0] - C3[0]).max() <= 1e-4
    # SciPy's default stopping rule (what the reference's call runs with): inside the reference's own noise floor
    Cr, rr = eng.fit_general(one, w, x0[None])
    print("reference stopping rule:", rr.nit, "iterations,", rr.nfev, "evaluations, status", rr.status, "; |C - C_tight| =", np.abs(Cr[0] - C3[0]).max())
    assert np.abs(Cr[0] - C3[0]).max() <= 2e-3 and rr.nfev[0] < r3.nfev[0]
    with pytest.raises(ValueError):
        eng.fit_general(batch, w, np.stack([x0] * 3), driver="scipy")
    # the evaluation loop behind the ABI (dm_fmap_fit_steps): one evaluation per call, four, or seven -- the same iterates, the same
    # counters (rounds past a pair's stop do not move it)
    for ce in (1, 7):
        Cc, rc_ = eng.fit_general(batch, w, np.stack([x0] * 3), lbfgs_options=tight, check_every=ce)
        assert np.array_equal(Cc, C3) and np.array_equal(rc_.nit, r3.nit) and np.array_equal(rc_.nfev, r3.nfev), ce


def test_zoomout_with_farthest_point_subsample_from_the_model(fx_cfg1):
    """FunctionalMapping.zoomout_refine(subsample=int) (functional.py:588-617): farthest point sampling of both meshes
    (TriMesh.extract_fps), the iterations on the samples with the least-squares p2p_to_FM, equal to the oracle's ZoomOut run
    on the same samples; Euclidean sampling follows the reference's arithmetic (geometry.py:813-845)"""
    import warnings
    from densematcher_amd.pyFM import FunctionalMapping, refine
    fx = fx_cfg1
    m1, m2 = _mesh(fx, 1, 48), _mesh(fx, 2, 48)
    s = m1.extract_fps(60, geodesic=False, rng=np.random.default_rng(5))
    assert len(set(s.tolist())) == 60
    V = m1.vertlist                                          # the reference's loop, restated
    d = np.linalg.norm(V - V[s[0]], axis=1)
    for q in range(1, 60):
        assert s[q] == int(np.argmax(d))
        d = np.minimum(d, np.linalg.norm(V - V[s[q]], axis=1))
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        g = m1.extract_fps(40, rng=np.random.default_rng(6))       # geodesic=True: edge-graph shortest paths, said loudly
    assert len(set(g.tolist())) == 40 and any("potpourri3d" in str(w.message) for w in caught)
    sub = (m1.extract_fps(300, geodesic=False, rng=np.random.default_rng(1)), m2.extract_fps(320, geodesic=False, rng=np.random.default_rng(2)))
    C0 = fx["C20"]
    Cs, ps = refine.mesh_zoomout_refine(C0, m1, m2, nit=6, step=2, subsample=sub, return_p2p=True)
    Cso, pso = orc.zoomout_refine(C0, fx["Phi1"].astype(np.float64), fx["Phi2"].astype(np.float64), nit=6, step=2, a2=fx["a2"],
                                  subsample=sub, return_p2p=True)
    assert np.abs(Cs - Cso).max() < 1e-8 and np.array_equal(ps, pso)
    model = FunctionalMapping(_mesh(fx, 1, 48), _mesh(fx, 2, 48), partial=False, optimizer="L-BFGS-B")
    model.preprocess(n_ev=(20, 20), n_descr=128, descr1=fx["F1"], descr2=fx["F2"], subsample_step=1, k_process=40)   # ZoomOut needs 30 columns
    model.FM = C0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.zoomout_refine(nit=5, step=2, subsample=200)
    assert model.FM_type == "zoomout" and model.FM.shape == (30, 30) and np.isfinite(model.FM).all()


def test_raw_non_delaunay_mesh_robust_laplacian():
    """A strongly perturbed torus (obtuse triangles: the cotangent Laplacian has negative weights): process(robust=True) -- what
    the reference always asks for -- assembles the tufted intrinsic-Delaunay Laplacian, the GPU eigensolver reproduces the
    dense generalised eigensolve of THAT pencil, and the printed numbers say what falling back to the cotangent Laplacian
    (round 2's behaviour without the wheel) would have cost in vertex-map agreement."""
    import warnings
    import scipy.linalg
    from densematcher_amd import synth
    from densematcher_amd.pyFM.mesh import TriMesh, laplacian as lap
    from densematcher_amd.pyFM import FunctionalMapping
    nu, nv, D, k = 32, 24, 96, 30
    (v1, f1), (v2, f2) = synth.torus_mesh(nu, nv, perturb=0.25, seed=7), synth.torus_mesh(nu, nv, perturb=0.22, seed=8)
    Wc, _ = synth.cotan_laplacian(v1, f1)
    assert (Wc.toarray() - np.diag(Wc.diagonal())).max() > 1e-6          # negative cotangent weights: not Delaunay
    Wr, Mr = lap.robust_mesh_laplacian(v1, f1)
    assert lap.robust_mesh_laplacian.last_info["flips"] > 0
    w_ref = scipy.linalg.eigh(Wr.toarray(), np.diag(Mr.diagonal().astype(np.float32).astype(np.float64)), eigvals_only=True,
                              subset_by_index=[0, k - 1])
    F1, F2, _ = synth.feature_pair(nu * nv, nu * nv, D, 5, 6, sigma=0.3, perm="identity")
    maps = {}
    for robust in (True, False):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m1, m2 = TriMesh(v1, f1).process(k, robust=robust), TriMesh(v2, f2).process(k, robust=robust)
        if robust:
            assert np.abs(m1.eigenvalues - w_ref).max() <= 1e-8 * w_ref[-1]
        model = FunctionalMapping(m1, m2, partial=False, optimizer="L-BFGS-B")
        model.preprocess(n_ev=(k, k), n_descr=D, descr1=F1, descr2=F2, subsample_step=1)
        model.fit(w_descr=1e4, w_lap=1e3, w_dcomm=0, optinit="zeros")
        p21, p12 = model.get_p2p()
        maps[robust] = (p21, p12, model.mesh1.eigenvalues.copy())
    print("non-Delaunay torus: robust vs cotangent Laplacian: eigenvalue change",
          float(np.abs(maps[True][2] - maps[False][2]).max() / maps[True][2][-1]),
          " p2p_21 agreement", float((maps[True][0] == maps[False][0]).mean()), " p2p_12 agreement", float((maps[True][1] == maps[False][1]).mean()))


def test_energy_keep_gram_option():
    """dm_set_option "energy_keep_gram": the Gram blocks of the projected descriptors are computed by the first evaluation and
    reused while the same A, B are passed (the L-BFGS driver's case): identical energies and gradients with and without;
    other operands, or setting the option again, recompute"""
    import torch
    from densematcher_amd.engine import default_engine
    eng = default_engine()
    rng = np.random.default_rng(5)
    B, k1, k2, D = 2, 15, 17, 40
    dev = eng.device
    A = torch.as_tensor(rng.standard_normal((B, k1, D)).astype(np.float32)).to(dev)
    Bm = torch.as_tensor(rng.standard_normal((B, k2, D)).astype(np.float32)).to(dev)
    A2 = torch.as_tensor(rng.standard_normal((B, k1, D)).astype(np.float32)).to(dev)
    lam1, lam2 = np.sort(rng.uniform(0, 50, (B, k1)), axis=1), np.sort(rng.uniform(0, 50, (B, k2)), axis=1)
    w = {"w_descr": 1.0, "w_lap": 0.1}
    Cs = [rng.standard_normal((B, k2, k1)) for _ in range(3)]
    plain = [eng.energy_grad(C, A, Bm, lam1, lam2, w) for C in Cs]
    plain2 = eng.energy_grad(Cs[0], A2, Bm, lam1, lam2, w)
    eng.set_option("energy_keep_gram", 1)
    try:
        kept = [eng.energy_grad(C, A, Bm, lam1, lam2, w) for C in Cs]
        kept2 = eng.energy_grad(Cs[0], A2, Bm, lam1, lam2, w)            # other operand: recomputed
        kept3 = eng.energy_grad(Cs[1], A, Bm, lam1, lam2, w)             # and back
    finally:
        eng.set_option("energy_keep_gram", 0)
    for (e0, g0), (e1, g1) in zip(plain + [plain2, plain[1]], kept + [kept2, kept3]):
        assert torch.equal(e0, e1) and torch.equal(g0, g1)
    assert not torch.equal(plain2[0], plain[0][0])
    # with the indicator statistics (w_sumto1: the bases' column sums ride along), other bases recompute
    N1, N2 = 300, 260
    e1 = torch.as_tensor((rng.standard_normal((B, N1, k1)) / np.sqrt(N1)).astype(np.float32)).to(dev)
    e2 = torch.as_tensor((rng.standard_normal((B, N2, k2)) / np.sqrt(N2)).astype(np.float32)).to(dev)
    e2b = torch.as_tensor((rng.standard_normal((B, N2, k2)) / np.sqrt(N2)).astype(np.float32)).to(dev)
    a1 = torch.as_tensor((rng.uniform(0.5, 1.5, (B, N1)) / N1).astype(np.float32)).to(dev)
    wm = {"w_descr": 1.0, "w_lap": 0.1, "w_sumto1": 2.0, "w_ent": 0.3}
    calls = [(Cs[0], e2), (Cs[1], e2), (Cs[2], e2b), (Cs[0], e2)]
    plain = [eng.energy_grad(C, A, Bm, lam1, lam2, wm, e1, p2, a1) for C, p2 in calls]
    eng.set_option("energy_keep_gram", 1)
    try:
        kept = [eng.energy_grad(C, A, Bm, lam1, lam2, wm, e1, p2, a1) for C, p2 in calls]
    finally:
        eng.set_option("energy_keep_gram", 0)
    for (e0, g0), (e1_, g1) in zip(plain, kept):
        assert torch.equal(e0, e1_) and torch.equal(g0, g1)
    for b in range(B):                                   # and the analytic sums against the oracle's dense indicator
        ev = orc.ev_sqdiff(lam1[b], lam2[b])
        Eo, Go = orc.energy_grad_general(Cs[0][b], A[b].cpu().numpy().astype(np.float64), Bm[b].cpu().numpy().astype(np.float64), ev,
                                         e1[b].cpu().numpy(), e2[b].cpu().numpy(), a1[b].cpu().numpy(), wm)
        assert abs(float(plain[0][0][b]) - Eo) <= 1e-11 * abs(Eo)
        assert np.abs(plain[0][1][b].cpu().numpy() - Go).max() <= 1e-11 * np.abs(Go).max()


def test_assignment_from_the_indicator_factors(fx_cfg1, monkeypatch):
    """dm_lsa_indicator: the linear assignments of mapped indicators evaluated from their factors (no N2 x N1 matrix) are the assignments
    of the dense matrices dm_mapped_indicator forms (same arithmetic, bit for bit), for 15 x 15 and 30 x 30 maps (the 16- and 32-wide
    register tiles), float64 and fp32 bases, with a dense matrix riding in the same launch; and _assign_many takes that route when
    the dense copies would pass its limit"""
    import scipy.optimize
    from densematcher_amd import functional_map as fmod
    from densematcher_amd.engine import default_engine
    from densematcher_amd.pyFM.spectral.convert import MappedIndicator
    eng = default_engine()
    fx = fx_cfg1
    rng = np.random.default_rng(4)
    for k, dt in ((15, np.float64), (30, np.float64), (15, np.float32)):
        P1, P2 = fx["Phi1"][None, :, :k].astype(dt), fx["Phi2"][None, :, :k].astype(dt)
        a1 = fx["a1"][None].astype(dt)
        Cs = np.stack([fx["C_f64"][:k, :k], fx["C_f64"][:k, :k] + 0.05 * rng.standard_normal((k, k))])
        rep = lambda x: np.concatenate([x, x])
        assert eng.lsa_indicator_ok(P1.shape[1], P2.shape[1], k, k)
        dense = eng.mapped_indicator(rep(P1), rep(P2), rep(a1), Cs)
        extra = torch_rand = rng.standard_normal((1, P2.shape[1], P1.shape[1]))
        got = eng.lsa_indicator(rep(P1), rep(P2), rep(a1), Cs, dense=extra).cpu().numpy()
        want = eng.linear_sum_assignment(dense, maximize=True).cpu().numpy()
        assert np.array_equal(got[:2], want), (k, dt)
        assert np.array_equal(got[2], scipy.optimize.linear_sum_assignment(extra[0], maximize=True)[1])
        assert np.array_equal(got[0], scipy.optimize.linear_sum_assignment(dense[0].cpu().numpy(), maximize=True)[1])
    # the route _assign_many takes beyond its stacking limit
    k = 15
    P1, P2, a1 = (eng._dev(x, __import__("torch").float64, "x") for x in (fx["Phi1"][None, :, :k], fx["Phi2"][None, :, :k], fx["a1"][None]))
    C = eng._dev(fx["C_f64"][None, :k, :k], __import__("torch").float64, "C")
    mi = MappedIndicator(eng, P1, P2, a1, C, None, None)
    ref = fmod._assign_many([MappedIndicator(eng, P1, P2, a1, C, None, None)])[0]
    monkeypatch.setattr(fmod, "_ASSIGN_STACK_LIMIT", 1)
    viaf = fmod._assign_many([mi])[0]
    assert getattr(mi, "_dev", None) is None                       # (no dense matrix was formed)
    assert np.array_equal(viaf[0], ref[0]) and np.array_equal(viaf[1], ref[1])


@pytest.mark.gpu
def test_fit_wider_than_the_closed_form_takes_the_iterative_scheme():
    """n_ev > 200 (the reference has no cap, functional.py:352): the in-LDS solvers of the closed form stop at 200
    columns, wider maps run the reference's own scheme -- L-BFGS on the two quadratic terms, on the device -- to the float64 minimiser:
    within 1e-4 of the oracle's closed form, first column pinned"""
    import types
    import scipy.sparse as sp
    from densematcher_amd import synth
    from densematcher_amd.pyFM import FunctionalMapping
    N, k, D = 1200, 208, 96
    lam1, phi1, a1 = synth.random_basis(N, k, 31)
    lam2, phi2, a2 = synth.random_basis(N, k, 32)
    F1, F2, _ = synth.feature_pair(N, N, D, 5, 6, sigma=0.3, perm="identity")

    def mesh(lam, phi, a):
        m = types.SimpleNamespace(eigenvalues=lam[:k].copy(), eigenvectors=phi[:, :k].astype(np.float64), A=sp.diags(a.astype(np.float64)).tocsr())
        m.process = lambda *a_, **kw: m
        m.area = float(a.astype(np.float64).sum())
        return m
    model = FunctionalMapping(mesh(lam1, phi1, a1), mesh(lam2, phi2, a2), partial=False, optimizer="L-BFGS-B")
    model.preprocess(n_ev=(k, k), n_descr=D, descr1=F1, descr2=F2, subsample_step=1)
    model.fit(w_descr=1e-1, w_lap=1e-3, w_dcomm=0, optinit="zeros")
    Co = orc.fit(phi1[:, :k], phi2[:, :k], lam1[:k], lam2[:k], a1, a2, F1, F2, 1e-1, 1e-3)
    err = np.abs(model.FM - Co).max()
    print("k = 208 fit:", model.fit_result.nit, "iterations; |C - C_oracle| =", err)
    assert err <= 1e-4
    assert np.array_equal(model.FM[:, 0], model.get_x0()[:, 0])

#!/usr/bin/env python
"""
Round-2 golden fixtures, again produced by IMPORTING THE REFERENCE (/root/reference) in the build container
(recipe and stubs: tools/make_golden.py, which this script re-uses).  The stored spectra of the round-1 fixtures are
loaded back into reference TriMesh objects (never recomputed: SURVEY.md Appendix B), so the new vectors belong to
the same inputs.

    fx_cfg2_icp.npz   reference icp_refine (nit = 10) from C_fit at config 2 (N = 2048, k = 128) + the four maps of C_icp
    fx_cfg1_terms.npz reference energy terms p2p / doubly_stochastic / entropy / range01 / sumto1 / op_commutation
                      (values and autograd gradients, float64 torch) at a fixed C, and the reference fit() with the
                      notebook's fit_params (example.ipynb cell 11: w_ent = 0.1, w_sumto1 = 10) and with w_dcomm = 1
    fx_cfg1_notebook_call.npz the 14-tuple of compute_surface_map(compute_extra=True, notebook fit_params) on the config-1 pair
    fx_cfg1_precise.npz reference get_precise_map (barycentric map) of the config-1 pair and its Hungarian assignment
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (imports the reference with the stubs; its __main__ block does not run)

OUT = mg.OUT


def mesh_from_fixture(fx, which):
    phi, lam, a = fx[f"Phi{which}"], fx[f"lam{which}"], fx[f"a{which}"]
    n = phi.shape[0]
    verts = fx[f"verts{which}"] if f"verts{which}" in fx else np.zeros((n, 3))
    faces = fx[f"faces{which}"] if f"faces{which}" in fx else np.zeros((1, 3), dtype=np.int64)
    m = mg.TriMesh(verts, faces)
    m.W = sp.identity(n).tocsr()                       # (not read by icp / fit with the terms used here)
    m.A = sp.diags(a.astype(np.float64)).tocsr()
    m.L = sp.diags(1.0 / a.astype(np.float64)).tocsr() @ m.W
    m.eigenvalues = lam.copy()
    m.eigenvectors = phi.astype(np.float64)
    return m


def case_cfg2_icp():
    fx = dict(np.load(os.path.join(OUT, "fx_cfg2.npz"), allow_pickle=False))
    k = int(fx["k"])
    m1, m2 = mg.truncated(mesh_from_fixture(fx, 1), k), mg.truncated(mesh_from_fixture(fx, 2), k)
    C_icp = mg.ref_refine.icp_refine(fx["C_fit"], m1.eigenvectors, m2.eigenvectors, m1.A, nit=10)   # icp.py:43
    k21, k12, i21, i12, _ = mg.ref_maps(C_icp, m1, m2)
    np.savez_compressed(os.path.join(OUT, "fx_cfg2_icp.npz"), C_icp=C_icp, icp_knn21=k21, icp_knn12=k12, icp_ind21=i21, icp_ind12=i12)
    print("cfg2 icp: orthogonality", np.abs(C_icp.T @ C_icp - np.eye(k)).max())


def case_cfg1_terms():
    fx = dict(np.load(os.path.join(OUT, "fx_cfg1.npz"), allow_pickle=False))
    k = int(fx["k"])
    bf = mg.ref_bf
    e1 = torch.tensor(fx["Phi1"][:, :k].astype(np.float64))
    e2 = torch.tensor(fx["Phi2"][:, :k].astype(np.float64))
    A1 = torch.tensor(np.diag(fx["a1"].astype(np.float64)))
    rng = np.random.default_rng(42)
    Ctest = fx["C_f64"] + 0.05 * rng.standard_normal((k, k))
    out = {"C_test": Ctest}
    for name, fn in (("p2p", bf.p2p), ("stochastic", bf.doubly_stochastic), ("ent", bf.entropy), ("range01", bf.range01),
                     ("sumto1", bf.sumto1)):
        C = torch.tensor(Ctest, requires_grad=True)
        ctx = {}
        loss = fn(C, None, e1, e2, A1, ctx)
        (gkey,) = [kk for kk in ctx if kk.endswith("_grad")]
        out[f"E_{name}"] = float(loss.item())
        out[f"G_{name}"] = ctx[gkey].numpy().copy()
    # descriptor commutativity operators and their energy / gradient (first 8 descriptors keep the fixture small)
    nd = 8
    F1, F2 = fx["F1"].astype(np.float64)[:, :nd], fx["F2"].astype(np.float64)[:, :nd]
    pinv1 = e1.T @ A1
    A2 = torch.tensor(np.diag(fx["a2"].astype(np.float64)))
    pinv2 = e2.T @ A2
    left = [pinv1 @ (torch.tensor(F1[:, i, None]) * e1) for i in range(nd)]
    right = [pinv2 @ (torch.tensor(F2[:, i, None]) * e2) for i in range(nd)]
    C = torch.tensor(Ctest, requires_grad=True)
    e, gC, _ = bf.oplist_commutation(C, None, list(zip(left, right)))
    out.update(E_dcomm=float(e.item()), G_dcomm=gC.numpy().copy(), dcomm_ndescr=nd,
               ops1=np.stack([m.numpy() for m in left]), ops2=np.stack([m.numpy() for m in right]))

    # reference fit() with the notebook's fit_params (cell 11) and with the pyFM default w_dcomm = 1
    m1, m2 = mesh_from_fixture(fx, 1), mesh_from_fixture(fx, 2)
    # fit() builds grad_mat from vertices / faces / normals whatever the weights: give the meshes real geometry
    for m, w in ((m1, 1), (m2, 2)):
        m.W = mg.ref_lap.cotangent_weights(fx[f"verts{w}"], fx[f"faces{w}"])
        m.L = sp.diags(1.0 / fx[f"a{w}"].astype(np.float64)).tocsr() @ m.W
    for tag, params in (("nb", dict(w_descr=1e4, w_lap=1e3, w_dcomm=0, w_ent=1e-1, w_sumto1=1e1, optinit="zeros", maxiter=5000)),
                        ("dcomm", dict(w_descr=1e4, w_lap=1e3, w_dcomm=1, optinit="zeros", maxiter=5000))):
        model = mg.FunctionalMapping(mg.truncated(m1, k), mg.truncated(m2, k), partial=False, optimizer="L-BFGS-B")
        model.preprocess(n_ev=(k, k), n_descr=fx["F1"].shape[1], landmarks=None, descr1=fx["F1"], descr2=fx["F2"], subsample_step=1)
        model.fit(**params, device=mg.CPU)
        out[f"C_fit_{tag}"] = model.FM.copy()
        print(tag, "reference fit done; |C - C_fit(w_descr,w_lap only)| =", np.abs(model.FM - fx["C_fit"]).max())
    np.savez_compressed(os.path.join(OUT, "fx_cfg1_terms.npz"), **out)


def case_cfg1_precise():
    """reference get_precise_map of the config-1 pair from C_fit (sparse triplets) and the assignment of its dense form
    (functional_map.py:62-66: hungarian_precise)"""
    import scipy.optimize
    fx = dict(np.load(os.path.join(OUT, "fx_cfg1.npz"), allow_pickle=False))
    k = int(fx["k"])
    m1, m2 = mg.truncated(mesh_from_fixture(fx, 1), k), mg.truncated(mesh_from_fixture(fx, 2), k)
    model = mg.FunctionalMapping(m1, m2, partial=False, optimizer="L-BFGS-B")
    model.descr1, model.descr2 = fx["F1"], fx["F2"]
    model.FM = fx["C_fit"]
    P = model.get_precise_map().toarray()                                   # functional.py:221-251
    hp = scipy.optimize.linear_sum_assignment(P, maximize=True)
    r, c = np.nonzero(P)
    np.savez_compressed(os.path.join(OUT, "fx_cfg1_precise.npz"), precise_rows=r, precise_cols=c, precise_vals=P[r, c],
                        hungarian_precise_cols=hp[1])
    print("precise map: nnz per row", (P != 0).sum(1).max())


def case_cfg1_notebook_call():
    """the reference's documented call (example.ipynb cell 11): compute_surface_map(..., compute_extra=True, fit_params =
    notebook values) on the config-1 pair, stored spectrum: every integer output of the 14-tuple"""
    fx = dict(np.load(os.path.join(OUT, "fx_cfg1.npz"), allow_pickle=False))
    k = int(fx["k"])
    by_verts = [(fx["verts1"], 1), (fx["verts2"], 2)]
    orig_process = mg.TriMesh.process

    def patched(self, k=200, **kw):
        for vv, which in by_verts:
            if np.array_equal(self.vertlist, vv):
                a = fx[f"a{which}"].astype(np.float64)
                self.W = mg.ref_lap.cotangent_weights(fx[f"verts{which}"], fx[f"faces{which}"])
                self.A = sp.diags(a).tocsr()
                self.L = sp.diags(1.0 / a).tocsr() @ self.W
                self.eigenvalues = fx[f"lam{which}"][:k].copy()
                self.eigenvectors = fx[f"Phi{which}"][:, :k].astype(np.float64)
                return self
        raise RuntimeError("unknown mesh")

    mg.TriMesh.process = patched
    orig_fit = mg.FunctionalMapping.fit
    mg.FunctionalMapping.fit = lambda self, **kw: orig_fit(self, **{**kw, "device": mg.CPU, "verbose": False})
    try:
        res = mg.compute_surface_map(mg._Duck(fx["verts1"], fx["faces1"]), mg._Duck(fx["verts2"], fx["faces2"]), fx["F1"], fx["F2"],
                                     n_ev=k, compute_extra=True, optimizer="L-BFGS-B",
                                     fit_params=dict(w_descr=1e4, w_lap=1e3, w_dcomm=0, w_ent=1e-1, w_sumto1=1e1, optinit="zeros", maxiter=5000))
    finally:
        mg.TriMesh.process = orig_process
        mg.FunctionalMapping.fit = orig_fit
    np.savez_compressed(os.path.join(OUT, "fx_cfg1_notebook_call.npz"),
                        p2p_21=res[0], p2p_12=res[1], hungarian_cols=res[2][1], hungarian_precise_cols=res[3][1],
                        p2p_21_icp=res[4], p2p_12_icp=res[5], hungarian_icp_cols=res[6][1], FM=res[7].FM, FM_base=res[7]._FM_base,
                        p2p_21_adjoint=res[10], p2p_12_adjoint=res[11], p2p_21_icp_adjoint=res[12], p2p_12_icp_adjoint=res[13])
    print("notebook call: done")


if __name__ == "__main__":
    np.random.seed(0)
    torch.manual_seed(0)
    which = sys.argv[1:] or ["icp", "terms", "precise", "notebook"]
    if "icp" in which:
        case_cfg2_icp()
    if "terms" in which:
        case_cfg1_terms()
    if "precise" in which:
        case_cfg1_precise()
    if "notebook" in which:
        case_cfg1_notebook_call()

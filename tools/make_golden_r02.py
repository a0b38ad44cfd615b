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


if __name__ == "__main__":
    np.random.seed(0)
    torch.manual_seed(0)
    which = sys.argv[1:] or ["icp", "terms"]
    if "icp" in which:
        case_cfg2_icp()
    if "terms" in which:
        case_cfg1_terms()

#!/usr/bin/env python
"""
Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE
(/root/reference) in the build container.  The reference cannot travel to the
GPU box; only the .npz vectors written here (inputs + the reference's outputs)
do.  Re-run:   python tools/make_golden.py

Recipe follows SURVEY.md Appendix B:
  * `potpourri3d` / `robust_laplacian` are not installed -> stub modules; the
    stub `mesh_laplacian` returns the reference's own cotangent weights and
    lumped area matrix (pyFM/mesh/laplacian.py:88,5).
  * eigenbases are computed once by the reference (`TriMesh.process`), ROUNDED
    TO FLOAT32 (the dtype the C ABI takes) and written back into fresh
    TriMesh objects, so that the reference and the HIP path consume bit-
    identical inputs.
  * `fit` is always called with optimizer "L-BFGS-B" and device cpu.
  * ZoomOut is run through the 2-line harness repair of SURVEY.md section 0.4
    (slice eigenvectors to the map's size, supply A1, take element [0]).
"""
import os
import sys
import types

import numpy as np
import scipy.optimize
import scipy.sparse as sp
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

sys.modules["potpourri3d"] = types.ModuleType("potpourri3d")
_rl = types.ModuleType("robust_laplacian")
sys.modules["robust_laplacian"] = _rl
from densematcher.pyFM.mesh import laplacian as ref_lap  # noqa: E402

_rl.mesh_laplacian = lambda V, F, mollify_factor=1e-5: (
    ref_lap.cotangent_weights(V, F), ref_lap.dia_area_mat(V, F).tocsr())

from densematcher.pyFM.mesh import TriMesh  # noqa: E402
from densematcher.pyFM.functional import FunctionalMapping  # noqa: E402
import densematcher.pyFM.spectral as ref_spectral  # noqa: E402
import densematcher.pyFM.refine as ref_refine  # noqa: E402
import densematcher.pyFM.refine.zoomout as ref_zo_mod  # noqa: E402
import densematcher.pyFM.optimize.base_functions as ref_bf  # noqa: E402
from densematcher.functional_map import compute_surface_map  # noqa: E402

from densematcher_amd import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
CPU = torch.device("cpu")
W_DESCR, W_LAP = 1e4, 1e3          # notebook cell 11 values
FIT = dict(w_descr=W_DESCR, w_lap=W_LAP, w_dcomm=0, optinit="zeros", maxiter=5000)


class _Duck:
    """What compute_surface_map needs from a pytorch3d Meshes (functional_map.py:17-18)."""
    def __init__(self, v, f):
        self.v, self.f = torch.tensor(v), torch.tensor(f)

    def verts_list(self):
        return [self.v]

    def faces_list(self):
        return [self.f]


def processed_mesh(verts, faces, k):
    """Reference spectrum, rounded to float32, loaded into a fresh TriMesh."""
    m0 = TriMesh(verts, faces)
    m0.process(k, robust=True)
    phi32 = m0.eigenvectors[:, :k].astype(np.float32)
    lam = m0.eigenvalues[:k].astype(np.float64)
    a32 = np.asarray(m0.A.diagonal(), dtype=np.float32)
    return load_mesh(verts, faces, phi32, lam, a32, m0.W), (phi32, lam, a32)


def load_mesh(verts, faces, phi32, lam, a32, W):
    m = TriMesh(verts, faces)
    m.W = W
    m.A = sp.diags(a32.astype(np.float64)).tocsr()
    m.L = sp.diags(1.0 / a32.astype(np.float64)).tocsr() @ W
    m.eigenvalues = lam.copy()
    m.eigenvectors = phi32.astype(np.float64)
    return m


def truncated(mesh, k):
    import copy
    m = copy.deepcopy(mesh)
    m.eigenvalues = m.eigenvalues[:k].copy()
    m.eigenvectors = m.eigenvectors[:, :k].copy()
    return m


def ref_fit(mesh1, mesh2, F1, F2, k):
    model = FunctionalMapping(truncated(mesh1, k), truncated(mesh2, k), partial=False, optimizer="L-BFGS-B")
    model.preprocess(n_ev=(k, k), n_descr=F1.shape[1], landmarks=None, descr1=F1, descr2=F2, subsample_step=1)
    model.fit(**FIT, device=CPU)
    A32 = ref_bf.descr1_red.detach().cpu().numpy().copy()   # fp32 projections cached by the reference
    B32 = ref_bf.descr2_red.detach().cpu().numpy().copy()
    return model, A32, B32


def ref_C_f64(model, F1, F2):
    """float64 L-BFGS-B driven by the reference's own analytic NumPy gradients
    (base_functions.py:58-76, 105-121) and its column-0 pin (:759)."""
    m1, m2 = model.mesh1, model.mesh2
    A = m1.eigenvectors.T @ (m1.A @ F1.astype(np.float64))
    B = m2.eigenvectors.T @ (m2.A @ F2.astype(np.float64))
    scale = max(m1.eigenvalues.max(), m2.eigenvalues.max())
    ev = np.square(m1.eigenvalues[None, :] / scale - m2.eigenvalues[:, None] / scale)
    x0 = model.get_x0(optinit="zeros")
    k2, k1 = x0.shape

    def f(x):
        C = x.reshape(k2, k1)
        return W_DESCR * 0.5 * np.square(C @ A - B).sum() + W_LAP * 0.5 * (np.square(C) * ev).sum()

    def g(x):
        C = x.reshape(k2, k1)
        gr = W_DESCR * ref_bf.descr_preservation_grad(C, A, B) + W_LAP * ref_bf.LB_commutation_grad(C, ev)
        gr[:, 0] = 0
        return gr.ravel()

    res = scipy.optimize.minimize(f, x0.ravel(), jac=g, method="L-BFGS-B",
                                  options={"maxiter": 200000, "maxfun": 2000000, "ftol": 1e-22,
                                           "gtol": 1e-13, "maxcor": 50})
    return res.x.reshape(k2, k1), x0, A, B, ev


def ref_maps(C, mesh1, mesh2):
    k2, k1 = C.shape
    p21, p12, ind = ref_spectral.FM_to_p2p(C, mesh1.eigenvectors[:, :k1], mesh2.eigenvectors[:, :k2], mesh1.A)
    eta = np.ones(ind.shape[0])
    i21 = (ind * eta[..., None]).argmax(axis=1)     # functional_map.py:49
    i12 = (ind * eta[..., None]).argmax(axis=0)     # functional_map.py:50
    return p21.astype(np.int64), p12.astype(np.int64), i21.astype(np.int64), i12.astype(np.int64), ind


def ref_zoomout(C0, mesh1, mesh2, nit, step):
    """zoomout_refine through the harness repair."""
    orig = ref_spectral.FM_to_p2p
    A1 = mesh1.A

    def repaired(FM, e1, e2, n_jobs=1):
        return orig(FM, e1[:, :FM.shape[1]], e2[:, :FM.shape[0]], A1, n_jobs=n_jobs)[0]

    ref_zo_mod.spectral.FM_to_p2p = repaired
    try:
        C, p21 = ref_refine.zoomout_refine(C0, mesh1.eigenvectors, mesh2.eigenvectors, nit, step=step,
                                           A2=mesh2.A, return_p2p=True)
    finally:
        ref_zo_mod.spectral.FM_to_p2p = orig
    return C, p21.astype(np.int64)


# --------------------------------------------------------------------------- #
def case_cfg1():
    """BASELINE.json config 1: N=500 (25x20 torus), D=128, k=30 -- full pair,
    plus ICP, ZoomOut (k 20->40) and the whole compute_surface_map tuple."""
    nu, nv, D, k, kbig = 25, 20, 128, 30, 48
    v1, f1 = synth.torus_mesh(nu, nv)
    v2, f2 = synth.torus_mesh(nu, nv, perturb=0.08, seed=1)
    m1, (phi1, lam1, a1) = processed_mesh(v1, f1, kbig)
    m2, (phi2, lam2, a2) = processed_mesh(v2, f2, kbig)
    F1, F2, perm = synth.feature_pair(nu * nv, nu * nv, D, 1000, 2000, sigma=0.5, perm="identity")

    model, A32, B32 = ref_fit(m1, m2, F1, F2, k)
    C_fit = model.FM.copy()
    C_f64, x0, A64, B64, ev = ref_C_f64(model, F1, F2)
    k21, k12, i21, i12, ind = ref_maps(C_fit, model.mesh1, model.mesh2)
    C_from_p2p = ref_spectral.p2p_to_FM(k21, model.mesh1.eigenvectors, model.mesh2.eigenvectors, A2=model.mesh2.A)
    C_from_p2p_lstsq = ref_spectral.p2p_to_FM(k21, model.mesh1.eigenvectors, model.mesh2.eigenvectors)

    # ICP (functional.py:564 -> icp.py:110 -> :43), nit=10
    model.icp_refine(nit=10)
    C_icp = model.FM.copy()
    ik21, ik12, ii21, ii12, _ = ref_maps(C_icp, model.mesh1, model.mesh2)

    # ZoomOut 20 -> 40, step 1, and 20 -> 44 with step 4
    C20 = ref_spectral.p2p_to_FM(k21, m1.eigenvectors[:, :20], m2.eigenvectors[:, :20], A2=m2.A)
    C_zo, p21_zo = ref_zoomout(C20, m1, m2, nit=20, step=1)
    C_zo4, p21_zo4 = ref_zoomout(C20, m1, m2, nit=6, step=4)

    # whole compute_surface_map on the SAME float32-rounded spectrum: patch
    # TriMesh.process so the fresh meshes it builds load the stored basis.
    stored = {id(None): None}
    by_nverts = [(v1, (phi1, lam1, a1, m1.W)), (v2, (phi2, lam2, a2, m2.W))]
    orig_process = TriMesh.process

    def patched(self, k=200, **kw):
        for vv, (p, l, a, W) in by_nverts:
            if np.array_equal(self.vertlist, vv):
                self.W = W
                self.A = sp.diags(a.astype(np.float64)).tocsr()
                self.L = sp.diags(1.0 / a.astype(np.float64)).tocsr() @ W
                self.eigenvalues = l[:k].copy()
                self.eigenvectors = p[:, :k].astype(np.float64)
                return self
        raise RuntimeError("unknown mesh")

    TriMesh.process = patched
    try:
        orig_fit = FunctionalMapping.fit
        FunctionalMapping.fit = lambda self, **kw: orig_fit(self, **{**kw, "device": CPU, "verbose": False})
        res = compute_surface_map(_Duck(v1, f1), _Duck(v2, f2), F1, F2, n_ev=k, optimizer="L-BFGS-B",
                                  fit_params=dict(FIT))
    finally:
        TriMesh.process = orig_process
        FunctionalMapping.fit = orig_fit
    del stored

    np.savez_compressed(
        os.path.join(OUT, "fx_cfg1.npz"),
        verts1=v1, faces1=f1, verts2=v2, faces2=f2,
        Phi1=phi1, Phi2=phi2, lam1=lam1, lam2=lam2, a1=a1, a2=a2, F1=F1, F2=F2, perm=perm,
        k=k, w_descr=W_DESCR, w_lap=W_LAP,
        x0=x0, A_f32=A32, B_f32=B32, A_f64=A64, B_f64=B64, ev_sqdiff=ev,
        C_fit=C_fit, C_f64=C_f64,
        knn21=k21, knn12=k12, ind21=i21, ind12=i12,
        ind_rows=ind[::50].copy(), ind_row_ids=np.arange(0, ind.shape[0], 50),
        C_from_p2p=C_from_p2p, C_from_p2p_lstsq=C_from_p2p_lstsq,
        C_icp=C_icp, icp_knn21=ik21, icp_knn12=ik12, icp_ind21=ii21, icp_ind12=ii12,
        C20=C20, C_zo=C_zo, p21_zo=p21_zo, C_zo4=C_zo4, p21_zo4=p21_zo4,
        csm_p2p_21=res[0], csm_p2p_12=res[1], csm_p2p_21_icp=res[4], csm_p2p_12_icp=res[5],
        csm_hungarian_icp_rows=res[6][0], csm_hungarian_icp_cols=res[6][1],
        csm_FM=res[7].FM, csm_FM_base=res[7]._FM_base,
        csm_p2p_21_adjoint=res[10], csm_p2p_12_adjoint=res[11],
        csm_p2p_21_icp_adjoint=res[12], csm_p2p_12_icp_adjoint=res[13],
    )
    print("cfg1: |C_fit - C_f64| =", np.abs(C_fit - C_f64).max(),
          " ind21==perm:", (i21 == perm).mean(), " csm==maps:", (res[0] == i21).all())
    return dict(m1=m1, m2=m2, F1=F1, F2=F2, C_fit=C_fit)


def case_ties(base):
    """Adversarial ties: duplicate rows in BOTH eigenbases (duplicated
    vertices) so several arg-reductions have exact ties."""
    m1, m2, C = base["m1"], base["m2"], base["C_fit"]
    k = C.shape[0]
    phi1 = m1.eigenvectors[:, :k].astype(np.float32).copy()
    phi2 = m2.eigenvectors[:, :k].astype(np.float32).copy()
    a1 = np.asarray(m1.A.diagonal(), dtype=np.float32).copy()
    rng = np.random.default_rng(7)
    src = rng.choice(500, size=40, replace=False)
    dst = rng.choice(np.setdiff1d(np.arange(500), src), size=40, replace=False)
    phi1[dst] = phi1[src]
    a1[dst] = a1[src]
    src2 = rng.choice(500, size=40, replace=False)
    dst2 = rng.choice(np.setdiff1d(np.arange(500), src2), size=40, replace=False)
    phi2[dst2] = phi2[src2]
    p21, p12, ind = ref_spectral.FM_to_p2p(C, phi1.astype(np.float64), phi2.astype(np.float64),
                                           sp.diags(a1.astype(np.float64)).tocsr())
    i21 = ind.argmax(axis=1)
    i12 = ind.argmax(axis=0)
    np.savez_compressed(os.path.join(OUT, "fx_ties.npz"), Phi1=phi1, Phi2=phi2, a1=a1, C=C,
                        knn21=p21.astype(np.int64), knn12=p12.astype(np.int64), ind21=i21, ind12=i12,
                        dup1_src=src, dup1_dst=dst, dup2_src=src2, dup2_dst=dst2)
    print("ties: written")


def case_cfg2():
    """BASELINE.json config 2 shape: N=2048 (64x32 torus), D=768, k=128.
    Descriptors are regenerated from seeds (sha256 pinned)."""
    nu, nv, D, k, kbig = 64, 32, 768, 128, 140
    v1, f1 = synth.torus_mesh(nu, nv)
    v2, f2 = synth.torus_mesh(nu, nv, perturb=0.08, seed=1)
    m1, (phi1, lam1, a1) = processed_mesh(v1, f1, kbig)
    m2, (phi2, lam2, a2) = processed_mesh(v2, f2, kbig)
    F1, F2, perm = synth.feature_pair(nu * nv, nu * nv, D, 1000, 2000, sigma=1.0, perm="identity")
    model, A32, B32 = ref_fit(m1, m2, F1, F2, k)
    C_fit = model.FM.copy()
    C_f64, x0, A64, B64, ev = ref_C_f64(model, F1, F2)
    k21, k12, i21, i12, _ = ref_maps(C_fit, model.mesh1, model.mesh2)
    f21, f12, fi21, fi12, _ = ref_maps(C_f64, model.mesh1, model.mesh2)
    # short ZoomOut at full N: 128 -> 140 step 4
    C_zo, p21_zo = ref_zoomout(C_fit, m1, m2, nit=3, step=4)
    np.savez_compressed(
        os.path.join(OUT, "fx_cfg2.npz"),
        Phi1=phi1, Phi2=phi2, lam1=lam1, lam2=lam2, a1=a1, a2=a2,
        feat_seeds=np.array([1000, 2000]), feat_sigma=1.0, feat_sha256=synth.sha256_of(F1, F2), D=D, k=k,
        w_descr=W_DESCR, w_lap=W_LAP, x0_col0=x0[:, 0],
        C_fit=C_fit, C_f64=C_f64,
        knn21=k21, knn12=k12, ind21=i21, ind12=i12,
        f64_knn21=f21, f64_knn12=f12, f64_ind21=fi21, f64_ind12=fi12,
        C_zo=C_zo, p21_zo=p21_zo,
    )
    print("cfg2: |C_fit - C_f64| =", np.abs(C_fit - C_f64).max(),
          " map agreement fit-vs-f64:", (k21 == f21).mean(), (i21 == fi21).mean(),
          " ind21==perm:", (i21 == perm).mean())



def case_signatures():
    """HKS / WKS descriptors (pyFM/signatures) of the config-1 meshes, plain and landmark versions, and the
    functional map the reference fits on HKS descriptors (functional.py:308-329 -> fit)."""
    import densematcher.pyFM.signatures as ref_sg
    nu, nv, k, kbig = 25, 20, 30, 48
    v1, f1 = synth.torus_mesh(nu, nv)
    v2, f2 = synth.torus_mesh(nu, nv, perturb=0.08, seed=1)
    m1, (phi1, lam1, a1) = processed_mesh(v1, f1, kbig)
    m2, (phi2, lam2, a2) = processed_mesh(v2, f2, kbig)
    lm = np.array([3, 77, 410])
    lm2 = np.array([[3, 5], [77, 90], [410, 400]])
    wks_big = ref_sg.mesh_WKS(m1, 2048, k=k)
    t1, t2 = truncated(m1, k), truncated(m2, k)
    model = FunctionalMapping(t1, t2, partial=False, optimizer="L-BFGS-B")
    # (process() would recompute the spectrum: the truncated meshes already hold exactly k eigenpairs, so the
    #  reference's `process(max(k, 1))` is a no-op only if patched -- same trick as compute_surface_map above)
    orig_process = TriMesh.process
    TriMesh.process = lambda self, k=200, **kw: self
    try:
        model.preprocess(n_ev=(k, k), n_descr=16, descr_type="HKS", landmarks=lm2, subsample_step=2)
    finally:
        TriMesh.process = orig_process
    d1, d2 = model.descr1.copy(), model.descr2.copy()
    model.fit(**FIT, device=CPU)
    np.savez_compressed(
        os.path.join(OUT, "fx_sig.npz"),
        Phi1=phi1, lam1=lam1, a1=a1, Phi2=phi2, lam2=lam2, a2=a2, k=k, landmarks=lm, landmarks2=lm2,
        hks=ref_sg.mesh_HKS(m1, 16, k=k), wks=ref_sg.mesh_WKS(m1, 24, k=k),
        hks_lm=ref_sg.mesh_HKS(m1, 5, landmarks=lm, k=k), wks_lm=ref_sg.mesh_WKS(m1, 7, landmarks=lm, k=k),
        hks_allk=ref_sg.mesh_HKS(m2, 9), wks_big_cols=wks_big[:, ::64].copy(), wks_big_sum=wks_big.sum(axis=1),
        pre_descr1=d1, pre_descr2=d2, C_fit_hks=model.FM.copy(),
    )
    print("signatures: hks", d1.shape, "C_fit_hks", model.FM.shape)


if __name__ == "__main__":
    np.random.seed(0)
    torch.manual_seed(0)
    if "--signatures-only" in sys.argv:
        case_signatures()
    else:
        base = case_cfg1()
        case_ties(base)
        case_cfg2()
        case_signatures()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))

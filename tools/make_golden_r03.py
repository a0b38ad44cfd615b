#!/usr/bin/env python
"""
Round-3 golden fixture, produced by IMPORTING THE REFERENCE (/root/reference) in the build container (recipe and stubs:
tools/make_golden.py, which this script re-uses).

    fx_cfg2_f64.npz   BASELINE config-2 shape (N = 2048, D = 768, k = 128) on the reference's UN-ROUNDED float64 spectrum:
                      TriMesh.process() output (eigenvectors, eigenvalues, lumped masses, all float64) is stored as it is
                      and the reference consumes exactly those arrays:
                        C_fit            FunctionalMapping.fit (functional.py:352-487; fit itself rounds to fp32, :410-414)
                        knn21 .. ind12   FM_to_p2p on the float64 basis + the indicator arg-maxes (convert.py:134-144,
                                         functional_map.py:49-50)
                        C_from_p2p(_lstsq)  p2p_to_FM with / without A2 (convert.py:39-51)
                        C_icp + maps     icp_refine, nit = 10 (icp.py:43-107)
                        C_zo, p21_zo     zoomout_refine 128 -> 136, step 4 (harness repair of SURVEY.md 0.4)
                      and, for the record, how many map entries change when the SAME reference code is fed the basis
                      rounded to float32 (what the round-1/2 fixtures did): *_r32 arrays.
The earlier fixtures (fx_cfg2.npz ...) hold a float32-rounded basis; this one closes the gap VERDICT r02 names: parity on
the reference's real inputs.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (imports the reference with the stubs; its __main__ block does not run)

OUT = mg.OUT


def mesh_with(verts, faces, phi, lam, a, W):
    m = mg.TriMesh(verts, faces)
    m.W = W
    m.A = sp.diags(np.asarray(a, dtype=np.float64)).tocsr()
    m.L = sp.diags(1.0 / np.asarray(a, dtype=np.float64)).tocsr() @ W
    m.eigenvalues = np.asarray(lam, dtype=np.float64).copy()
    m.eigenvectors = np.asarray(phi, dtype=np.float64).copy()
    return m


def case_cfg2_f64():
    nu, nv, D, k, kbig = 64, 32, 768, 128, 136
    v1, f1 = mg.synth.torus_mesh(nu, nv)
    v2, f2 = mg.synth.torus_mesh(nu, nv, perturb=0.08, seed=1)
    raw = []
    for v, f in ((v1, f1), (v2, f2)):
        m0 = mg.TriMesh(v, f)
        m0.process(kbig, robust=True)                          # trimesh.py:498-531: float64 spectrum, never rounded
        raw.append((m0.eigenvectors[:, :kbig].copy(), m0.eigenvalues[:kbig].copy(), np.asarray(m0.A.diagonal()).copy(), m0.W))
    (phi1, lam1, a1, W1), (phi2, lam2, a2, W2) = raw
    assert phi1.dtype == np.float64 and a1.dtype == np.float64
    m1, m2 = mesh_with(v1, f1, phi1, lam1, a1, W1), mesh_with(v2, f2, phi2, lam2, a2, W2)
    F1, F2, perm = mg.synth.feature_pair(nu * nv, nu * nv, D, 1000, 2000, sigma=1.0, perm="identity")

    model, A32, B32 = mg.ref_fit(m1, m2, F1, F2, k)
    C_fit = model.FM.copy()
    x0 = model.get_x0(optinit="zeros")
    k21, k12, i21, i12, _ = mg.ref_maps(C_fit, model.mesh1, model.mesh2)
    C_from_p2p = mg.ref_spectral.p2p_to_FM(k21, model.mesh1.eigenvectors, model.mesh2.eigenvectors, A2=model.mesh2.A)
    C_from_p2p_lstsq = mg.ref_spectral.p2p_to_FM(k21, model.mesh1.eigenvectors, model.mesh2.eigenvectors)
    C_icp = mg.ref_refine.icp_refine(C_fit, model.mesh1.eigenvectors, model.mesh2.eigenvectors, model.mesh1.A, nit=10)
    ik21, ik12, ii21, ii12, _ = mg.ref_maps(C_icp, model.mesh1, model.mesh2)
    C_zo, p21_zo = mg.ref_zoomout(C_fit, m1, m2, nit=2, step=4)

    # the same reference code on the basis rounded to float32 (masses too): what a float32 boundary computes
    r1 = mesh_with(v1, f1, phi1.astype(np.float32), lam1, a1.astype(np.float32), W1)
    r2 = mesh_with(v2, f2, phi2.astype(np.float32), lam2, a2.astype(np.float32), W2)
    t1, t2 = mg.truncated(r1, k), mg.truncated(r2, k)
    rk21, rk12, ri21, ri12, _ = mg.ref_maps(C_fit, t1, t2)
    rik21, rik12, rii21, rii12, _ = mg.ref_maps(C_icp, t1, t2)

    np.savez_compressed(
        os.path.join(OUT, "fx_cfg2_f64.npz"),
        Phi1=phi1, Phi2=phi2, lam1=lam1, lam2=lam2, a1=a1, a2=a2,
        feat_seeds=np.array([1000, 2000]), feat_sigma=1.0, feat_sha256=mg.synth.sha256_of(F1, F2), D=D, k=k,
        w_descr=mg.W_DESCR, w_lap=mg.W_LAP, x0_col0=x0[:, 0],
        C_fit=C_fit, knn21=k21, knn12=k12, ind21=i21, ind12=i12,
        C_from_p2p=C_from_p2p, C_from_p2p_lstsq=C_from_p2p_lstsq,
        C_icp=C_icp, icp_knn21=ik21, icp_knn12=ik12, icp_ind21=ii21, icp_ind12=ii12,
        C_zo=C_zo, p21_zo=p21_zo,
        knn21_r32=rk21, knn12_r32=rk12, ind21_r32=ri21, ind12_r32=ri12,
        icp_knn21_r32=rik21, icp_knn12_r32=rik12, icp_ind21_r32=rii21, icp_ind12_r32=rii12,
    )
    diff = {n: int((a != b).sum()) for n, a, b in (("knn21", k21, rk21), ("knn12", k12, rk12), ("ind21", i21, ri21),
                                                    ("ind12", i12, ri12), ("icp_knn21", ik21, rik21), ("icp_knn12", ik12, rik12),
                                                    ("icp_ind21", ii21, rii21), ("icp_ind12", ii12, rii12))}
    print("cfg2 f64 basis: entries (of 2048) that change when the reference is fed the float32-rounded basis:", diff)
    print("  max |Phi - fp32(Phi)| =", np.abs(phi1 - phi1.astype(np.float32)).max(), " ind21 == perm:", (i21 == perm).mean())


if __name__ == "__main__":
    np.random.seed(0)
    mg.torch.manual_seed(0)
    case_cfg2_f64()
    p = os.path.join(OUT, "fx_cfg2_f64.npz")
    print(p, os.path.getsize(p))

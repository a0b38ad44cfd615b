"""
Spectral ICP with the reference's signatures (densematcher/pyFM/refine/icp.py); the loop runs on the GPU (dm_icp).
"""
import warnings

import numpy as np


def icp_host_svd(FM_12, evects1, evects2, nit):
    """`nit` iterations of the reference's own arithmetic (icp.py:36-40) for ONE pair whose least-squares map the device cannot
    orthogonalise: vertex map on the GPU (exact float64 nearest neighbour, as everywhere), then scipy.linalg.lstsq and
    scipy.linalg.svd on the host -- the calls the reference makes.  The device path replaces the SVD by a polar iteration
    (U V^T = X (X^T X)^-1/2), which needs X of full rank; LAPACK's U I V^T exists for any rank (the completion it picks for a null
    space is LAPACK's choice: no other implementation can reproduce it, this one IS it).  A rare path (a vertex map with fewer
    distinct images than the map has columns), taken per pair, announced by a warning."""
    import scipy.linalg
    from ...engine import default_engine
    from ..spectral.convert import _basis, _real_dtype
    FM = np.array(FM_12, dtype=np.float64)
    k2, k1 = FM.shape
    eng = default_engine()
    dt = _real_dtype(evects1, evects2)
    P1, P2 = _basis(evects1, k1, dt), _basis(evects2, k2, dt)
    e1, e2 = np.asarray(P1[0], dtype=np.float64), np.asarray(P2[0], dtype=np.float64)
    for _ in range(int(nit)):
        p21 = eng.fm_to_p2p(P1, P2, None, FM[None], knn=True, ind=False)["knn21"][0].cpu().numpy().astype(np.int64)
        FM_icp = scipy.linalg.lstsq(e2, e1[p21])[0]                           # convert.py:51
        U, _, VT = scipy.linalg.svd(FM_icp)                                   # icp.py:38-40
        FM = U @ np.eye(k2, k1) @ VT
    return FM


def _run(FM_12, evects1, evects2, nit):
    from ...engine import default_engine
    from ..spectral.convert import _basis, _real_dtype
    FM_12 = np.asarray(FM_12, dtype=np.float64)
    k2, k1 = FM_12.shape
    eng = default_engine()
    dt = _real_dtype(evects1, evects2)            # float64 eigenvectors: the float64-basis kernels (reference icp.py:36-40)
    C, resid, info = eng.icp(_basis(evects1, k1, dt), _basis(evects2, k2, dt), FM_12[None], nit, return_resid=True)
    if int(info[0]) != 0 or not float(resid[0]) <= 1e-8:
        # (the reference's lstsq + SVD take any rank, icp.py:38-40: this pair re-runs with exactly those calls)
        warnings.warn("ICP: " + ("Phi2^T Phi2 is not positive definite" if int(info[0]) != 0 else
                                 f"the polar iteration did not converge (|C^T C - I| = {float(resid[0]):.2e}: the least-squares map is close to "
                                 "rank deficient)") + "; this pair runs the reference's lstsq + SVD on the host")
        return icp_host_svd(FM_12, evects1, evects2, nit)
    return C[0].cpu().numpy()


def icp_iteration(FM_12, evects1, evects2, A1=None, use_adj=False, n_jobs=1):
    """reference icp.py:10-40"""
    return _run(FM_12, evects1, evects2, 1)


def icp_refine(FM_12, evects1, evects2, A1=None, nit=10, tol=1e-10, use_adj=False, return_p2p=False, n_jobs=1, verbose=False):
    """reference icp.py:43-107.  A fixed iteration count runs as one GPU call; with nit = None / 0 the reference iterates
    until max |C_new - C| <= tol (at most 10000 times, :84-96): one GPU iteration per host decision, as there."""
    if nit is not None and nit > 0:
        FM_icp = _run(FM_12, evects1, evects2, nit)
    else:
        FM_curr = np.array(FM_12, dtype=np.float64)
        for _ in range(10000):
            FM_icp = _run(FM_curr, evects1, evects2, 1)
            if np.max(np.abs(FM_curr - FM_icp)) <= tol:
                break
            FM_curr = FM_icp
    if return_p2p:
        from .. import spectral
        p2p_21 = spectral.FM_to_p2p(FM_icp, evects1, evects2, A1)[0]
        return FM_icp, p2p_21
    return FM_icp


def mesh_icp_refine(FM_12, mesh1, mesh2, nit=10, tol=1e-10, use_adj=False, return_p2p=False, n_jobs=1, verbose=False):
    """reference icp.py:110-150"""
    k2, k1 = np.asarray(FM_12).shape
    return icp_refine(FM_12, mesh1.eigenvectors[:, :k1], mesh2.eigenvectors[:, :k2], mesh1.A, nit=nit, tol=tol, use_adj=use_adj,
                      return_p2p=return_p2p, n_jobs=n_jobs, verbose=verbose)

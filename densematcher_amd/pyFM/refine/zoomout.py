"""
ZoomOut with the reference's signatures (densematcher/pyFM/refine/zoomout.py), upstream-pyFM semantics for the
FM -> p2p step (the fork's call at zoomout.py:40,112 is broken as shipped).
The whole loop runs on the GPU without host synchronisation (dm_zoomout).
"""
import numpy as np

from .. import spectral


def _steps(step):
    try:
        step1, step2 = step
    except TypeError:
        step1 = step2 = step
    return step1, step2


def _knn21(eng, FM, evects1, evects2):
    """p2p_21 of upstream-pyFM FM_to_p2p: NN(tree = Phi1[:, :k1] C^T, query = Phi2[:, :k2])"""
    from ..spectral.convert import _basis, _real_dtype
    k2, k1 = FM.shape
    dt = _real_dtype(evects1, evects2)
    out = eng.fm_to_p2p(_basis(evects1, k1, dt), _basis(evects2, k2, dt), None, np.ascontiguousarray(FM)[None], knn=True, ind=False)
    return out["knn21"]


def _run(FM_12, evects1, evects2, nit, step, A2, return_p2p):
    """A2 given, square map, one step size: the fused GPU loop (dm_zoomout, no host synchronisation).  Rectangular maps,
    two step sizes, or A2 = None (least-squares p2p_to_FM, what the reference does on subsampled vertices): the same two
    GPU kernels per iteration, chained from the host like the reference chains its two calls (zoomout.py:40-42)."""
    from ...engine import default_engine
    from ..spectral.convert import _basis, _diag_of, _real_dtype
    dt = _real_dtype(evects1, evects2)
    step1, step2 = _steps(step)
    k2_0, k1_0 = FM_12.shape
    eng = default_engine()
    FM_12 = np.ascontiguousarray(FM_12, dtype=np.float64)
    if A2 is not None and step1 == step2 and k1_0 == k2_0:
        a2 = _diag_of(A2, evects2.shape[0]).astype(dt)
        kf = k1_0 + nit * step1
        res = eng.zoomout(_basis(evects1, kf, dt), _basis(evects2, kf, dt), a2[None], FM_12[None], nit, step1, return_p2p=return_p2p)
        if return_p2p:
            return res[0][0].cpu().numpy(), res[1][0].cpu().numpy().astype(np.int64)
        return res[0].cpu().numpy()
    a2 = None if A2 is None else _diag_of(A2, evects2.shape[0]).astype(dt)
    FM = FM_12
    for _ in range(nit):
        k2, k1 = FM.shape
        p21 = _knn21(eng, FM, evects1, evects2)
        E1 = _basis(evects1, k1 + step1, dt)
        E2 = _basis(evects2, k2 + step2, dt)
        if a2 is None:
            FM = eng.p2p_to_fm_lstsq(p21, E1, E2, k1 + step1, k2 + step2)[0].cpu().numpy()
        else:
            FM = eng.p2p_to_fm(p21, E1, E2, a2[None], k1 + step1, k2 + step2)[0].cpu().numpy()
    if return_p2p:
        return FM, _knn21(eng, FM, evects1, evects2)[0].cpu().numpy().astype(np.int64)
    return FM


def zoomout_iteration(FM_12, evects1, evects2, step=1, A2=None, n_jobs=1):
    """reference zoomout.py:7-44"""
    return _run(np.asarray(FM_12), evects1, evects2, 1, step, A2, False)


def zoomout_refine(FM_12, evects1, evects2, nit=10, step=1, A2=None, subsample=None, return_p2p=False, n_jobs=1, verbose=False):
    """reference zoomout.py:47-115.  subsample = (sub1, sub2): the iterations run on those vertices with the least-squares
    p2p_to_FM (:100-102), the final vertex map on all vertices (:111-113)."""
    FM_12 = np.asarray(FM_12)
    k2_0, k1_0 = FM_12.shape
    step1, step2 = _steps(step)
    assert k1_0 + nit * step1 <= evects1.shape[1], \
        f"Not enough eigenvectors on source : {k1_0 + nit * step1} are needed when {evects1.shape[1]} are provided"
    assert k2_0 + nit * step2 <= evects2.shape[1], \
        f"Not enough eigenvectors on target : {k2_0 + nit * step2} are needed when {evects2.shape[1]} are provided"
    if subsample is not None:
        sub1, sub2 = subsample
        FM = _run(FM_12, np.asarray(evects1)[sub1], np.asarray(evects2)[sub2], nit, step, None, False)
        if return_p2p:
            from ...engine import default_engine
            return FM, _knn21(default_engine(), FM, evects1, evects2)[0].cpu().numpy().astype(np.int64)
        return FM
    return _run(FM_12, evects1, evects2, nit, step, A2, return_p2p)


def mesh_zoomout_refine(FM_12, mesh1, mesh2, nit=10, step=1, subsample=None, return_p2p=False, n_jobs=1, verbose=False):
    """reference zoomout.py:118-161"""
    if np.issubdtype(type(subsample), np.integer):                              # zoomout.py:151-155
        if verbose:
            print(f'Computing farthest point sampling of size {subsample}')
        subsample = (mesh1.extract_fps(subsample), mesh2.extract_fps(subsample))
    return zoomout_refine(FM_12, mesh1.eigenvectors, mesh2.eigenvectors, nit, step=step, A2=mesh2.A, subsample=subsample,
                          return_p2p=return_p2p, n_jobs=n_jobs, verbose=verbose)


def mesh_zoomout_refine_p2p(p2p_21, mesh1, mesh2, k_init, nit=10, step=1, subsample=None, return_p2p=False, n_jobs=1,
                            p2p_on_sub=False, verbose=False):
    """reference zoomout.py:164-217"""
    if subsample is not None or p2p_on_sub:
        raise NotImplementedError("farthest-point subsampling is outside the matching path")
    FM_12_init = spectral.mesh_p2p_to_FM(p2p_21, mesh1, mesh2, dims=k_init, subsample=None)
    return zoomout_refine(FM_12_init, mesh1.eigenvectors, mesh2.eigenvectors, nit, step=step, A2=mesh2.A,
                          return_p2p=return_p2p, n_jobs=n_jobs, verbose=verbose)

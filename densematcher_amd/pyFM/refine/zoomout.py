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


def _run(FM_12, evects1, evects2, nit, step, A2, return_p2p):
    from ...engine import default_engine
    from ..spectral.convert import _diag_of
    step1, step2 = _steps(step)
    k2_0, k1_0 = FM_12.shape
    if step1 != step2 or k1_0 != k2_0:
        raise NotImplementedError("the GPU ZoomOut handles square maps with one step size")
    eng = default_engine()
    a2 = _diag_of(A2, evects2.shape[0])
    kf = k1_0 + nit * step1
    res = eng.zoomout(np.ascontiguousarray(evects1[:, :kf], dtype=np.float32)[None],
                      np.ascontiguousarray(evects2[:, :kf], dtype=np.float32)[None], a2[None],
                      np.ascontiguousarray(FM_12, dtype=np.float64)[None], nit, step1, return_p2p=return_p2p)
    if return_p2p:
        return res[0][0].cpu().numpy(), res[1][0].cpu().numpy().astype(np.int64)
    return res[0].cpu().numpy()


def zoomout_iteration(FM_12, evects1, evects2, step=1, A2=None, n_jobs=1):
    """reference zoomout.py:7-44"""
    if A2 is None:
        raise NotImplementedError("ZoomOut on subsampled eigenvectors (least-squares p2p_to_FM) is not on the GPU path")
    return _run(np.asarray(FM_12), evects1, evects2, 1, step, A2, False)


def zoomout_refine(FM_12, evects1, evects2, nit=10, step=1, A2=None, subsample=None, return_p2p=False, n_jobs=1, verbose=False):
    """reference zoomout.py:47-115"""
    FM_12 = np.asarray(FM_12)
    k2_0, k1_0 = FM_12.shape
    step1, step2 = _steps(step)
    assert k1_0 + nit * step1 <= evects1.shape[1], \
        f"Not enough eigenvectors on source : {k1_0 + nit * step1} are needed when {evects1.shape[1]} are provided"
    assert k2_0 + nit * step2 <= evects2.shape[1], \
        f"Not enough eigenvectors on target : {k2_0 + nit * step2} are needed when {evects2.shape[1]} are provided"
    if subsample is not None or A2 is None:
        raise NotImplementedError("ZoomOut on subsampled eigenvectors (least-squares p2p_to_FM) is not on the GPU path")
    return _run(FM_12, evects1, evects2, nit, step, A2, return_p2p)


def mesh_zoomout_refine(FM_12, mesh1, mesh2, nit=10, step=1, subsample=None, return_p2p=False, n_jobs=1, verbose=False):
    """reference zoomout.py:118-161"""
    if subsample is not None:
        raise NotImplementedError("farthest-point subsampling is outside the matching path")
    return zoomout_refine(FM_12, mesh1.eigenvectors, mesh2.eigenvectors, nit, step=step, A2=mesh2.A, subsample=None,
                          return_p2p=return_p2p, n_jobs=n_jobs, verbose=verbose)


def mesh_zoomout_refine_p2p(p2p_21, mesh1, mesh2, k_init, nit=10, step=1, subsample=None, return_p2p=False, n_jobs=1,
                            p2p_on_sub=False, verbose=False):
    """reference zoomout.py:164-217"""
    if subsample is not None or p2p_on_sub:
        raise NotImplementedError("farthest-point subsampling is outside the matching path")
    FM_12_init = spectral.mesh_p2p_to_FM(p2p_21, mesh1, mesh2, dims=k_init, subsample=None)
    return zoomout_refine(FM_12_init, mesh1.eigenvectors, mesh2.eigenvectors, nit, step=step, A2=mesh2.A,
                          return_p2p=return_p2p, n_jobs=n_jobs, verbose=verbose)

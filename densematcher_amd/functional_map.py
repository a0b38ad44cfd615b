"""
compute_surface_map with the reference's signature and 14-tuple (densematcher/functional_map.py:9-81).
"""
import os
import time

import numpy as np

from .pyFM.functional import FunctionalMapping
from .pyFM.mesh import TriMesh


def _np(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def compute_surface_map(mesh1_t, mesh2_t, c1, c2, n_ev=50, compute_extra=False, optimizer="fmin_l_bfgs_b", descr_type="neural",
                        maxiter=100000, optimize_p2p=False, fit_params=None):
    '''
    Returns (reference functional_map.py:81):
        p2p_21, p2p_12, hungarian, hungarian_precise, p2p_21_icp, p2p_12_icp, hungarian_icp, model, mesh1, mesh2,
        p2p_21_adjoint, p2p_12_adjoint, p2p_21_icp_adjoint, p2p_12_icp_adjoint
    mesh arguments only need `.verts_list()[0]` / `.faces_list()[0]` (functional_map.py:17-18).
    p2p_21 / p2p_12 are the indicator arg-max maps (:49-50), the *_adjoint ones the kd-tree maps (:48).
    '''
    assert descr_type in ["neural", "HKS", "WKS"]
    mesh1 = TriMesh(_np(mesh1_t.verts_list()[0]), _np(mesh1_t.faces_list()[0]))
    mesh2 = TriMesh(_np(mesh2_t.verts_list()[0]), _np(mesh2_t.faces_list()[0]))
    if descr_type == "neural":
        process_params = {'n_ev': (n_ev, n_ev), 'n_descr': c1.shape[1], 'landmarks': None, 'descr1': _np(c1), 'descr2': _np(c2),
                          'subsample_step': 1}
    else:                                     # spectral signatures instead of network features (functional_map.py:19-35)
        process_params = {'n_ev': (n_ev, n_ev), 'n_descr': 16 if descr_type == "HKS" else 2048, 'landmarks': None,
                          'descr_type': descr_type, 'subsample_step': 1}
    model = FunctionalMapping(mesh1, mesh2, partial=False, optimizer=optimizer)
    model.preprocess(**process_params, verbose=False)
    fit_params = dict(fit_params or {})
    fit_params.pop("verbose", None)
    model.fit(**fit_params)
    p2p_21_adjoint, p2p_12_adjoint = model.get_p2p(n_jobs=1)                    # sets model.mapped_indicator
    p2p_21 = (model.mapped_indicator * model.eta[..., None]).argmax(axis=1)     # functional_map.py:49
    p2p_12 = (model.mapped_indicator * model.eta[..., None]).argmax(axis=0)     # functional_map.py:50

    timing = os.environ.get("TIMEIT", False)
    hungarian = hungarian_precise = None
    if compute_extra:
        # Hungarian on the dense indicator and the precise (barycentric) map: SURVEY.md 'next #3', not accelerated yet
        raise NotImplementedError("compute_extra (Hungarian on the plain map, precise map) is not on the accelerated path yet")

    start_s = time.time()
    model.icp_refine()                                                          # functional_map.py:71, nit=10
    if timing:
        print("ICP refinement took", time.time() - start_s, "seconds")
    p2p_21_icp_adjoint, p2p_12_icp_adjoint = model.get_p2p(n_jobs=1)
    p2p_21_icp = (model.mapped_indicator * model.eta[..., None]).argmax(axis=1)
    p2p_12_icp = (model.mapped_indicator * model.eta[..., None]).argmax(axis=0)
    # functional_map.py:78: the reference always runs the Hungarian algorithm on the ICP indicator (host SciPy,
    # 1-2 s at N = 2048).  It is NOT on the accelerated path (SURVEY.md 'next #3'): this is the reference's own SciPy
    # call applied to the GPU-built dense matrix, kept so that the 14-tuple is complete.  DENSEMATCHER_HUNGARIAN=0
    # skips it (slot 6 = None).
    hungarian_icp = None
    if os.environ.get("DENSEMATCHER_HUNGARIAN", "1") != "0":
        from scipy.optimize import linear_sum_assignment
        hungarian_icp = linear_sum_assignment(np.asarray(model.mapped_indicator), maximize=True)
    return (p2p_21, p2p_12, hungarian, hungarian_precise, p2p_21_icp, p2p_12_icp, hungarian_icp, model, model.mesh1, model.mesh2,
            p2p_21_adjoint, p2p_12_adjoint, p2p_21_icp_adjoint, p2p_12_icp_adjoint)

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


def _assign(matrix):
    """scipy.optimize.linear_sum_assignment(matrix, maximize=True) (functional_map.py:57,66,78; eta == 1 after fit) on the GPU:
    same algorithm and tie rules as SciPy, identical (row_ind, col_ind).  `matrix`: a MappedIndicator or a device tensor."""
    from .engine import default_engine
    eng = default_engine()
    dev = matrix.device_tensor() if hasattr(matrix, "device_tensor") else matrix
    if dev.dim() == 2:
        dev = dev[None]
    col = eng.linear_sum_assignment(dev, maximize=True)[0].cpu().numpy().astype(np.int64)
    rows = np.nonzero(col >= 0)[0]
    return rows, col[rows]


def _assign_many(matrices):
    """the assignments of several (n2, n1) matrices in ONE batched call: one workgroup per matrix, so three assignments take
    the time of the longest instead of their sum"""
    import torch
    from .engine import default_engine
    eng = default_engine()
    devs = [m.device_tensor() if hasattr(m, "device_tensor") else m for m in matrices]
    devs = [d if d.dim() == 2 else d[0] for d in devs]
    col = eng.linear_sum_assignment(torch.stack(devs), maximize=True).cpu().numpy().astype(np.int64)
    out = []
    for c in col:
        rows = np.nonzero(c >= 0)[0]
        out.append((rows, c[rows]))
    return out


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

    timing = os.environ.get("TIMEIT", False)                                    # functional_map.py:52-54
    if timing:
        compute_extra = True

    # The reference interleaves its three Hungarian calls with the precise map and ICP (functional_map.py:57-78).  None of the
    # three depends on another one's result, so here the matrices are formed first and the assignments run as ONE batched
    # call (a matrix's shortest-augmenting-path search is sequential: one workgroup each, side by side on the GPU).
    start_s = time.time()
    ind_plain = model.mapped_indicator if compute_extra else None               # functional_map.py:57
    precise = None
    if compute_extra:
        precise = model._precise_map_device()                                   # functional_map.py:62 (get_precise_map().toarray())
        if timing:
            print("getting precise map took", time.time() - start_s, "seconds")

    start_s = time.time()
    model.icp_refine()                                                          # functional_map.py:71, nit=10
    if timing:
        print("ICP refinement took", time.time() - start_s, "seconds")
    start_s = time.time()
    p2p_21_icp_adjoint, p2p_12_icp_adjoint = model.get_p2p(n_jobs=1)
    p2p_21_icp = (model.mapped_indicator * model.eta[..., None]).argmax(axis=1)
    p2p_12_icp = (model.mapped_indicator * model.eta[..., None]).argmax(axis=0)
    if compute_extra:
        hungarian, hungarian_precise, hungarian_icp = _assign_many([ind_plain, precise, model.mapped_indicator])   # :57, :66, :78
    else:
        hungarian, hungarian_precise = None, None
        hungarian_icp = _assign_many([model.mapped_indicator])[0]               # functional_map.py:78
    if timing:
        print("Hungarian for vanilla, precise and icp took", time.time() - start_s, "seconds")
    return (p2p_21, p2p_12, hungarian, hungarian_precise, p2p_21_icp, p2p_12_icp, hungarian_icp, model, model.mesh1, model.mesh2,
            p2p_21_adjoint, p2p_12_adjoint, p2p_21_icp_adjoint, p2p_12_icp_adjoint)

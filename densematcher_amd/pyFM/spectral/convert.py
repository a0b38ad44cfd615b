"""
FM <-> p2p conversion with the reference's signatures (densematcher/pyFM/spectral/convert.py).
"""
import numpy as np
import scipy.sparse as sparse


def _diag_of(A, n):
    """lumped mass: accept a sparse diagonal matrix, a dense square matrix or the diagonal itself"""
    if A is None:
        return None
    if sparse.issparse(A):
        d = A.diagonal()
    else:
        A = np.asarray(A)
        d = A if A.ndim == 1 else np.diag(A)
    if d.shape[0] != n:
        raise ValueError("Can't compute exact pseudo inverse with subsampled eigenvectors")
    return np.ascontiguousarray(d, dtype=_real_dtype(d))


def _real_dtype(*arrays):
    """float64 inputs stay float64 (the reference's dtype; the *_f64 entry points of the library), anything else is fp32"""
    return np.float64 if any(np.asarray(a).dtype == np.float64 for a in arrays if a is not None) else np.float32


def _basis(evects, k, dtype):
    return np.ascontiguousarray(np.asarray(evects)[:, :k], dtype=dtype)[None]


class MappedIndicator:
    """The (n2, n1) matrix Phi2 C Phi1^T A1 of the reference (convert.py:144), kept implicit.

    `argmax(axis=...)` comes from the fused G-tile kernel (the matrix is never formed); converting to an array
    (np.asarray, arithmetic, indexing) materialises it once on the GPU with dm_mapped_indicator."""
    def __init__(self, eng, Phi1, Phi2, a1, C, ind21, ind12):
        self._eng, self._args = eng, (Phi1, Phi2, a1, C)
        self._ind21, self._ind12 = ind21, ind12
        self._dense = None
        self.shape = (Phi2.shape[1], Phi1.shape[1])
        self.ndim = 2
        self.dtype = np.dtype(np.float64)

    def argmax(self, axis=None):
        if axis in (1, -1):
            return self._ind21
        if axis == 0:
            return self._ind12
        return np.asarray(self).argmax(axis)

    def device_tensor(self):
        """the (n2, n1) matrix on the GPU (formed once by dm_mapped_indicator): input of the assignment kernel"""
        if getattr(self, "_dev", None) is None:
            # (the engine of the CALLER's stream: the indicator may have been set up by a chunk thread of compute_surface_map_batch on
            #  its own stream, whose work is complete by the time anybody holds this object)
            from ...engine import default_engine
            self._dev = default_engine().mapped_indicator(*self._args)[0]
        return self._dev

    def __array__(self, dtype=None, copy=None):
        if self._dense is None:
            self._dense = self.device_tensor().cpu().numpy()
        return self._dense if dtype is None else self._dense.astype(dtype)

    def __mul__(self, other):
        other = np.asarray(other)
        if other.shape in ((self.shape[0], 1), (1, 1), ()) and np.all(other == 1):
            return self                  # functional_map.py:49: `mapped_indicator * eta[..., None]`, eta == 1 after fit
        return np.asarray(self) * other

    __rmul__ = __mul__

    def __getitem__(self, idx):
        return np.asarray(self)[idx]


def FM_to_p2p(FM_12, evects1, evects2, A1, use_adj=False, n_jobs=1):
    """(p2p_21, p2p_12, mapped_indicator) -- reference convert.py:96-147 (`use_adj`, `n_jobs` accepted, ignored
    there too).  Eigenvectors are sliced to the map's size (the fork's unsliced :144 only works when they are)."""
    from ...engine import default_engine
    FM_12 = np.asarray(FM_12, dtype=np.float64)
    k2, k1 = FM_12.shape
    assert k1 <= evects1.shape[1], f'At least {k1} should be provided, here only {evects1.shape[1]} are given'
    assert k2 <= evects2.shape[1], f'At least {k2} should be provided, here only {evects2.shape[1]} are given'
    eng = default_engine()
    a1 = _diag_of(A1, evects1.shape[0])
    # float64 eigenvectors (what TriMesh holds, like the reference) run the float64-basis kernels: the maps are those of
    # convert.py:134-144 on the same numbers; an fp32 basis takes the fp32 entry points
    dt = _real_dtype(evects1, evects2)
    _, Phi1, Phi2 = eng._reals(_basis(evects1, k1, dt), _basis(evects2, k2, dt))
    if a1 is not None:
        a1 = a1.astype(dt)
    out = eng.fm_to_p2p(Phi1, Phi2, None if a1 is None else a1[None], FM_12[None], knn=True, ind=a1 is not None)
    p2p_21 = out["knn21"][0].cpu().numpy().astype(np.int64)
    p2p_12 = out["knn12"][0].cpu().numpy().astype(np.int64)
    indicator = None
    if a1 is not None:
        indicator = MappedIndicator(eng, Phi1, Phi2, a1[None], FM_12[None], out["ind21"][0].cpu().numpy().astype(np.int64),
                                    out["ind12"][0].cpu().numpy().astype(np.int64))
    return p2p_21, p2p_12, indicator


def mesh_FM_to_p2p(FM_12, mesh1, mesh2, use_adj=False, subsample=None, n_jobs=1):
    """reference convert.py:149-182"""
    k2, k1 = FM_12.shape
    if subsample is None:
        return FM_to_p2p(FM_12, mesh1.eigenvectors[:, :k1], mesh2.eigenvectors[:, :k2], mesh1.A, use_adj=use_adj, n_jobs=n_jobs)
    sub1, sub2 = subsample
    return FM_to_p2p(FM_12, mesh1.eigenvectors[sub1, :k1], mesh2.eigenvectors[sub2, :k2], None, use_adj=use_adj, n_jobs=n_jobs)


def mesh_FM_to_p2p_precise(FM_12, mesh1, mesh2, precompute_dmin=True, use_adj=True, batch_size=None, n_jobs=1, verbose=False):
    """reference convert.py:185-229: (n2, n1) sparse matrix with the barycentric coordinates of the image of every vertex
    of mesh 2 on mesh 1.  GPU: dm_precise_map (use_adj = True, the form get_precise_map uses; the memory / batching
    switches of the reference have no counterpart)."""
    from ...engine import default_engine
    if not use_adj:
        raise NotImplementedError("only the adjoint form (use_adj=True, the default of get_precise_map) is on the GPU path")
    FM_12 = np.asarray(FM_12, dtype=np.float64)
    k2, k1 = FM_12.shape
    faces = np.ascontiguousarray(mesh1.facelist, dtype=np.int32)
    dt = _real_dtype(mesh1.eigenvectors, mesh2.eigenvectors)
    fm, bary = default_engine().precise_map(_basis(mesh1.eigenvectors, k1, dt), _basis(mesh2.eigenvectors, k2, dt), FM_12[None], faces[None])
    fm, bary = fm[0].cpu().numpy().astype(np.int64), bary[0].cpu().numpy()
    n2, n1 = mesh2.eigenvectors.shape[0], mesh1.eigenvectors.shape[0]
    rows = np.tile(np.arange(n2), 3)                                         # projection_utils.py:360-399
    cols = np.concatenate([faces[fm, 0], faces[fm, 1], faces[fm, 2]])
    return sparse.csr_matrix((np.concatenate([bary[:, 0], bary[:, 1], bary[:, 2]]), (rows, cols)), shape=(n2, n1))


def p2p_to_FM(p2p_21, evects1, evects2, A2=None):
    """reference convert.py:14-51.  With A2: Phi2^T (A2 Phi1[p2p_21]) (dm_p2p_to_fm).  Without A2 the reference solves the
    least-squares problem lstsq(Phi2, Phi1[p2p_21]) (:51): normal equations on the GPU (dm_p2p_to_fm_lstsq).
    p2p_21 may also be a sparse (n2, n1) map P (:39): the pulled-back basis P Phi1 is then formed on the host (a sparse
    product, as in the reference) and enters the same kernels with the identity as index map."""
    from ...engine import default_engine
    eng = default_engine()
    evects1 = np.asarray(evects1)
    evects2 = np.asarray(evects2)
    k1, k2 = evects1.shape[1], evects2.shape[1]
    if sparse.issparse(p2p_21) or np.asarray(p2p_21).ndim != 1:
        pulled = np.asarray(p2p_21 @ evects1)                               # (n2, k1)
        idx = np.arange(pulled.shape[0], dtype=np.int32)
    else:
        pulled, idx = evects1, np.ascontiguousarray(p2p_21, dtype=np.int32)
    dt = _real_dtype(pulled, evects2)
    P1 = np.ascontiguousarray(pulled, dtype=dt)[None]
    P2 = np.ascontiguousarray(evects2, dtype=dt)[None]
    if A2 is None:
        return eng.p2p_to_fm_lstsq(idx[None], P1, P2, k1, k2)[0].cpu().numpy()
    a2 = _diag_of(A2, evects2.shape[0]).astype(dt)
    return eng.p2p_to_fm(idx[None], P1, P2, a2[None], k1, k2)[0].cpu().numpy()


def mesh_p2p_to_FM(p2p_21, mesh1, mesh2, dims=None, subsample=None):
    """reference convert.py:54-93"""
    if dims is None:
        k1, k2 = len(mesh1.eigenvalues), len(mesh2.eigenvalues)
    elif np.issubdtype(type(dims), np.integer):
        k1 = k2 = dims
    else:
        k1, k2 = dims
    if subsample is None:
        return p2p_to_FM(p2p_21, mesh1.eigenvectors[:, :k1], mesh2.eigenvectors[:, :k2], A2=mesh2.A)
    sub1, sub2 = subsample
    return p2p_to_FM(p2p_21, mesh1.eigenvectors[sub1, :k1], mesh2.eigenvectors[sub2, :k2], A2=None)

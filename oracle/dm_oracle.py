"""
CPU oracle for the DenseMatcher matching hot path.

TEST INFRASTRUCTURE ONLY.  This module is a NumPy float64 restatement of the
reference's algorithm.  It may be imported by `tests/`, by
`__graft_entry__.smoke()` and by the `cpu_baseline` leg of `bench.py` as the
*checker* / the *timed CPU baseline*.  It is never imported by the product
package `densematcher_amd` (which fails loudly when its HIP library is missing).

Parity pinning: the reference has no tests or golden vectors on this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, generated in the build container by `tools/make_golden.py` (which
imports `/root/reference`) and committed under `tests/golden/`.
`tests/test_oracle_golden.py` checks every function below against them.

Every function cites the reference file:line it restates (paths relative to
`/root/reference/densematcher/`).

Conventions:
    Phi1 (N1,k1), Phi2 (N2,k2)   mass-orthonormal Laplace-Beltrami eigenvectors
    lam1 (k1,),  lam2 (k2,)      eigenvalues
    a1 (N1,),    a2 (N2,)        diagonal of the lumped mass matrices A1, A2
    F1 (N1,D),   F2 (N2,D)       per-vertex descriptors
    C (k2,k1)                    functional map: basis-1 coefficients -> basis-2
    p2p_21 (N2,)                 for each vertex of mesh 2, a vertex of mesh 1
"""
import numpy as np
import scipy.linalg
import scipy.optimize


# --------------------------------------------------------------------------- #
# spectral projection
# --------------------------------------------------------------------------- #
def project(phi, mass, F):
    """Phi^T (a * F)   -- pyFM/optimize/base_functions.py:526-532
    (`evects.T @ A @ descr`, A = dense diag(mass); eta == 1) and
    pyFM/mesh/trimesh.py:533-556 (`TriMesh.project`)."""
    phi = np.asarray(phi, dtype=np.float64)
    F = np.asarray(F, dtype=np.float64)
    mass = np.asarray(mass, dtype=np.float64)
    return phi.T @ (mass[:, None] * F)


def ev_sqdiff(lam1, lam2):
    """((lam1[None,:] - lam2[:,None]) / scale)^2, scale = max(lam1.max, lam2.max)
    -- pyFM/functional.py:404-405.  Note: the reference divides each term by
    `scale` before subtracting; kept in that order."""
    lam1 = np.asarray(lam1, dtype=np.float64)
    lam2 = np.asarray(lam2, dtype=np.float64)
    scale = max(lam1.max(), lam2.max())
    return np.square(lam1[None, :] / scale - lam2[:, None] / scale)


def get_x0(k1, k2, phi1_00, phi2_00, area1, area2, optinit="zeros", rng=None):
    """Initial functional map with the hand-set first column
    -- pyFM/functional.py:629-660."""
    if optinit == "random":
        rng = np.random if rng is None else rng
        x0 = rng.random((k2, k1))
        x0 = x0 / x0.sum()
    elif optinit == "identity":
        x0 = np.eye(k2, k1)
    elif optinit == "zeros":
        x0 = np.zeros((k2, k1))
    else:
        raise ValueError(f"optinit arg should be 'random', 'identity' or 'zeros', not {optinit}")
    ev_sign = np.sign(phi1_00 * phi2_00)
    area_ratio = np.sqrt(area2 / area1)
    x0[:, 0] = np.zeros(k2)
    x0[0, 0] = ev_sign * area_ratio
    return x0


# --------------------------------------------------------------------------- #
# energy, gradient, minimiser
# --------------------------------------------------------------------------- #
def energy(C, A, B, ev, w_descr, w_lap):
    """w_descr * 0.5 ||C A - B||^2 + w_lap * 0.5 sum(C^2 * ev)
    -- pyFM/optimize/base_functions.py:49 (descr_preservation), :95
    (LB_commutation), :534-544 (weights)."""
    return w_descr * 0.5 * np.square(C @ A - B).sum() + w_lap * 0.5 * (np.square(C) * ev).sum()


def grad_energy(C, A, B, ev, w_descr, w_lap):
    """w_descr (C A - B) A^T + w_lap C * ev with column 0 zeroed
    -- pyFM/optimize/base_functions.py:58-76, :105-121, :759."""
    g = w_descr * (C @ A - B) @ A.T + w_lap * C * ev
    g[:, 0] = 0
    return g


def fmap_fit_lbfgs(A, B, ev, x0, w_descr, w_lap, maxiter=100000):
    """float64 L-BFGS-B on the reference energy with its analytic gradients
    -- pyFM/functional.py:477 (scipy.optimize.minimize) driven in float64
    instead of the fp32 torch autograd energy.  Tight tolerances so that it
    converges to the minimiser that `fmap_solve` gives in closed form."""
    k2, k1 = x0.shape

    def f(x):
        return energy(x.reshape(k2, k1), A, B, ev, w_descr, w_lap)

    def g(x):
        return grad_energy(x.reshape(k2, k1), A, B, ev, w_descr, w_lap).ravel()

    res = scipy.optimize.minimize(f, x0.ravel(), jac=g, method="L-BFGS-B",
                                  options={"maxiter": maxiter, "maxfun": 10 * maxiter,
                                           "ftol": 1e-20, "gtol": 1e-12, "maxcor": 30})
    return res.x.reshape(k2, k1), res


def fmap_solve(A, B, lam1, lam2, x0, w_descr, w_lap):
    """Closed-form minimiser of `energy` with column 0 pinned to x0[:,0]
    (SURVEY.md Appendix A.5).  Rows decouple:
        (P[f,f] + w_lap diag(ev[i,f])) C[i,f] = Q[i,f] - P[f,0] x0[i,0],
    P = w_descr A A^T, Q = w_descr B A^T, f = 1..k1-1.
    This is what any optimiser applied to base_functions.py:480-763 converges
    to when only w_descr and w_lap are non-zero."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    k1, k2 = A.shape[0], B.shape[0]
    ev = ev_sqdiff(lam1, lam2)
    P = w_descr * (A @ A.T)
    Q = w_descr * (B @ A.T)
    C = np.zeros((k2, k1))
    C[:, 0] = x0[:, 0]
    Pff = P[1:, 1:]
    for i in range(k2):
        M = Pff + w_lap * np.diag(ev[i, 1:])
        rhs = Q[i, 1:] - P[1:, 0] * x0[i, 0]
        C[i, 1:] = scipy.linalg.solve(M, rhs, assume_a="pos")
    return C


def fit(phi1, phi2, lam1, lam2, a1, a2, F1, F2, w_descr, w_lap, optinit="zeros"):
    """FunctionalMapping.fit restated (w_descr / w_lap terms only)
    -- pyFM/functional.py:352-487.  Returns C (k2,k1) float64."""
    k1, k2 = phi1.shape[1], phi2.shape[1]
    A = project(phi1, a1, F1)
    B = project(phi2, a2, F2)
    x0 = get_x0(k1, k2, float(phi1[0, 0]), float(phi2[0, 0]),
                float(np.asarray(a1, dtype=np.float64).sum()),
                float(np.asarray(a2, dtype=np.float64).sum()), optinit)
    return fmap_solve(A, B, lam1, lam2, x0, w_descr, w_lap)


# --------------------------------------------------------------------------- #
# functional map <-> point-to-point map
# --------------------------------------------------------------------------- #
def knn_query(X, Y, chunk=2048):
    """For every row of Y the index of the nearest row of X (k = 1)
    -- pyFM/spectral/nn_utils.py:4-38 (sklearn kd-tree, exact Euclidean NN).
    Restated as the brute-force argmin of |x|^2 - 2 <x,y>, lowest index on
    ties.  Agrees with the kd-tree wherever the nearest neighbour is unique."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    nx = np.einsum("ij,ij->i", X, X)
    out = np.empty(Y.shape[0], dtype=np.int64)
    for s in range(0, Y.shape[0], chunk):
        d = nx[None, :] - 2.0 * (Y[s:s + chunk] @ X.T)
        out[s:s + chunk] = d.argmin(axis=1)
    return out


def fm_to_p2p(C, phi1, phi2, a1, with_indicator=True):
    """FM_to_p2p -- pyFM/spectral/convert.py:96-147.
    Returns (p2p_21 (N2,), p2p_12 (N1,), mapped_indicator (N2,N1) or None).
    Eigenvectors are sliced to the map's size (the fork's unsliced :144 only
    works when they already are)."""
    k2, k1 = C.shape
    assert k1 <= phi1.shape[1], f"At least {k1} should be provided, here only {phi1.shape[1]} are given"
    assert k2 <= phi2.shape[1], f"At least {k2} should be provided, here only {phi2.shape[1]} are given"
    e1 = np.asarray(phi1[:, :k1], dtype=np.float64)
    e2 = np.asarray(phi2[:, :k2], dtype=np.float64)
    # convert.py:134-136  tree = Phi2 C, query = Phi1
    p2p_12 = knn_query(e2 @ C, e1)
    # convert.py:138-140  tree = Phi1 C^T, query = Phi2
    p2p_21 = knn_query(e1 @ C.T, e2)
    ind = None
    if with_indicator:
        # convert.py:144  Phi2 C Phi1^T A1, A1 diagonal
        ind = ((e2 @ C) @ e1.T) * np.asarray(a1, dtype=np.float64)[None, :]
    return p2p_21, p2p_12, ind


def indicator_argmax(ind, eta=None):
    """functional_map.py:49-50: argmax over axis 1 and axis 0 of
    mapped_indicator * eta[:,None] (eta == 1 after fit, functional.py:483)."""
    if eta is not None:
        ind = ind * eta[..., None]
    return ind.argmax(axis=1), ind.argmax(axis=0)


def fm_to_p2p_all(C, phi1, phi2, a1, chunk=1024):
    """All four integer maps of one pair without materialising the indicator
    at once (same arithmetic as fm_to_p2p + indicator_argmax, row-chunked).
    Returns (knn21, knn12, ind21, ind12)."""
    k2, k1 = C.shape
    e1 = np.asarray(phi1[:, :k1], dtype=np.float64)
    e2 = np.asarray(phi2[:, :k2], dtype=np.float64)
    a1 = np.asarray(a1, dtype=np.float64)
    knn12 = knn_query(e2 @ C, e1)
    knn21 = knn_query(e1 @ C.T, e2)
    emb2 = e2 @ C
    N2, N1 = e2.shape[0], e1.shape[0]
    ind21 = np.empty(N2, dtype=np.int64)
    colbest = np.full(N1, -np.inf)
    ind12 = np.zeros(N1, dtype=np.int64)
    for s in range(0, N2, chunk):
        blk = (emb2[s:s + chunk] @ e1.T) * a1[None, :]
        ind21[s:s + chunk] = blk.argmax(axis=1)
        r = blk.argmax(axis=0)
        v = blk[r, np.arange(N1)]
        upd = v > colbest          # strict: earlier chunk wins ties (first index)
        colbest[upd] = v[upd]
        ind12[upd] = r[upd] + s
    return knn21, knn12, ind21, ind12


def p2p_to_fm(p2p_21, phi1, phi2, a2=None):
    """p2p_to_FM -- pyFM/spectral/convert.py:14-51.
    With a2: Phi2^T (a2 * Phi1[p2p_21])  (:41-48); without: least squares (:51)."""
    e1 = np.asarray(phi1, dtype=np.float64)
    e2 = np.asarray(phi2, dtype=np.float64)
    pb = e1[np.asarray(p2p_21), :]
    if a2 is not None:
        a2 = np.asarray(a2, dtype=np.float64)
        if a2.shape[0] != e2.shape[0]:
            raise ValueError("Can't compute exact pseudo inverse with subsampled eigenvectors")
        return e2.T @ (a2[:, None] * pb)
    return scipy.linalg.lstsq(e2, pb)[0]


# --------------------------------------------------------------------------- #
# refinement
# --------------------------------------------------------------------------- #
def zoomout_refine(C, phi1, phi2, nit, step=1, a2=None, return_p2p=False, trajectory=None, subsample=None):
    """ZoomOut -- pyFM/refine/zoomout.py:7-44 (iteration), :47-115 (loop),
    with upstream-pyFM FM_to_p2p semantics (the fork's call at :40/:112 is
    broken): p2p_21 = NN(tree = Phi1[:, :k1] C^T,
    query = Phi2[:, :k2])."""
    k2_0, k1_0 = C.shape
    try:
        step1, step2 = step
    except TypeError:
        step1 = step2 = step
    assert k1_0 + nit * step1 <= phi1.shape[1], \
        f"Not enough eigenvectors on source : {k1_0 + nit * step1} are needed when {phi1.shape[1]} are provided"
    assert k2_0 + nit * step2 <= phi2.shape[1], \
        f"Not enough eigenvectors on target : {k2_0 + nit * step2} are needed when {phi2.shape[1]} are provided"
    e1 = np.asarray(phi1, dtype=np.float64)
    e2 = np.asarray(phi2, dtype=np.float64)
    C = np.array(C, dtype=np.float64)
    s1, s2, a2_it = e1, e2, a2
    if subsample is not None:                                    # zoomout.py:94-102: subsampled vertices, least squares
        s1, s2, a2_it = e1[subsample[0]], e2[subsample[1]], None
    for _ in range(nit):
        k2, k1 = C.shape
        p21 = knn_query(s1[:, :k1] @ C.T, s2[:, :k2])           # zoomout.py:40
        C = p2p_to_fm(p21, s1[:, :k1 + step1], s2[:, :k2 + step2], a2_it)   # zoomout.py:42
        if trajectory is not None:
            trajectory.append((p21, C))
    if return_p2p:
        k2, k1 = C.shape
        p21 = knn_query(e1[:, :k1] @ C.T, e2[:, :k2])           # zoomout.py:111-113
        return C, p21
    return C


def icp_refine(C, phi1, phi2, nit=10):
    """ICP -- pyFM/refine/icp.py:10-40 (iteration), :43-107 (loop, fixed nit).
    p2p_21 -> least-squares map (no mass, convert.py:51) -> U eye V^T."""
    k2, k1 = C.shape
    e1 = np.asarray(phi1[:, :k1], dtype=np.float64)
    e2 = np.asarray(phi2[:, :k2], dtype=np.float64)
    C = np.array(C, dtype=np.float64)
    for _ in range(nit):
        p21 = knn_query(e1 @ C.T, e2)
        Ch = scipy.linalg.lstsq(e2, e1[p21, :])[0]
        U, _, VT = scipy.linalg.svd(Ch)
        C = U @ np.eye(k2, k1) @ VT
    return C


# --------------------------------------------------------------------------- #
# feature-similarity nearest neighbour (BASELINE.json config 3)
# --------------------------------------------------------------------------- #
def simnn(F_tgt, F_src, chunk=2048):
    """nn[i] = argmax_j <F_tgt[i], F_src[j]>, lowest index on ties.
    No reference symbol exists for this: parity is
    unpinned by the reference; this float64 statement is the definition."""
    T = np.asarray(F_tgt, dtype=np.float64)
    S = np.asarray(F_src, dtype=np.float64)
    out = np.empty(T.shape[0], dtype=np.int64)
    for s in range(0, T.shape[0], chunk):
        out[s:s + chunk] = (T[s:s + chunk] @ S.T).argmax(axis=1)
    return out


# --------------------------------------------------------------------------- #
# one pair end to end (what bench.py's cpu_baseline times)
# --------------------------------------------------------------------------- #
def match_pair(phi1, phi2, lam1, lam2, a1, a2, F1, F2, w_descr=1e4, w_lap=1e3):
    """project -> closed-form fit -> four integer maps, for one pair."""
    C = fit(phi1, phi2, lam1, lam2, a1, a2, F1, F2, w_descr, w_lap)
    knn21, knn12, ind21, ind12 = fm_to_p2p_all(C, phi1, phi2, a1)
    return C, knn21, knn12, ind21, ind12


def knn_query_kdtree(X, Y):
    """knn_query exactly as the reference runs it: sklearn NearestNeighbors(kd_tree, leaf_size=40), k = 1
    -- pyFM/spectral/nn_utils.py:28-30.  Falls back to the brute-force restatement when scikit-learn is absent."""
    try:
        from sklearn.neighbors import NearestNeighbors
    except ImportError:
        return knn_query(X, Y)
    tree = NearestNeighbors(n_neighbors=1, leaf_size=40, algorithm="kd_tree", n_jobs=1)
    tree.fit(np.asarray(X, dtype=np.float64))
    return tree.kneighbors(np.asarray(Y, dtype=np.float64), return_distance=False).squeeze()


def match_pair_reference_faithful(phi1, phi2, lam1, lam2, a1, a2, F1, F2, w_descr=1e4, w_lap=1e3, maxiter=100000):
    """One pair with the reference's own algorithm choices (the timed "reference-faithful" CPU baseline of bench.py,
    BASELINE.md section 3): projections (base_functions.py:526-532), ITERATIVE L-BFGS-B on the energy with SciPy's
    default tolerances as in pyFM/functional.py:477 (float64 energy / analytic gradient instead of the fp32 torch
    autograd), kd-tree nearest neighbours in both directions (convert.py:134-140), the dense N2 x N1 mapped indicator
    (convert.py:144) and its two arg-maxes (functional_map.py:49-50)."""
    k1, k2 = phi1.shape[1], phi2.shape[1]
    A = project(phi1, a1, F1)
    B = project(phi2, a2, F2)
    ev = ev_sqdiff(lam1, lam2)
    x0 = get_x0(k1, k2, float(phi1[0, 0]), float(phi2[0, 0]), float(np.asarray(a1, dtype=np.float64).sum()),
                float(np.asarray(a2, dtype=np.float64).sum()), "zeros")
    res = scipy.optimize.minimize(lambda x: energy(x.reshape(k2, k1), A, B, ev, w_descr, w_lap), x0.ravel(),
                                  jac=lambda x: grad_energy(x.reshape(k2, k1), A, B, ev, w_descr, w_lap).ravel(),
                                  method="L-BFGS-B", options={"maxiter": maxiter})
    C = res.x.reshape(k2, k1)
    e1 = np.asarray(phi1, dtype=np.float64)
    e2 = np.asarray(phi2, dtype=np.float64)
    knn12 = knn_query_kdtree(e2 @ C, e1)
    knn21 = knn_query_kdtree(e1 @ C.T, e2)
    ind = ((e2 @ C) @ e1.T) * np.asarray(a1, dtype=np.float64)[None, :]
    ind21, ind12 = indicator_argmax(ind)
    return C, knn21, knn12, ind21, ind12


# --------------------------------------------------------------------------- #
# the other energy terms of FunctionalMapping.fit
# --------------------------------------------------------------------------- #
M_TERMS = ("w_p2p", "w_stochastic", "w_ent", "w_range01", "w_sumto1")


def mapped_indicator(C, phi1, phi2, a1):
    """evects2 @ C @ evects1.T @ A1 (N2 x N1) -- base_functions.py:315, 348, 364, 375, 408 (first line of each term)."""
    return (np.asarray(phi2, np.float64) @ C) @ (np.asarray(phi1, np.float64) * np.asarray(a1, np.float64)[:, None]).T


def m_terms_energy_grad(C, phi1, phi2, a1, w_p2p=0.0, w_stochastic=0.0, w_ent=0.0, w_range01=0.0, w_sumto1=0.0):
    """Sum of the weighted energy terms that are functions of the mapped indicator M = Phi2 C Phi1^T A1, and its
    gradient with respect to C (what torch autograd returns in the reference: dE/dC = Phi2^T (dE/dM) (A1 Phi1)):
        p2p               sum (M^2 - M)^2                                        base_functions.py:296-322
        doubly_stochastic sum_j (sum_i M_ij^2 - n2/n1)^2 + sum_i (sum_j M_ij^2 - 1)^2     :324-361
        entropy           sum -c log(c + 1e-10), c = clamp(M, 0, 1)              :363-372
        range01           sum relu(-M)^2 + relu(M - 1)^2                         :374-385
        sumto1 (v = None) sum_j (cs_j - mean cs)^2 + sum_i (rs_i - mean rs)^2, cs / rs = column / row sums   :387-428
    Returns (energy, grad (k2,k1))."""
    phi1 = np.asarray(phi1, np.float64)
    phi2 = np.asarray(phi2, np.float64)
    a1 = np.asarray(a1, np.float64)
    M = mapped_indicator(C, phi1, phi2, a1)
    n2, n1 = M.shape
    E = 0.0
    D = np.zeros_like(M)
    if w_p2p > 0:
        q = M * M - M
        E += w_p2p * np.square(q).sum()
        D += w_p2p * 2.0 * q * (2.0 * M - 1.0)
    if w_stochastic > 0:
        Msq = M * M
        dc = Msq.sum(axis=0) - n2 / n1
        dr = Msq.sum(axis=1) - 1.0
        E += w_stochastic * (np.square(dc).sum() + np.square(dr).sum())
        D += w_stochastic * (2.0 * dc[None, :] + 2.0 * dr[:, None]) * 2.0 * M
    if w_ent > 0:
        c = np.clip(M, 0.0, 1.0)
        E += w_ent * np.sum(-c * np.log(c + 1e-10))
        inside = (M >= 0.0) & (M <= 1.0)                 # torch.clamp passes the gradient on the closed interval
        D += w_ent * np.where(inside, -np.log(c + 1e-10) - c / (c + 1e-10), 0.0)
    if w_range01 > 0:
        lo = np.maximum(-M, 0.0)
        hi = np.maximum(M - 1.0, 0.0)
        E += w_range01 * (np.square(lo).sum() + np.square(hi).sum())
        D += w_range01 * (-2.0 * lo + 2.0 * hi)
    if w_sumto1 > 0:
        cs, rs = M.sum(axis=0), M.sum(axis=1)
        dc, dr = cs - cs.mean(), rs - rs.mean()
        E += w_sumto1 * (np.square(dc).sum() + np.square(dr).sum())
        D += w_sumto1 * (2.0 * dc[None, :] + 2.0 * dr[:, None])      # (the deviations sum to zero: the mean's derivative drops out)
    G = phi2.T @ (D @ (phi1 * a1[:, None]))
    return float(E), G


def descr_ops(phi, mass, F):
    """Multiplication operators of the descriptors in the reduced basis, (D, k, k): Phi^T A diag(f_i) Phi
    -- base_functions.py:550-555 (commute_left / commute_right; pinv = Phi^T A, functional.py:416-417)."""
    phi = np.asarray(phi, np.float64)
    F = np.asarray(F, np.float64)
    pinv = phi.T * np.asarray(mass, np.float64)[None, :]
    return np.stack([pinv @ (F[:, i, None] * phi) for i in range(F.shape[1])])


def dcomm_energy_grad(C, ops1, ops2):
    """sum_i 0.5 |C L_i - R_i C|^2 and its gradient -- base_functions.py:124-226 (op_commutation / oplist_commutation)."""
    E = 0.0
    G = np.zeros_like(C)
    for L, R in zip(ops1, ops2):
        T = C @ L - R @ C
        E += 0.5 * np.square(T).sum()
        G += T @ L.T - R.T @ T
    return float(E), G


def area_energy_grad(C):
    """1/2 |C^T C - I|^2 and 2 C C^T C - 2 C -- base_functions.py:228-255 (area, area_grad)."""
    M = C.T @ C - np.eye(C.shape[1])
    return float(0.5 * np.square(M).sum()), 2.0 * C @ M


def conformal_energy_grad(C, lam1, lam2):
    """1/2 |C^T D2 C - D1|^2, D = diag(lam / max(lam1.max, lam2.max)), and 2 D2 C C^T D2 C - 2 D2 C D1
    -- base_functions.py:257-294 (conformal, conformal_grad)."""
    lam1, lam2 = np.asarray(lam1, np.float64), np.asarray(lam2, np.float64)
    scale = max(lam1.max(), lam2.max())
    d1, d2 = lam1 / scale, lam2 / scale
    M = C.T @ (d2[:, None] * C) - np.diag(d1)
    return float(0.5 * np.square(M).sum()), 2.0 * (d2[:, None] * C) @ M


def face_normals(verts, faces):
    """unit face normals -- pyFM/mesh/geometry.py:110-133 (compute_normals)."""
    v1, v2, v3 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    n = np.cross(v2 - v1, v3 - v1)
    return n / np.linalg.norm(n, axis=1, keepdims=True)


def face_gradients(verts, faces, F):
    """per-face gradients of the vertex functions F (n, p) by linear interpolation, (m, 3, p)
    -- pyFM/mesh/geometry.py:284-316 (_get_grad_dir), :319-370 (grad_mat), :373-430 (grad_f)."""
    verts = np.asarray(verts, np.float64)
    F = np.asarray(F, np.float64)
    v1, v2, v3 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    nrm = face_normals(verts, faces)
    fa = 0.5 * np.linalg.norm(np.cross(v2 - v1, v3 - v1), axis=1)
    g1 = np.cross(nrm, v3 - v2) / (2 * fa[:, None])
    g2 = np.cross(nrm, v1 - v3) / (2 * fa[:, None])
    g3 = np.cross(nrm, v2 - v1) / (2 * fa[:, None])
    return (g1[:, :, None] * F[faces[:, 0]][:, None, :] + g2[:, :, None] * F[faces[:, 1]][:, None, :]
            + g3[:, :, None] * F[faces[:, 2]][:, None, :])


def orientation_ops(phi, mass, verts, faces, F, vertex_areas=None):
    """The orientation operators of the descriptors in the reduced basis, (p, k, k): pinv diag(1 / area) W_i Phi with W_i the
    sparse operator g -> <n x grad f_i, grad g> summed over the faces around a vertex
    -- pyFM/mesh/geometry.py:919-985 (get_orientation_op), pyFM/functional.py:686-728 (compute_orientation_op: area = the mesh's
    vertex_areas, the row sums of A) and base_functions.py:430-478, :567-597 (orientation_op_torch inside energy_func_std: area =
    diag(A), which cancels against pinv = Phi^T A).  vertex_areas None = the second form."""
    phi = np.asarray(phi, np.float64)
    verts = np.asarray(verts, np.float64)
    mass = np.asarray(mass, np.float64)
    nrm = face_normals(verts, faces)
    v1, v2, v3 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    Jc1, Jc2, Jc3 = np.cross(nrm, v3 - v2) / 2, np.cross(nrm, v1 - v3) / 2, np.cross(nrm, v2 - v1) / 2
    rot = np.cross(nrm[:, :, None], face_gradients(verts, faces, F), axis=1)                    # (m, 3, p)
    dot = lambda J: np.einsum("fc,fcp->fp", J, rot)
    Sij = np.concatenate([dot(Jc2), dot(Jc3), dot(Jc1)]) / 3.0                                    # (3 m, p)
    Sji = np.concatenate([dot(Jc1), dot(Jc2), dot(Jc3)]) / 3.0
    I = np.concatenate([faces[:, 0], faces[:, 1], faces[:, 2]])
    J = np.concatenate([faces[:, 1], faces[:, 2], faces[:, 0]])
    left = phi * (mass / (mass if vertex_areas is None else np.asarray(vertex_areas, np.float64)))[:, None]   # rows of pinv^T / area
    k = phi.shape[1]
    # sum_e S_e outer(left[In_e], phi[Jn_e]) over In = [I, J, I, J], Jn = [J, I, I, J], S = [Sij, Sji, -Sij, -Sji]
    ops = (np.einsum("ep,ea,eb->pab", Sij, left[I], phi[J] - phi[I], optimize=True)
           + np.einsum("ep,ea,eb->pab", Sji, left[J], phi[I] - phi[J], optimize=True))
    return ops.reshape(F.shape[1], k, k)


def energy_grad_general(C, A, B, ev, phi1, phi2, a1, weights, ops1=None, ops2=None, lam1=None, lam2=None, orient_ops=None):
    """energy_func_std / grad_energy_std restated for the terms the GPU path implements (base_functions.py:480-763):
    w_descr, w_lap, w_dcomm, w_orient (orient_ops = (ops1, ops2)), w_area, w_conformal (lam1, lam2), w_p2p, w_stochastic, w_ent,
    w_range01, w_sumto1; the gradient's column 0 is zeroed (:759)."""
    w = dict(w_descr=0.0, w_lap=0.0, w_dcomm=0.0, w_p2p=0.0, w_stochastic=0.0, w_ent=0.0, w_range01=0.0, w_sumto1=0.0,
             w_orient=0.0, w_area=0.0, w_conformal=0.0)
    w.update(weights)
    E = energy(C, A, B, ev, w["w_descr"], w["w_lap"])
    G = w["w_descr"] * (C @ A - B) @ A.T + w["w_lap"] * C * ev
    if w["w_dcomm"] > 0:
        e, g = dcomm_energy_grad(C, ops1, ops2)
        E += w["w_dcomm"] * e
        G += w["w_dcomm"] * g
    if w["w_orient"] > 0:
        e, g = dcomm_energy_grad(C, orient_ops[0], orient_ops[1])
        E += w["w_orient"] * e
        G += w["w_orient"] * g
    if w["w_area"] > 0:
        e, g = area_energy_grad(C)
        E += w["w_area"] * e
        G += w["w_area"] * g
    if w["w_conformal"] > 0:
        e, g = conformal_energy_grad(C, lam1, lam2)
        E += w["w_conformal"] * e
        G += w["w_conformal"] * g
    if any(w[t] > 0 for t in M_TERMS):
        e, g = m_terms_energy_grad(C, phi1, phi2, a1, **{t: w[t] for t in M_TERMS})
        E += e
        G += g
    G[:, 0] = 0
    return E, G


def fit_general(phi1, phi2, lam1, lam2, a1, a2, F1, F2, weights, optinit="zeros", maxiter=100000, x0=None, tight=True):
    """FunctionalMapping.fit with any of the implemented terms switched on -- pyFM/functional.py:352-487: L-BFGS-B
    (scipy.optimize.minimize, :477) on energy_func_std / grad_energy_std, here in float64 with tight tolerances.
    Returns (C, scipy result)."""
    k1, k2 = phi1.shape[1], phi2.shape[1]
    A = project(phi1, a1, F1)
    B = project(phi2, a2, F2)
    ev = ev_sqdiff(lam1, lam2)
    if x0 is None:
        x0 = get_x0(k1, k2, float(phi1[0, 0]), float(phi2[0, 0]), float(np.asarray(a1, np.float64).sum()),
                    float(np.asarray(a2, np.float64).sum()), optinit)
    ops1 = ops2 = None
    if weights.get("w_dcomm", 0) > 0:
        ops1, ops2 = descr_ops(phi1, a1, F1), descr_ops(phi2, a2, F2)
    cache = {}

    def fg(x):
        key = x.tobytes()
        if key not in cache:
            cache.clear()
            cache[key] = energy_grad_general(x.reshape(k2, k1), A, B, ev, phi1, phi2, a1, weights, ops1, ops2)
        return cache[key]

    opts = {"maxiter": maxiter, "maxfun": 10 * maxiter}
    if tight:
        opts.update({"ftol": 1e-20, "gtol": 1e-10, "maxcor": 30})
    res = scipy.optimize.minimize(lambda x: fg(x)[0], x0.ravel(), jac=lambda x: fg(x)[1].ravel(), method="L-BFGS-B", options=opts)
    return res.x.reshape(k2, k1), res


# --------------------------------------------------------------------------- #
# linear assignment (functional_map.py:57,66,78: scipy.optimize.linear_sum_assignment(..., maximize=True))
# --------------------------------------------------------------------------- #
def linear_sum_assignment(cost, maximize=False):
    """Third-party arithmetic: scipy.optimize.linear_sum_assignment (SciPy is unpinned in the reference's setup.py; the
    build container has 1.15.3), i.e. the shortest-augmenting-path algorithm of D. F. Crouse, "On implementing 2D
    rectangular assignment algorithms", IEEE TAES 52(4), 2016, as implemented in scipy/optimize/rectangular_lsap:
    rows are augmented in order; each augmentation is a Dijkstra search over the columns in the order of a `remaining`
    list that starts reversed and is compacted by swap-removal; among columns of equal tentative cost the first in scan
    order wins unless an unassigned column ties, in which case the last unassigned one does; duals are updated after
    each search.  Restated here step for step (pure Python: small cases only) so that the GPU kernel, which follows
    the same steps, can be checked bit for bit; tests/test_oracle_golden.py checks this restatement against SciPy.
    Returns (row_ind, col_ind) like SciPy."""
    cost = np.array(cost, dtype=np.float64)
    if maximize:
        cost = -cost
    transposed = cost.shape[0] > cost.shape[1]
    if transposed:
        cost = cost.T.copy()
    nr, nc = cost.shape
    u, v = np.zeros(nr), np.zeros(nc)
    col4row, row4col = -np.ones(nr, dtype=np.int64), -np.ones(nc, dtype=np.int64)
    path = -np.ones(nc, dtype=np.int64)
    for cur in range(nr):
        spc = np.full(nc, np.inf)
        SR, SC = np.zeros(nr, dtype=bool), np.zeros(nc, dtype=bool)
        remaining = [nc - it - 1 for it in range(nc)]
        nrem, min_val, sink, i = nc, 0.0, -1, cur
        while sink == -1:
            index, lowest = -1, np.inf
            SR[i] = True
            for it in range(nrem):
                j = remaining[it]
                r = min_val + cost[i, j] - u[i] - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest, index = spc[j], it
            min_val = lowest
            if not np.isfinite(min_val):
                raise ValueError("cost matrix is infeasible")
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            nrem -= 1
            remaining[index] = remaining[nrem]
        u[cur] += min_val
        for i2 in np.nonzero(SR)[0]:
            if i2 != cur:
                u[i2] += min_val - spc[col4row[i2]]
        v[SC] -= min_val - spc[SC]
        j = sink
        while True:
            i2 = path[j]
            row4col[j] = i2
            col4row[i2], j = j, col4row[i2]
            if i2 == cur:
                break
    if transposed:
        order = np.argsort(col4row)
        return col4row[order], order
    return np.arange(nr), col4row


# --------------------------------------------------------------------------- #
# precise (barycentric) map -- FunctionalMapping.get_precise_map, functional.py:221-251
# --------------------------------------------------------------------------- #
def _point_triangle(a, b, c, d, e, f, multi):
    """Closest point of a triangle to a point, in the triangle's (s, t) coordinates -- D. Eberly's regions as coded in
    pyFM/spectral/projection_utils.py.  multi = False: pointTriangleDistance (:731-998), used when a point has ONE
    candidate face.  multi = True: the vectorised point_to_triangles_projection (:369-728), used otherwise; it differs
    from the scalar code in region 4 only, where two of its squared distances are formed with the unclamped s / t
    (:535, :559) -- kept, because the face with the smallest distance is chosen on these values.
    Returns (s, t, squared distance)."""
    det = a * c - b * b
    s = b * e - c * d
    t = b * d - a * e
    if s + t <= det:
        if s < 0.0:
            if t < 0.0:                                           # region 4
                if d < 0:
                    if -d >= a:
                        return 1.0, 0.0, a + 2.0 * d + f
                    s_ = -d / a
                    return s_, 0.0, (d * s + f) if multi else (d * s_ + f)
                if e >= 0.0:
                    return 0.0, 0.0, f
                if -e >= c:
                    return 0.0, 1.0, c + 2.0 * e + f
                t_ = -e / c
                return 0.0, t_, (e * t + f) if multi else (e * t_ + f)
            if e >= 0:                                            # region 3
                return 0.0, 0.0, f
            if -e >= c:
                return 0.0, 1.0, c + 2.0 * e + f
            t_ = -e / c
            return 0.0, t_, e * t_ + f
        if t < 0:                                                 # region 5
            if d >= 0:
                return 0.0, 0.0, f
            if -d >= a:
                return 1.0, 0.0, a + 2.0 * d + f
            s_ = -d / a
            return s_, 0.0, d * s_ + f
        inv = 1.0 / det                                           # region 0
        s_, t_ = s * inv, t * inv
        return s_, t_, s_ * (a * s_ + b * t_ + 2.0 * d) + t_ * (b * s_ + c * t_ + 2.0 * e) + f
    if s < 0.0:                                                   # region 2
        tmp0, tmp1 = b + d, c + e
        if tmp1 > tmp0:
            numer, denom = tmp1 - tmp0, a - 2.0 * b + c
            if numer >= denom:
                return 1.0, 0.0, a + 2.0 * d + f
            s_ = numer / denom
            t_ = 1 - s_
            return s_, t_, s_ * (a * s_ + b * t_ + 2 * d) + t_ * (b * s_ + c * t_ + 2 * e) + f
        if tmp1 <= 0.0:
            return 0.0, 1.0, c + 2.0 * e + f
        if e >= 0.0:
            return 0.0, 0.0, f
        t_ = -e / c
        return 0.0, t_, e * t_ + f
    if t < 0.0:                                                   # region 6
        tmp0, tmp1 = b + e, a + d
        if tmp1 > tmp0:
            numer, denom = tmp1 - tmp0, a - 2.0 * b + c
            if numer >= denom:
                return 0.0, 1.0, c + 2.0 * e + f
            t_ = numer / denom
            s_ = 1 - t_
            return s_, t_, s_ * (a * s_ + b * t_ + 2.0 * d) + t_ * (b * s_ + c * t_ + 2.0 * e) + f
        if tmp1 <= 0.0:
            return 1.0, 0.0, a + 2.0 * d + f
        if d >= 0.0:
            return 0.0, 0.0, f
        s_ = -d / a
        return s_, 0.0, d * s_ + f
    numer = c + e - b - d                                         # region 1
    if numer <= 0:
        return 0.0, 1.0, c + 2.0 * e + f
    denom = a - 2.0 * b + c
    if numer >= denom:
        return 1.0, 0.0, a + 2.0 * d + f
    s_ = numer / denom
    t_ = 1 - s_
    return s_, t_, s_ * (a * s_ + b * t_ + 2.0 * d) + t_ * (b * s_ + c * t_ + 2.0 * e) + f


def project_pc_to_triangles(vert_emb, faces, points_emb):
    """For every point the face of the embedded mesh it projects onto and the barycentric coordinates of the projection
    -- pyFM/spectral/projection_utils.py:16-115 (precompute_dmin = True): candidate faces are those whose nearest
    vertex is closer than the point's nearest vertex plus the face's longest edge (:356), the closest projection wins
    (first face index on equal distances).  Returns (face_match (n2,), bary (n2,3))."""
    V = np.asarray(vert_emb, dtype=np.float64)
    P = np.asarray(points_emb, dtype=np.float64)
    faces = np.asarray(faces)
    e0, e1, e2 = V[faces[:, 0]], V[faces[:, 1]], V[faces[:, 2]]
    lmax = np.max(np.stack([np.linalg.norm(e1 - e0, axis=1), np.linalg.norm(e2 - e1, axis=1), np.linalg.norm(e0 - e2, axis=1)]), axis=0)   # :118-142
    nn = knn_query(V, P)
    Deltamin = np.linalg.norm(V[nn] - P, axis=1)                                                   # :145-178
    vs, ps = np.linalg.norm(V, axis=1) ** 2, np.linalg.norm(P, axis=1) ** 2
    face_match = np.zeros(P.shape[0], dtype=np.int64)
    bary = np.zeros((P.shape[0], 3))
    ax1, ax2 = e1 - e0, e2 - e0
    fa, fb, fc = np.einsum("ij,ij->i", ax1, ax1), np.einsum("ij,ij->i", ax1, ax2), np.einsum("ij,ij->i", ax2, ax2)
    for i in range(P.shape[0]):
        dv = np.sqrt(np.maximum((-2.0 * (V @ P[i]) + vs) + ps[i], 0.0))                           # mycdist, :181-229
        dmin = np.minimum(np.minimum(dv[faces[:, 0]], dv[faces[:, 1]]), dv[faces[:, 2]])           # :282-327
        cand = np.where(dmin - lmax < Deltamin[i])[0]                                              # :356
        multi = len(cand) > 1
        best = None
        for fi in cand:
            diff = e0[fi] - P[i]
            s, t, sq = _point_triangle(fa[fi], fb[fi], fc[fi], ax1[fi] @ diff, ax2[fi] @ diff, diff @ diff, multi)
            dist = np.sqrt(max(sq, 0.0))
            if best is None or dist < best[0]:
                best = (dist, fi, s, t)
        face_match[i] = best[1]
        bary[i] = (1.0 - best[2] - best[3], best[2], best[3])
    return face_match, bary


def precise_map_dense(C, phi1, phi2, faces1):
    """get_precise_map().toarray() -- functional.py:221-251 -> convert.py:185-229 (use_adj = True: emb1 = Phi1[:, :k1],
    emb2 = Phi2[:, :k2] C) -> projection_utils.py:16,360-399 (three barycentric weights per row)."""
    k2, k1 = C.shape
    e1 = np.asarray(phi1[:, :k1], dtype=np.float64)
    e2 = np.asarray(phi2[:, :k2], dtype=np.float64) @ C
    fm, bary = project_pc_to_triangles(e1, faces1, e2)
    M = np.zeros((e2.shape[0], e1.shape[0]))
    faces1 = np.asarray(faces1)
    for c in range(3):
        np.add.at(M, (np.arange(e2.shape[0]), faces1[fm, c]), bary[:, c])
    return M, fm, bary


def knn_query_topk(X, Y, k):
    """The k nearest rows of X for every row of Y, nearest first, lowest index on equal distances
    -- pyFM/spectral/nn_utils.py:4-38 with k > 1 (sklearn kneighbors returns neighbours sorted by distance).
    Brute force on directly-formed differences -> (dist (ny,k), idx (ny,k))."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    d = np.sqrt(((Y[:, None, :] - X[None, :, :]) ** 2).sum(-1))
    idx = np.argsort(d, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(d, idx, axis=1), idx

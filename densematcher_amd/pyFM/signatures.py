"""
Spectral point signatures with the reference's interface and parameter choices
(densematcher/pyFM/signatures/{HKS_functions,WKS_functions}.py): an alternative descriptor source of
FunctionalMapping.preprocess (functional.py:308-329) besides the neural features.  Host NumPy float64 like the
reference -- one (N x k) by (k x n_descr) product per mesh, not part of the accelerated path.

Both signatures are   S[n, t] = sum_k w[t, k] Phi[n, k]^2 / sum_k w[t, k]     (scaled=True everywhere in the reference)
and their landmark versions   S_p[n, t] = sum_k w[t, k] Phi[p, k] Phi[n, k] / sum_k w[t, k]   for each landmark p:
  HKS:  w[t, k] = exp(-t lambda_k),  t log-spaced in [4 ln10 / lambda_max, 4 ln10 / lambda_1]        HKS_functions.py:97-98
  WKS:  w[e, k] = exp(-(e - ln lambda_k)^2 / (2 sigma^2)),  sigma = 7 (ln lambda_max - ln lambda_1) / n,
        e linearly spaced in [ln lambda_1 + 2 sigma, ln lambda_max - 2 sigma], eigenvalues <= 1e-5 (landmark version:
        <= 1e-2) left out                                                                             WKS_functions.py:29,71,118-126
"""
import numpy as np


def _weighted(weights, evects, landmarks):
    """weights (T, K), evects (N, K) -> (N, T), or (N, p*T) with the landmark-major column order of the reference
    (HKS_functions.py:73: reshape of a (p, T, N) array)."""
    scale = 1.0 / weights.sum(axis=1)                                        # (T,)
    if landmarks is None:
        return (np.square(evects) @ weights.T) * scale[None, :]
    lm = np.asarray(landmarks).reshape(-1)
    cols = [(evects * evects[p][None, :]) @ weights.T * scale[None, :] for p in lm]      # each (N, T)
    return np.concatenate(cols, axis=1)


def auto_HKS(evals, evects, num_T, landmarks=None, scaled=True):
    if not scaled:
        raise NotImplementedError("the reference only ever calls the scaled signature")
    lam = np.sort(np.abs(np.asarray(evals, dtype=np.float64).reshape(-1)))
    times = np.geomspace(4 * np.log(10) / lam[-1], 4 * np.log(10) / lam[1], num_T)
    weights = np.exp(-np.outer(times, lam))
    return _weighted(weights, np.asarray(evects, dtype=np.float64), landmarks)


def auto_WKS(evals, evects, num_E, landmarks=None, scaled=True):
    if not scaled:
        raise NotImplementedError("the reference only ever calls the scaled signature")
    lam = np.sort(np.abs(np.asarray(evals, dtype=np.float64).reshape(-1)))
    e_min, e_max = np.log(lam[1]), np.log(lam[-1])
    sigma = 7 * (e_max - e_min) / num_E
    assert sigma > 0, f"Sigma should be positive ! Given value : {sigma}"
    energies = np.linspace(e_min + 2 * sigma, e_max - 2 * sigma, num_E)
    keep = lam > (1e-5 if landmarks is None else 1e-2)
    weights = np.exp(-np.square(energies[:, None] - np.log(lam[keep])[None, :]) / (2 * sigma ** 2))
    return _weighted(weights, np.asarray(evects, dtype=np.float64)[:, keep], landmarks)


def _mesh_signature(fn, mesh, num, landmarks, k):
    assert mesh.eigenvalues is not None, "Eigenvalues should be processed"
    if k is None:
        k = len(mesh.eigenvalues)
    return fn(mesh.eigenvalues[:k], mesh.eigenvectors[:, :k], num, landmarks=landmarks, scaled=True)


def mesh_HKS(mesh, num_T, landmarks=None, k=None):
    """Heat kernel signature of a processed mesh, (N, num_T) or (N, p*num_T)  -- HKS_functions.py:106-135"""
    return _mesh_signature(auto_HKS, mesh, num_T, landmarks, k)


def mesh_WKS(mesh, num_E, landmarks=None, k=None):
    """Wave kernel signature of a processed mesh, (N, num_E) or (N, p*num_E)  -- WKS_functions.py:127-152"""
    return _mesh_signature(auto_WKS, mesh, num_E, landmarks, k)

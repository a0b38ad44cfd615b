import numpy as np


def knn_query(X, Y, k=1, return_distance=False, n_jobs=1):
    """The k nearest rows of X for every row of Y (reference: pyFM/spectral/nn_utils.py:4-38, sklearn kd-tree; returns
    (n2,) for k = 1, (n2, k) otherwise, nearest first).  Exact float64 brute force on the GPU, lowest index on ties:
    k = 1 through the matrix-core search of the matching path (dm_knn_query_f64), k > 1 through dm_knn_query_topk_f64."""
    from ...engine import default_engine
    X = np.ascontiguousarray(X, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    eng = default_engine()
    if k == 1:
        matches = eng.knn_query(X[None], Y[None])[0].cpu().numpy().astype(np.int64)
        if return_distance:
            return np.linalg.norm(X[matches] - Y, axis=1), matches
        return matches
    idx, dist = eng.knn_query_topk(X[None], Y[None], int(k))
    matches = idx[0].cpu().numpy().astype(np.int64)
    if return_distance:
        return dist[0].cpu().numpy(), matches
    return matches

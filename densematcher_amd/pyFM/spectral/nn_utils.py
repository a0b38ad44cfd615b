import numpy as np


def knn_query(X, Y, k=1, return_distance=False, n_jobs=1):
    """Nearest neighbour of every row of Y among the rows of X (reference: pyFM/spectral/nn_utils.py:4-38,
    sklearn kd-tree).  Exact float64 brute force on the f64 matrix cores, lowest index on ties.  Only k = 1 is on
    the matching path."""
    if k != 1:
        raise NotImplementedError("only k = 1 is used by the matching path")
    from ...engine import default_engine
    X = np.ascontiguousarray(X, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    matches = default_engine().knn_query(X[None], Y[None])[0].cpu().numpy().astype(np.int64)
    if return_distance:
        dists = np.linalg.norm(X[matches] - Y, axis=1)
        return dists, matches
    return matches

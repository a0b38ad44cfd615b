"""
Synthetic inputs for the matching hot path (host side, NumPy/SciPy).

These are the *inputs* of the path -- meshes, Laplace-Beltrami eigenbases,
lumped masses and per-vertex descriptors -- generated as SURVEY.md section 8(d)
prescribes.  Producing an eigenbasis is outside the accelerated path (in the
reference it is `TriMesh.process`, pyFM/mesh/trimesh.py:498-531 ->
pyFM/mesh/laplacian.py:143-182, SciPy/ARPACK on the host); nothing here is
timed by bench.py.
"""
import hashlib

import numpy as np
import scipy.linalg
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# --------------------------------------------------------------------------- #
# meshes
# --------------------------------------------------------------------------- #
def torus_mesh(nu, nv, R=1.0, r=0.4, perturb=0.0, seed=0):
    """Closed torus triangulated on an nu x nv periodic grid (N = nu*nv).
    `perturb` > 0 moves every vertex radially (smooth low-frequency bump +
    a little white jitter) to make a second, near-isometric shape."""
    u = 2.0 * np.pi * np.arange(nu) / nu
    v = 2.0 * np.pi * np.arange(nv) / nv
    U, V = np.meshgrid(u, v, indexing="ij")
    rr = np.full_like(U, r)
    if perturb > 0.0:
        rng = np.random.default_rng(seed)
        ph = rng.uniform(0.0, 2.0 * np.pi, size=4)
        rr = rr * (1.0 + perturb * (np.sin(2 * U + ph[0]) * np.cos(V + ph[1])
                                    + 0.5 * np.cos(3 * U + ph[2]) * np.sin(2 * V + ph[3]))
                   + 0.05 * perturb * rng.standard_normal(U.shape))
    x = (R + rr * np.cos(V)) * np.cos(U)
    y = (R + rr * np.cos(V)) * np.sin(U)
    z = rr * np.sin(V)
    verts = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)

    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    i1 = (i + 1) % nu
    j1 = (j + 1) % nv
    v00 = (i * nv + j).ravel()
    v10 = (i1 * nv + j).ravel()
    v01 = (i * nv + j1).ravel()
    v11 = (i1 * nv + j1).ravel()
    faces = np.concatenate([np.stack([v00, v10, v11], axis=1),
                            np.stack([v00, v11, v01], axis=1)], axis=0)
    return verts, faces.astype(np.int64)


def cotan_laplacian(verts, faces):
    """Cotangent stiffness matrix W (CSR, symmetric PSD) and lumped vertex
    masses a (one third of the incident triangle areas)."""
    verts = np.asarray(verts, dtype=np.float64)
    faces = np.asarray(faces, dtype=np.int64)
    n = verts.shape[0]
    rows, cols, vals = [], [], []
    mass = np.zeros(n)
    for c in range(3):
        # corner c of every face; the opposite edge is (p, q)
        o = faces[:, c]
        p = faces[:, (c + 1) % 3]
        q = faces[:, (c + 2) % 3]
        e1 = verts[p] - verts[o]
        e2 = verts[q] - verts[o]
        cr = np.linalg.norm(np.cross(e1, e2), axis=1)
        cot = np.einsum("ij,ij->i", e1, e2) / np.maximum(cr, 1e-300)
        w = 0.5 * cot
        rows += [p, q, p, q]
        cols += [q, p, p, q]
        vals += [-w, -w, w, w]
        if c == 0:
            np.add.at(mass, faces.ravel(), np.repeat(cr / 6.0, 3))
    W = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(n, n)).tocsr()
    return W, mass


def eigenbasis(verts, faces, k, method="auto"):
    """First k generalised eigenpairs of W phi = lam diag(a) phi.
    Returns (lam (k,), Phi (N,k) with Phi^T diag(a) Phi = I, a (N,))."""
    W, a = cotan_laplacian(verts, faces)
    n = W.shape[0]
    if method == "auto":
        method = "dense" if n <= 3000 else "arpack"
    if method == "dense":
        s = 1.0 / np.sqrt(a)
        S = (W.multiply(s[:, None]).multiply(s[None, :])).toarray()
        lam, Y = scipy.linalg.eigh(S, subset_by_index=[0, k - 1])
        phi = Y * s[:, None]
    else:
        lam, phi = spla.eigsh(W, k=k, M=sp.diags(a).tocsc(), sigma=-0.01)
        order = np.argsort(lam)
        lam, phi = lam[order], phi[:, order]
    return lam, phi, a


def random_basis(n, k, seed):
    """Throughput-only stand-in for an eigenbasis: a
    seeded Gaussian made mass-orthonormal by QR, column 0 constant, with
    sorted positive 'eigenvalues'.  Parity-irrelevant; used where an N=8192
    eigensolve would dominate a test's run time."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(0.5, 1.5, size=n) / n
    G = rng.standard_normal((n, k))
    G[:, 0] = 1.0
    Qm, _ = np.linalg.qr(np.sqrt(a)[:, None] * G)
    phi = Qm / np.sqrt(a)[:, None]
    if phi[0, 0] < 0:
        phi[:, 0] = -phi[:, 0]
    lam = np.sort(rng.uniform(0.0, 1.0, size=k)) * (4.0 * k)
    lam[0] = 0.0
    return lam, phi, a


# --------------------------------------------------------------------------- #
# descriptors
# --------------------------------------------------------------------------- #
def _unit_rows(x):
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def feature_pair(n1, n2, d, seed1, seed2, sigma=0.1, perm="random"):
    """F1 = unit_rows(N(0,1)) (seed1); F2 = unit_rows(F1[perm] + sigma N(0,1))
    (seed2); both returned as fp16 (what the reference receives from the
    autocast DiffusionNet, example.ipynb cells 2/5/11).  Also returns perm
    (ground-truth map 2->1).  perm="identity" keeps vertex i of mesh 2 matched
    to vertex i of mesh 1, so that a functional map between two near-isometric
    meshes with the same connectivity is meaningful (C close to a signed
    permutation of the low eigenfunctions)."""
    r1 = np.random.default_rng(seed1)
    r2 = np.random.default_rng(seed2)
    F1 = _unit_rows(r1.standard_normal((n1, d)))
    if isinstance(perm, str) and perm == "identity":
        perm = np.arange(n2) % n1
    else:
        perm = r2.permutation(n1)[:n2] if n2 <= n1 else r2.integers(0, n1, size=n2)
    F2 = _unit_rows(F1[perm] + sigma * r2.standard_normal((n2, d)))
    return F1.astype(np.float16), F2.astype(np.float16), perm


def smooth_feature_pair(phi1, phi2, d, seed1, seed2, sigma=0.05):
    """'Smooth' descriptors that mimic real network features: random spectral
    coefficients with 1/sqrt(j) decay expanded in each mesh's own basis
    (meshes are near-isometric so the same coefficients give corresponding
    functions), plus a little noise; rows normalised; fp16."""
    r1 = np.random.default_rng(seed1)
    r2 = np.random.default_rng(seed2)
    k = min(phi1.shape[1], phi2.shape[1])
    coef = r1.standard_normal((k, d)) / np.sqrt(np.arange(1, k + 1))[:, None]
    F1 = phi1[:, :k] @ coef
    F2 = phi2[:, :k] @ coef
    F1 = _unit_rows(F1 + sigma * F1.std() * r1.standard_normal(F1.shape))
    F2 = _unit_rows(F2 + sigma * F2.std() * r2.standard_normal(F2.shape))
    return F1.astype(np.float16), F2.astype(np.float16)


def sha256_of(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


# --------------------------------------------------------------------------- #
# whole batches (bench.py / tests)
# --------------------------------------------------------------------------- #
def make_pair_batch(B, nu, nv, d, k, sigma=0.1, n_distinct_meshes=2, basis="eig", seed0=0, perm="identity", real_dtype=np.float32):
    """A batch of B pairs in the layout the C ABI takes (batch-major,
    row-major, contiguous): Phi1/Phi2 (B,N,k) f32, lam1/lam2 (B,k) f64,
    a1/a2 (B,N) f32, F1/F2 (B,N,d) f16.  real_dtype = np.float64 keeps the eigenvectors and masses in the
    float64 the eigensolver produced (the reference's dtype; the *_f64 entry points).  Pair i uses descriptor seeds
    (1000+i, 2000+i).  Eigenbases are computed for
    `n_distinct_meshes` shape pairs and cycled over the batch."""
    n = nu * nv
    bases = []
    for m in range(n_distinct_meshes):
        if basis == "eig":
            v1, f1 = torus_mesh(nu, nv, perturb=0.0 if m == 0 else 0.03, seed=seed0 + 10 * m)
            v2, f2 = torus_mesh(nu, nv, perturb=0.08, seed=seed0 + 10 * m + 1)
            b1 = eigenbasis(v1, f1, k)
            b2 = eigenbasis(v2, f2, k)
        else:
            b1 = random_basis(n, k, seed0 + 10 * m)
            b2 = random_basis(n, k, seed0 + 10 * m + 1)
        bases.append((b1, b2))
    out = {
        "Phi1": np.empty((B, n, k), real_dtype), "Phi2": np.empty((B, n, k), real_dtype),
        "lam1": np.empty((B, k), np.float64), "lam2": np.empty((B, k), np.float64),
        "a1": np.empty((B, n), real_dtype), "a2": np.empty((B, n), real_dtype),
        "F1": np.empty((B, n, d), np.float16), "F2": np.empty((B, n, d), np.float16),
    }
    for i in range(B):
        (l1, p1, m1), (l2, p2, m2) = bases[i % n_distinct_meshes]
        out["Phi1"][i], out["Phi2"][i] = p1, p2
        out["lam1"][i], out["lam2"][i] = l1, l2
        out["a1"][i], out["a2"][i] = m1, m2
        out["F1"][i], out["F2"][i], _ = feature_pair(n, n, d, 1000 + i, 2000 + i, sigma, perm=perm)
    return out

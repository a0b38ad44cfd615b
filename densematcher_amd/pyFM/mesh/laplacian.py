"""
Laplacians of a triangle mesh, assembled on the host (they are INPUTS of the matching path; the eigensolve runs on the GPU).

`robust_mesh_laplacian` restates what the reference gets from the external `robust_laplacian` wheel
(`robust_laplacian.mesh_laplacian(V, F, mollify_factor=1e-5)`, called by pyFM/mesh/trimesh.py:465-470 whenever
`process(robust=True)` -- i.e. always, pyFM/functional.py:294-295): the "tufted" intrinsic-Delaunay Laplacian of
N. Sharp and K. Crane, "A Laplacian for Nonmanifold Triangle Meshes", SGP 2020:

  1. intrinsic edge lengths, mollified: the smallest eps is added to EVERY edge length such that each triangle satisfies the
     triangle inequality with margin delta = mollify_factor * (mean edge length);
  2. the tufted cover: every face gets a front and a back copy; around each edge with incident faces f_0 .. f_{n-1} the
     copy of f_i that runs b -> a is glued to the copy of f_{i+1} that runs a -> b (cyclically; n = 1 glues the face's own two
     copies along a boundary edge, n = 2 on a consistently oriented manifold mesh glues fronts to fronts and backs to
     backs).  The cover is an edge-manifold surface without boundary whatever the input;
  3. intrinsic edge flips (lengths only) until every cover edge is Delaunay (cot alpha + cot beta >= 0);
  4. cotangent weights and lumped areas of the cover, accumulated on the original vertices, times 1/2.

The wheel is not installed here and the reference has no fixture of its output: PARITY UNPINNED against the wheel itself.
What is tested instead (tests/test_laplacian_cpu.py): on a Delaunay mesh the result is the cotangent Laplacian; flipping edges
of a planar triangulation (same geometry, worse triangles) gives the SAME matrix (the intrinsic Delaunay triangulation is
unique); all edge weights are non-negative, rows sum to zero, masses sum to the area.
"""
import os

import numpy as np
import scipy.sparse as sparse

# What `process(robust=True)` may run when the reference's `robust_laplacian` wheel cannot be imported:
#   "wheel"     (default) nothing: ImportError -- a drop-in must not silently compute on an operator that is not pinned to the reference's
#   "restated"  this package's own construction (dm_tufted_cover + device assembly in the product; robust_mesh_laplacian below is
#               the NumPy restatement the CPU tests pin it to)
_ROBUST_BACKEND = [os.environ.get("DENSEMATCHER_AMD_ROBUST_LAPLACIAN", "wheel")]


def set_robust_backend(name):
    if name not in ("wheel", "restated"):
        raise ValueError("robust backend must be 'wheel' or 'restated'")
    _ROBUST_BACKEND[0] = name


def robust_backend():
    return _ROBUST_BACKEND[0]


def cotangent_laplacian(verts, faces):
    """Classical cotangent stiffness matrix W (CSR) and lumped masses (reference pyFM/mesh/laplacian.py:88, :5)."""
    from ... import synth
    return synth.cotan_laplacian(verts, faces)


def _areas_and_cots(l):
    """l (T, 3): side lengths, side s runs from corner s to corner s + 1.  Returns the triangle areas (T,) and cot (T, 3):
    cot[:, s] = cotangent of the angle OPPOSITE side s (at corner s + 2)."""
    a, b, c = l[:, 0], l[:, 1], l[:, 2]
    s = 0.5 * (a + b + c)
    area = np.sqrt(np.maximum(s * (s - a) * (s - b) * (s - c), 0.0))
    den = 4.0 * np.maximum(area, 1e-300)
    cot = np.stack([(b * b + c * c - a * a) / den, (c * c + a * a - b * b) / den, (a * a + b * b - c * c) / den], axis=1)
    return area, cot


def robust_mesh_laplacian(verts, faces, mollify_factor=1e-5, max_flips=None):
    """(W, M): (n, n) CSR stiffness matrix (positive semi-definite, rows sum to 0) and (n, n) diagonal CSR mass matrix of
    the tufted intrinsic-Delaunay Laplacian (see the module docstring)."""
    V = np.asarray(verts, dtype=np.float64)
    F = np.asarray(faces, dtype=np.int64)
    n, nf = V.shape[0], F.shape[0]
    # ---- 1. intrinsic lengths and mollification
    l0 = np.linalg.norm(V[F[:, [1, 2, 0]]] - V[F], axis=2)             # (nf, 3): side s = F[:, s] -> F[:, s + 1]
    delta = mollify_factor * l0.mean()
    viol = np.stack([delta + l0[:, 0] - l0[:, 1] - l0[:, 2], delta + l0[:, 1] - l0[:, 2] - l0[:, 0], delta + l0[:, 2] - l0[:, 0] - l0[:, 1]])
    eps = max(0.0, float(viol.max()))
    l0 = l0 + eps
    # ---- 2. tufted cover: faces 0 .. nf-1 = front copies, nf .. 2 nf - 1 = back copies (reversed orientation)
    T = np.concatenate([F, F[:, [0, 2, 1]]])                            # back copy of (a, b, c) is (a, c, b)
    L = np.concatenate([l0, l0[:, [2, 1, 0]]])                          # its sides: a->c (= side 2), c->b (= side 1), b->a (= side 0)
    nt = 2 * nf
    glue_t = np.full((nt, 3), -1, dtype=np.int64)
    glue_s = np.full((nt, 3), -1, dtype=np.int64)
    # the halfedges of the cover, grouped by undirected edge: for the front side (f, s): a = F[f, s] -> b = F[f, s + 1];
    # its reversed twin lives in the back copy: front side s <-> back side (2 - s)
    fa, fb = F, F[:, [1, 2, 0]]
    lo, hi = np.minimum(fa, fb), np.maximum(fa, fb)
    key = (lo * n + hi).ravel()                                         # (nf * 3,) undirected edge id of front side (f, s)
    order = np.argsort(key, kind="stable")                              # incident faces of an edge in face order
    sk = key[order]
    starts = np.flatnonzero(np.r_[True, sk[1:] != sk[:-1]])
    ends = np.r_[starts[1:], len(sk)]
    f_of, s_of = order // 3, order % 3
    fwd = (fa.ravel()[order] == lo.ravel()[order])                      # does the front side run lo -> hi ?
    # for the i-th incident face: plus = the copy whose side runs lo -> hi, minus = the copy that runs hi -> lo
    plus_t = np.where(fwd, f_of, f_of + nf)
    plus_s = np.where(fwd, s_of, 2 - s_of)
    minus_t = np.where(fwd, f_of + nf, f_of)
    minus_s = np.where(fwd, 2 - s_of, s_of)
    nxt = np.arange(len(order)) + 1
    nxt[ends - 1] = starts                                              # cyclic successor inside each edge's group
    glue_t[minus_t, minus_s] = plus_t[nxt]
    glue_s[minus_t, minus_s] = plus_s[nxt]
    glue_t[plus_t[nxt], plus_s[nxt]] = minus_t
    glue_s[plus_t[nxt], plus_s[nxt]] = minus_s
    assert (glue_t >= 0).all()
    # ---- 3. intrinsic Delaunay flips
    flips, limit = 0, (20 * nt if max_flips is None else max_flips)
    # 3a. vectorised rounds: all non-Delaunay edges at once, of which an independent set (no two flipped pairs share or
    # neighbour a triangle, decided by random priorities) is flipped with array operations; the intrinsic Delaunay
    # triangulation is unique, so the order of the flips does not matter.  A mesh of 4 k triangles needs ~2 k flips when its
    # quads are close to cocircular: one by one in the interpreter that was 80 ms of the 150 ms a TriMesh.process call took.
    rng = np.random.default_rng(0)
    ar3 = np.arange(3)
    for _round in range(200):
        _, cot = _areas_and_cots(L)
        bad = (cot + cot[glue_t, glue_s] < -1e-12) & (glue_t != np.arange(nt)[:, None])
        bt, bs = np.nonzero(bad)
        keep = bt < glue_t[bt, bs]                                        # each edge once (from its lower triangle)
        bt, bs = bt[keep], bs[keep]
        if len(bt) == 0 or flips >= limit:
            break
        t2, s2 = glue_t[bt, bs], glue_s[bt, bs]
        s_1, s_2, q_1, q_2 = (bs + 1) % 3, (bs + 2) % 3, (s2 + 1) % 3, (s2 + 2) % 3
        touched = np.stack([bt, t2, glue_t[bt, s_1], glue_t[bt, s_2], glue_t[t2, q_1], glue_t[t2, q_2]], axis=1)   # (c, 6)
        prio = rng.permutation(len(bt)) + 1
        best = np.zeros(nt, dtype=np.int64)
        np.maximum.at(best, touched.ravel(), np.repeat(prio, 6))
        win = (best[touched] == prio[:, None]).all(axis=1)
        # (a pair whose two triangles coincide with a neighbour, e.g. tiny closed surfaces: left to the sequential loop)
        distinct = (np.sort(touched, axis=1)[:, 1:] != np.sort(touched, axis=1)[:, :-1]).all(axis=1)
        win &= distinct
        if not win.any():
            break
        t, s, t2, s2 = bt[win], bs[win], t2[win], s2[win]
        s_1, s_2, q_1, q_2 = s_1[win], s_2[win], q_1[win], q_2[win]
        i, j, k, m = T[t, s], T[t, s_1], T[t, s_2], T[t2, q_2]
        lij, ljk, lki, lim, lmj = L[t, s], L[t, s_1], L[t, s_2], L[t2, q_1], L[t2, q_2]
        xk = (lki * lki - ljk * ljk + lij * lij) / (2.0 * lij)
        yk = np.sqrt(np.maximum(lki * lki - xk * xk, 0.0))
        xm = (lim * lim - lmj * lmj + lij * lij) / (2.0 * lij)
        ym = -np.sqrt(np.maximum(lim * lim - xm * xm, 0.0))
        lkm = np.hypot(xk - xm, yk - ym)
        ok = lkm > 0.0
        if not ok.all():
            t, s, t2, s2, s_1, s_2, q_1, q_2 = (a[ok] for a in (t, s, t2, s2, s_1, s_2, q_1, q_2))
            i, j, k, m, lij, ljk, lki, lim, lmj, lkm = (a[ok] for a in (i, j, k, m, lij, ljk, lki, lim, lmj, lkm))
            if len(t) == 0:
                break
        g_jk = (glue_t[t, s_1].copy(), glue_s[t, s_1].copy()); g_ki = (glue_t[t, s_2].copy(), glue_s[t, s_2].copy())
        g_im = (glue_t[t2, q_1].copy(), glue_s[t2, q_1].copy()); g_mj = (glue_t[t2, q_2].copy(), glue_s[t2, q_2].copy())
        # new triangles: t = (k, i, m): k->i, i->m, m->k;   t2 = (m, j, k): m->j, j->k, k->m   (as in the sequential flip below)
        T[t] = np.stack([k, i, m], axis=1); L[t] = np.stack([lki, lim, lkm], axis=1)
        T[t2] = np.stack([m, j, k], axis=1); L[t2] = np.stack([lmj, ljk, lkm], axis=1)
        for (ta, sa), (tb, sb) in (((t, 0), g_ki), ((t, 1), g_im), ((t2, 0), g_mj), ((t2, 1), g_jk)):
            sa_ = np.full(len(ta), sa)
            glue_t[ta, sa_], glue_s[ta, sa_] = tb, sb               # the neighbours are outside every flipped pair (independence)
            glue_t[tb, sb], glue_s[tb, sb] = ta, sa_
        two = np.full(len(t), 2)
        glue_t[t, two], glue_s[t, two] = t2, two
        glue_t[t2, two], glue_s[t2, two] = t, two
        flips += len(t)
    # 3b. whatever is left (pairs that touch themselves, the tail of a long flip sequence): one by one
    _, cot = _areas_and_cots(L)
    bad = cot + cot[glue_t, glue_s] < -1e-12
    stack = [(int(t), int(s)) for t, s in zip(*np.nonzero(bad))]

    def cot_opp(t, s):
        a, b, c = L[t, s], L[t, (s + 1) % 3], L[t, (s + 2) % 3]
        sp_ = 0.5 * (a + b + c)
        ar = np.sqrt(max(sp_ * (sp_ - a) * (sp_ - b) * (sp_ - c), 0.0))
        return (b * b + c * c - a * a) / (4.0 * max(ar, 1e-300))

    while stack and flips < limit:
        t, s = stack.pop()
        t2, s2 = int(glue_t[t, s]), int(glue_s[t, s])
        if t2 == t:
            continue                                                    # an edge glued to its own face cannot be flipped
        if cot_opp(t, s) + cot_opp(t2, s2) >= -1e-12:
            continue
        # triangle t: i -> j (side s), j -> k, k -> i;  triangle t2: j -> i (side s2), i -> m, m -> j
        i, j, k = T[t, s], T[t, (s + 1) % 3], T[t, (s + 2) % 3]
        m = T[t2, (s2 + 2) % 3]
        lij, ljk, lki = L[t, s], L[t, (s + 1) % 3], L[t, (s + 2) % 3]
        lim, lmj = L[t2, (s2 + 1) % 3], L[t2, (s2 + 2) % 3]
        # unfold the two triangles in the plane: i = (0, 0), j = (lij, 0), k above the axis, m below
        xk = (lki * lki - ljk * ljk + lij * lij) / (2.0 * lij)
        yk = np.sqrt(max(lki * lki - xk * xk, 0.0))
        xm = (lim * lim - lmj * lmj + lij * lij) / (2.0 * lij)
        ym = -np.sqrt(max(lim * lim - xm * xm, 0.0))
        lkm = float(np.hypot(xk - xm, yk - ym))
        if not (lkm > 0.0):
            continue
        # neighbours across the four outer sides
        g_jk = (int(glue_t[t, (s + 1) % 3]), int(glue_s[t, (s + 1) % 3]))
        g_ki = (int(glue_t[t, (s + 2) % 3]), int(glue_s[t, (s + 2) % 3]))
        g_im = (int(glue_t[t2, (s2 + 1) % 3]), int(glue_s[t2, (s2 + 1) % 3]))
        g_mj = (int(glue_t[t2, (s2 + 2) % 3]), int(glue_s[t2, (s2 + 2) % 3]))
        # new triangles: t = (k, i, m): k->i, i->m, m->k;   t2 = (m, j, k): m->j, j->k, k->m
        T[t] = (k, i, m)
        L[t] = (lki, lim, lkm)
        T[t2] = (m, j, k)
        L[t2] = (lmj, ljk, lkm)

        def link(ta, sa, tb, sb):
            glue_t[ta, sa], glue_s[ta, sa] = tb, sb
            glue_t[tb, sb], glue_s[tb, sb] = ta, sa
        # (a neighbour may be one of the two triangles themselves: translate its old side to the new one)
        old2new = {(t, (s + 2) % 3): (t, 0), (t2, (s2 + 1) % 3): (t, 1), (t2, (s2 + 2) % 3): (t2, 0), (t, (s + 1) % 3): (t2, 1)}
        for (ta, sa), g in (((t, 0), g_ki), ((t, 1), g_im), ((t2, 0), g_mj), ((t2, 1), g_jk)):
            tb, sb = old2new.get(g, g)
            link(ta, sa, tb, sb)
        link(t, 2, t2, 2)
        flips += 1
        stack += [(t, 0), (t, 1), (t2, 0), (t2, 1)]
    # ---- 4. assembly on the original vertices, times 1/2 for the double cover
    area, cot = _areas_and_cots(L)
    a_id, b_id = T, T[:, [1, 2, 0]]
    w = 0.25 * cot                                                      # 1/2 (cotangent weight) x 1/2 (cover)
    rows = np.concatenate([a_id.ravel(), b_id.ravel(), a_id.ravel(), b_id.ravel()])
    cols = np.concatenate([b_id.ravel(), a_id.ravel(), a_id.ravel(), b_id.ravel()])
    vals = np.concatenate([-w.ravel(), -w.ravel(), w.ravel(), w.ravel()])
    W = sparse.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    mass = np.zeros(n)
    np.add.at(mass, T.ravel(), np.repeat(0.5 * area / 3.0, 3))
    robust_mesh_laplacian.last_info = {"mollify_eps": eps, "flips": flips, "converged": not stack}
    return W, sparse.diags(mass).tocsr()

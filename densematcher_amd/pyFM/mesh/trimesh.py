"""
TriMesh: the data carrier of the matching path (reference: densematcher/pyFM/mesh/trimesh.py:16).

Only what the hot path consumes is kept: vertices / faces, the cotangent stiffness matrix W, the lumped
(diagonal) mass matrix A, L = A^-1 W, the Laplace-Beltrami spectrum, `area`, `process`, `project`, `decode`.
Producing the spectrum (reference trimesh.py:440-531 -> laplacian.py:143-182): the stiffness rows,
the lumped masses and A^-1/2 W A^-1/2 in ELL are assembled ON THE DEVICE for all meshes of a call (dm_laplacian_rows / dm_laplacian_ell),
the eigensolve -- ARPACK shift-invert there -- runs on the GPU (dm_eigenbasis: Chebyshev-filtered subspace iteration; meshes too
small for it take a dense Jacobi eigensolve).  `robust=False` is the classical cotangent Laplacian with lumped masses
(laplacian.cotangent_weights / dia_area_mat of the reference, same arithmetic).
The reference's `robust=True` uses the external `robust_laplacian` wheel (tufted intrinsic-Delaunay Laplacian with mollification);
it is used here too when it can be imported.  When it cannot, `robust=True` FAILS unless the caller has opted into this package's
own implementation of the same construction (laplacian.set_robust_backend("restated") or DENSEMATCHER_AMD_ROBUST_LAPLACIAN=restated
in the environment): its parity with the wheel is unpinned (the wheel is not installable here, no network), and a drop-in must
not silently compute on a different operator.  The restatement: dm_tufted_cover (host C++ like the wheel: mollified lengths,
tufted cover, intrinsic Delaunay flips) + the device assembly on the cover's intrinsic lengths.
"""
import warnings

import numpy as np
import scipy.sparse as sparse

from ... import synth


class TriMesh:
    def __init__(self, *args, **kwargs):
        assert 0 < len(args) < 3, "Provide vertices / faces"
        if isinstance(args[0], str):
            raise NotImplementedError("mesh file loading is outside the matching path: pass (vertices, faces)")
        self._W = None
        self._W_dev = None               # (cols (N, nnz), w (N, nnz)) device tensors of the last device assembly: W is built from them on demand
        self._W_recipe = None            # robust flag of a device assembly whose rows were released (release_device_rows): W is re-assembled on the host on demand
        self._A = None
        self._mass = None                # lumped masses of the last device assembly: A (CSR) is built from them on demand
        self._L = None
        self.eigenvalues = None
        self.eigenvectors = None
        self.vertlist = args[0]
        self.facelist = args[1] if len(args) > 1 else None

    # ------------------------------------------------------------- geometry
    @property
    def vertlist(self):
        return self._vertlist

    @vertlist.setter
    def vertlist(self, vertlist):
        vertlist = np.asarray(vertlist, dtype=float)          # trimesh.py:118 (float64 copy)
        if vertlist.ndim != 2:
            raise ValueError('Vertex list has to be 2D')
        elif vertlist.shape[1] != 3:
            raise ValueError('Vertex list requires 3D coordinates')
        self._vertlist = vertlist.copy()
        self._W = self._W_dev = self._W_recipe = self._A = self._mass = self._L = self.eigenvalues = self.eigenvectors = None

    @property
    def facelist(self):
        return self._facelist

    @facelist.setter
    def facelist(self, facelist):
        if facelist is not None:
            facelist = np.asarray(facelist)
            if facelist.ndim != 2:
                raise ValueError('Faces list has to be 2D')
            elif facelist.shape[1] != 3:
                raise ValueError('Each face is made of 3 points')
            self._facelist = facelist.astype(np.int64).copy()
        else:
            self._facelist = None

    @property
    def W(self):
        """cotangent stiffness matrix (CSR).  After a device assembly it is materialised on first use (the matching path itself
        never reads it)."""
        if self._W is None and self._W_dev is not None:
            cols, w = (t.cpu().numpy() for t in self._W_dev)
            n = self.n_vertices
            rows = np.repeat(np.arange(n), cols.shape[1])
            Wm = sparse.coo_matrix((w.ravel(), (rows, cols.ravel().astype(np.int64))), shape=(n, n)).tocsr()   # (padding: zeros on the diagonal)
            Wm.eliminate_zeros()
            self._W = Wm
            self._W_dev = None
        elif self._W is None and self._W_recipe is not None and self._facelist is not None:
            # the device rows were released when the mesh left compute_surface_map[_batch] (they are views of a whole chunk's arrays:
            # one surviving mesh would pin them all): the same construction on the host, for the rare caller that reads W
            from . import laplacian as _lap
            self._W = (_lap.robust_mesh_laplacian(self.vertlist, self.facelist, mollify_factor=1e-5)[0] if self._W_recipe == "robust"
                       else sparse.csr_matrix(_lap.cotangent_laplacian(self.vertlist, self.facelist)[0]))
        return self._W

    def release_device_rows(self):
        """forget the device copy of the stiffness rows (views of the batched assembly's (B, N, nnz) arrays, which they keep alive in
        HBM); `W` is re-assembled on the host if somebody asks for it later.  compute_surface_map[_batch] call this on the meshes
        they return."""
        if self._W_dev is not None:
            self._W_dev = None

    @W.setter
    def W(self, value):
        self._W = value
        self._W_dev = None
        self._W_recipe = None

    @property
    def A(self):
        """lumped (diagonal) mass matrix (CSR), built on first use after a device assembly (a batch of 128 meshes spent 14 ms in
        scipy.sparse constructors nobody read: the batched path takes `vertex_masses`)"""
        if self._A is None and self._mass is not None:
            self._A = sparse.diags(self._mass).tocsr()
        return self._A

    @A.setter
    def A(self, value):
        self._A = value
        self._mass = None if value is None else np.asarray(value.diagonal())

    @property
    def vertex_masses(self):
        """diag(A) as an array (None before the Laplacian exists)"""
        return self._mass

    vertices = property(lambda self: self._vertlist)
    faces = property(lambda self: self._facelist)
    n_vertices = property(lambda self: self._vertlist.shape[0])
    n_faces = property(lambda self: 0 if self._facelist is None else self._facelist.shape[0])

    @property
    def area(self):
        """trimesh.py:206-221: A.sum() once the Laplacian exists, else the sum of the face areas."""
        if self._mass is None and self.A is None:
            if self.facelist is None:
                return None
            v = self.vertlist
            f = self.facelist
            return 0.5 * np.linalg.norm(np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]), axis=1).sum()
        return self._mass.sum() if self._mass is not None else self.A.sum()

    # -- differential geometry of the faces (inputs of the orientation term of FunctionalMapping.fit; host arithmetic, as in
    #    the reference: pyFM/mesh/geometry.py)
    @property
    def normals(self):
        """(m, 3) unit face normals (trimesh.py:255-266 -> geometry.py:110-133)"""
        v, f = self.vertlist, self.facelist
        n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
        return n / np.linalg.norm(n, axis=1, keepdims=True)

    @property
    def faces_areas(self):
        v, f = self.vertlist, self.facelist
        return 0.5 * np.linalg.norm(np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]), axis=1)

    @property
    def vertex_areas(self):
        """trimesh.py:286-298: the row sums of A once the Laplacian exists, else a third of the adjacent face areas"""
        if self.A is None:
            f = self.facelist
            return np.bincount(f.ravel(), weights=np.repeat(self.faces_areas / 3.0, 3), minlength=self.n_vertices)
        return np.asarray(self.A.sum(1)).squeeze()

    def _hat_gradients(self):
        """gradients of the three hat functions on every face, 3 x (m, 3) (geometry.py:284-316)"""
        v, f, n = self.vertlist, self.facelist, self.normals
        inv2a = 1.0 / (2.0 * self.faces_areas)[:, None]
        v1, v2, v3 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        return np.cross(n, v3 - v2) * inv2a, np.cross(n, v1 - v3) * inv2a, np.cross(n, v2 - v1) * inv2a

    def gradient(self, f, normalize=False):
        """per-face gradient of a vertex function, (m, 3) (trimesh.py:949-972 -> geometry.py:373-430)"""
        fl = self.facelist
        g1, g2, g3 = self._hat_gradients()
        f = np.asarray(f, dtype=np.float64)
        grad = f[fl[:, 0], None] * g1 + f[fl[:, 1], None] * g2 + f[fl[:, 2], None] * g3
        if normalize:
            grad = grad / np.linalg.norm(grad, axis=1, keepdims=True)
        return grad

    def orientation_op(self, gradf, normalize=False, per_vert_area=None):
        """(n, n) sparse operator g -> <n x grad f, grad g> at the vertices (trimesh.py:992-1019 -> geometry.py:919-985);
        per_vert_area: the areas the rows are divided by (default: vertex_areas, like the reference's method)"""
        gradf = np.asarray(gradf, dtype=np.float64)
        if normalize:
            gradf = gradf / np.linalg.norm(gradf, axis=1, keepdims=True)
        v, f, nrm = self.vertlist, self.facelist, self.normals
        area = self.vertex_areas if per_vert_area is None else np.asarray(per_vert_area)
        v1, v2, v3 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        j1, j2, j3 = np.cross(nrm, v3 - v2) / 2, np.cross(nrm, v1 - v3) / 2, np.cross(nrm, v2 - v1) / 2
        rot = np.cross(nrm, gradf)
        rows = np.concatenate([f[:, 0], f[:, 1], f[:, 2]])
        cols = np.concatenate([f[:, 1], f[:, 2], f[:, 0]])
        d = lambda j: np.einsum("ij,ij->i", j, rot)
        sij = np.concatenate([d(j2), d(j3), d(j1)]) / 3.0
        sji = np.concatenate([d(j1), d(j2), d(j3)]) / 3.0
        n = self.n_vertices
        Wm = sparse.coo_matrix((np.concatenate([sij, sji, -sij, -sji]),
                                (np.concatenate([rows, cols, rows, cols]), np.concatenate([cols, rows, rows, cols]))), shape=(n, n)).tocsc()
        return sparse.diags(1.0 / area, shape=(n, n), format="csc") @ Wm

    @property
    def L(self):
        """A^-1 W (trimesh.py:482); built lazily, the matching path never needs it densified."""
        if self._L is None and self.W is not None:
            self._L = sparse.diags(1.0 / self.A.diagonal()).tocsr() @ self.W
        return self._L

    @L.setter
    def L(self, value):
        self._L = value

    # ------------------------------------------------------------- spectrum
    def _assemble_laplacian(self, robust=False):
        """W, A of this mesh (trimesh.py:440-482 -> laplacian.py:143-160); returns the lumped masses."""
        TriMesh._assemble_many([self], robust)
        return np.asarray(self.A.diagonal())

    @staticmethod
    def _assemble_many(meshes, robust=False):
        """Stiffness rows and lumped masses of several meshes.  Returns the dict MatchEngine.laplacian_ell produced (the operands of
        dm_eigenbasis, on the device, meshes padded to the largest) -- or None when the robust_laplacian wheel assembled the
        matrices on the host (the caller then hands W, A to the eigensolver as SciPy matrices)."""
        from ...engine import default_engine
        from . import laplacian as _lap
        for mesh in meshes:
            if mesh.facelist is None:
                raise NotImplementedError("point-cloud Laplacians are outside the matching path")
        if robust:
            try:
                import robust_laplacian                                       # trimesh.py:465-470
            except ImportError:                                               # (only the import itself: an ImportError from inside the wheel is the wheel's)
                robust_laplacian = None
            if robust_laplacian is not None:
                for mesh in meshes:
                    Wm, Am = robust_laplacian.mesh_laplacian(mesh.vertlist, mesh.facelist, mollify_factor=1e-5)
                    mesh.W = sparse.csr_matrix(Wm)
                    mass = np.asarray(Am.diagonal())
                    if np.any(mass <= 0):
                        raise ValueError("vertices with zero lumped mass (isolated vertices or degenerate faces): clean the mesh first")
                    mesh.A = sparse.diags(mass).tocsr()
                    mesh._L = None
                return None
            if _lap.robust_backend() != "restated":
                raise ImportError(
                    "process(robust=True) -- what FunctionalMapping.preprocess and compute_surface_map always ask for -- needs the "
                    "`robust_laplacian` package (what the reference calls, pyFM/mesh/trimesh.py:465-470), which is not installed.  This "
                    "package carries its own implementation of the same construction (tufted cover, intrinsic Delaunay flips, "
                    "mollification), NOT pinned against the wheel: opt in with compute_surface_map(..., robust_backend='restated'), "
                    "densematcher_amd.pyFM.mesh.laplacian.set_robust_backend('restated') or DENSEMATCHER_AMD_ROBUST_LAPLACIAN=restated, "
                    "or pass robust=False for the plain cotangent Laplacian")
            warnings.warn("robust=True: the robust_laplacian package is not installed; using this package's own tufted "
                          "intrinsic-Delaunay Laplacian (same construction, parity with the wheel unpinned)")
        eng = default_engine()
        if robust:
            covers = eng.tufted_covers([(m.vertlist, m.facelist) for m in meshes], mollify_factor=1e-5)
            bad = [i for i, c in enumerate(covers) if not c[3]]
            if bad:
                raise RuntimeError(f"intrinsic Delaunay flips did not converge for meshes {bad[:8]}")
            ell = eng.laplacian_ell([c[0] for c in covers], lens=[c[1] for c in covers], verts=[m.vertlist for m in meshes], scale=0.5)
        else:
            ell = eng.laplacian_ell([m.facelist for m in meshes], verts=[m.vertlist for m in meshes], scale=1.0)
        mass = ell["mass64"].cpu().numpy()
        for b, mesh in enumerate(meshes):
            n = mesh.n_vertices
            mesh._W = None
            mesh._W_dev = (ell["cols"][b, :n], ell["w"][b, :n])
            mesh._W_recipe = "robust" if robust else "cotangent"
            mesh._A = None
            mesh._mass = mass[b, :n].copy()
            mesh._L = None
        return ell

    def _n_eigs(self, k):
        return min(max(20, k), self.n_vertices - 1)                           # laplacian.py:165 computes at least 20 pairs

    def _store_spectrum(self, lam, phi, resid, k):
        """lam (>= k,), phi (N, >= k): device tensors of the solver; keeps the first k pairs (laplacian.py:165-167)"""
        if float(resid) > 1e-6 * max(1.0, float(lam[-1])):
            raise RuntimeError(f"eigensolver did not converge (residual {float(resid):.2e})")
        self.eigenvalues = np.array(lam[:k].cpu().numpy())
        self.eigenvectors = np.ascontiguousarray(phi[:self.n_vertices, :k].cpu().numpy())
        if abs(self.eigenvalues[0]) < 1e-9 * max(1.0, self.eigenvalues[-1]):
            self.eigenvalues[0] = max(self.eigenvalues[0], 0.0)

    def _has_spectrum(self, k):
        return (self.eigenvectors is not None) and (self.eigenvalues is not None) and (len(self.eigenvalues) >= k)

    # ------------------------------------------------------------- spectrum (host side input of the path)
    def laplacian_spectrum(self, k, intrinsic=False, return_spectrum=True, robust=False, verbose=False):
        """trimesh.py:440-496 -> laplacian.py:143-182.  W, A assembled on the device; the k smallest eigenpairs on the GPU."""
        ell = TriMesh._assemble_many([self], robust)
        if k > 0:
            kk = self._n_eigs(k)
            from ...engine import default_engine
            if ell is None:
                lam, phi, resid, _ = default_engine().eigenbasis([self.W], np.asarray(self.A.diagonal())[None], kk, tol=1e-10)
            else:
                lam, phi, resid, _ = default_engine().eigenbasis(None, None, kk, tol=1e-10, ell=ell)
            self._store_spectrum(lam[0], phi[0], resid[0], k)
            if return_spectrum:
                return self.eigenvalues, self.eigenvectors

    def process(self, k=200, skip_normals=True, intrinsic=False, robust=False, verbose=False):
        """trimesh.py:498-531: reuse a stored spectrum when it is large enough, else compute it."""
        if self._has_spectrum(k):
            self.eigenvectors = self.eigenvectors[:, :k]
            self.eigenvalues = self.eigenvalues[:k]
        else:
            self.laplacian_spectrum(k, return_spectrum=False, intrinsic=intrinsic, robust=robust, verbose=verbose)
        return self

    @staticmethod
    def process_many(meshes, ks, robust=False, verbose=False):
        """mesh.process(k) for several meshes (FunctionalMapping.preprocess: functional.py:300-301 processes its two meshes one
        after the other).  One device assembly and ONE batched eigensolve for the meshes that still need a spectrum (the
        eigensolver is a chain of small launches whose duration does not depend on how many meshes ride along; the smaller meshes
        are padded); each keeps its own k pairs."""
        todo = []
        for mesh, k in zip(meshes, ks):
            # (a subclass or a patched `process` -- a caller that supplies its own spectra -- keeps the say)
            if mesh._has_spectrum(k) or k <= 0 or type(mesh).process is not _TRIMESH_PROCESS:
                mesh.process(k, robust=robust, verbose=verbose)
            else:
                todo.append((mesh, k))
        if not todo:
            return meshes
        from ...engine import default_engine
        eng = default_engine()
        kk = max(mesh._n_eigs(k) for mesh, k in todo)
        nmin, nmax = min(m.n_vertices for m, _ in todo), max(m.n_vertices for m, _ in todo)
        # one batched call pads the smaller meshes with decoupled vertices at the top of their spectrum and solves all meshes for the
        # largest k; when that k (plus the solver's guard vectors) would leave the lower half of the smallest mesh's spectrum -- or
        # exceed it -- the meshes are solved one by one, each with its own k (small ones by the dense route)
        if len(todo) > 1 and (2 * (kk + 32) > nmin or kk > nmin - 1):
            for mesh, k in todo:
                mesh.process(k, robust=robust, verbose=verbose)
            return meshes
        ell = TriMesh._assemble_many([mesh for mesh, _ in todo], robust)
        if ell is None:
            lam, phi, resid, _ = eng.eigenbasis([mesh.W for mesh, _ in todo], [np.asarray(mesh.A.diagonal()) for mesh, _ in todo], kk, tol=1e-10)
        else:
            lam, phi, resid, _ = eng.eigenbasis(None, None, kk, tol=1e-10, ell=ell)
        lam, phi, resid = lam.cpu(), phi.cpu(), resid.cpu()                      # (one copy each for the batch, not three per mesh)
        for q, (mesh, k) in enumerate(todo):
            mesh._store_spectrum(lam[q], phi[q], resid[q], k)
        return meshes

    # ------------------------------------------------------------- vertex sampling (input of the subsampled ZoomOut)
    def extract_fps(self, size, random_init=True, geodesic=True, no_load=False, verbose=False, rng=None):
        """Farthest point sampling (trimesh.py:847-893 -> geometry.py:813-845): start at a random vertex, then repeatedly
        take the vertex farthest from the ones taken.  geodesic=False: Euclidean distances, the reference's arithmetic.
        geodesic=True: the reference measures with the heat method of the external potpourri3d wheel (`geod_from`); here the
        distance is the shortest path along mesh edges (Dijkstra on the edge graph) and a warning says so: the samples
        spread the same way, they are not the same vertices.  `rng`: numpy Generator for the start vertex (the reference
        draws from an unseeded one: its samples are not reproducible either).  Host code: it selects inputs of the path."""
        rng = np.random.default_rng() if rng is None else rng
        n = self.n_vertices
        if not geodesic:
            def dist_from(i):
                return np.linalg.norm(self.vertlist - self.vertlist[i, None, :], axis=1)
        else:
            import scipy.sparse.csgraph as csgraph
            warnings.warn("extract_fps(geodesic=True): potpourri3d's heat-method geodesics are not available; using shortest paths "
                          "along the mesh edges")
            f = self.facelist
            e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
            w = np.linalg.norm(self.vertlist[e[:, 0]] - self.vertlist[e[:, 1]], axis=1)
            G = sparse.coo_matrix((w, (e[:, 0], e[:, 1])), shape=(n, n)).tocsr()
            G = G.maximum(G.T)

            def dist_from(i):
                return csgraph.dijkstra(G, directed=False, indices=i)
        inds = [int(rng.integers(n))]                                           # geometry.py:833
        dists = dist_from(inds[0])
        for _ in range(size - 1):                                               # geometry.py:838-843
            newid = int(np.argmax(dists))
            inds.append(newid)
            dists = np.minimum(dists, dist_from(newid))
        return np.asarray(inds)

    # ------------------------------------------------------------- spectral helpers
    def project(self, func, k=None):
        """Phi[:, :k]^T (A func) (trimesh.py:533-556), on the GPU."""
        if k is not None and k > self.eigenvectors.shape[1]:
            raise ValueError(f'At least {k} eigenvectors should be computed before projecting')
        from ...engine import default_engine
        func = np.asarray(func)
        one_d = func.ndim == 1
        F = (func[:, None] if one_d else func).astype(np.float32)
        kk = self.eigenvectors.shape[1] if k is None else k
        # float64 eigenvectors / masses go in unrounded (dm_project_f64, float64 matrix cores); the result is fp32
        out = default_engine().project(np.ascontiguousarray(self.eigenvectors)[None], np.ascontiguousarray(self.A.diagonal())[None],
                                       F[None], kk, exact=True)[0].cpu().numpy().astype(np.float64)
        return out[:, 0] if one_d else out

    def decode(self, projection):
        """Phi[:, :k] @ projection (trimesh.py:558-577); a trivially small host product."""
        k = projection.shape[0]
        if k <= self.eigenvectors.shape[1]:
            return self.eigenvectors[:, :k] @ projection
        raise ValueError(f'At least {k} eigenvectors should be computed before decoding')

    def l2_sqnorm(self, func):
        return self.l2_inner(func, func)

    def l2_inner(self, func1, func2):
        return np.einsum('np,np->p', func1, self.A @ func2) if func1.ndim > 1 else func1 @ (self.A @ func2)

    def integrate(self, func):
        return func.T @ self.A.diagonal()


_TRIMESH_PROCESS = TriMesh.process

"""
FunctionalMapping with the reference's interface (densematcher/pyFM/functional.py:19), restricted to what the
matching hot path uses.  Arithmetic: libdensematch (HIP) through densematcher_amd.engine.
"""
import copy

import numpy as np

from . import refine, signatures as sg, spectral

_OPTIMIZERS = ("L-BFGS-B", "fmin_l_bfgs_b", "l-bfgs-b")
# L-BFGS stopping rule of the iterative fit.  The reference passes only maxiter (SciPy defaults ftol = 2.2e-9, gtol = 1e-5,
# maxcor = 10) and evaluates in float32, which stops 1e-4 .. 1e-3 short of the minimiser.  The float64
# evaluation here makes a tight rule meaningful: fit(..., stopping="tight") runs the options below (the 1e-4 parity bar on C against
# the float64 minimiser is tested with it); the DEFAULT is stopping="reference", SciPy's default rule, because a drop-in should end
# where the code it replaces ends: against the reference's own 14-tuple the ICP slots agree 0.97 / 0.98 with it, 0.92 / 0.90 tight.
# ftol = 1e-12: the energy is flat around its minimiser, and WHEN a relative decrease of a few machine epsilons is first seen
# is decided by rounding noise -- on the notebook's fit the same rule took 289, 322, 351, 440 or 448 evaluations at 1e-13 (740
# or 1690 at 1e-15) as summation orders inside the evaluation changed, for maps that agree to 2e-5.  Measured on that fit
# (tools/fit_profile.py <ftol>; distance of C from the 1e-15 result): 1e-14 464 evaluations 8e-7, 1e-13 448 1e-6,
# 1e-12 332 9e-6, 1e-11 261 4e-5, 1e-10 224 1.3e-4.  1e-12 keeps a factor ten under the 1e-4 bar on C.
CLOSED_FORM_MAX_K1 = 200          # dm_fmap_solve / dm_fmap_fit: the in-LDS solvers take systems of order <= 199
LBFGS_OPTIONS = {"ftol": 1e-12, "gtol": 1e-9, "maxcor": 30, "maxfun": 15000}
# maps wider than the closed form takes (quadratic energy, k^2 > 40 000 unknowns): run until the gradient test or the line search's noise floor
LBFGS_WIDE = {"ftol": 1e-15, "gtol": 1e-11, "maxcor": 30, "maxfun": 50000}


class FunctionalMapping:
    def __init__(self, mesh1, mesh2, partial=False, optimizer="fmin_l_bfgs_b"):
        self.mesh1 = copy.deepcopy(mesh1)          # functional.py:58-59
        self.mesh2 = copy.deepcopy(mesh2)
        self.descr1 = None
        self.descr2 = None
        self._FM_type = 'classic'
        self._FM_base = None
        self._FM_icp = None
        self._FM_zo = None
        self._k1, self._k2 = None, None
        self.optimizer = optimizer
        self.partial = partial
        self.eta = None
        self.mapped_indicator = None

    # ---------------------------------------------------------------- dimensions / state (functional.py:79-199)
    @property
    def k1(self):
        if self._k1 is None and not self.preprocessed and not self.fitted:
            raise ValueError('No information known about dimensions')
        return self.FM.shape[1] if self.fitted else self._k1

    @k1.setter
    def k1(self, k1):
        self._k1 = k1

    @property
    def k2(self):
        if self._k2 is None and not self.preprocessed and not self.fitted:
            raise ValueError('No information known about dimensions')
        return self.FM.shape[0] if self.fitted else self._k2

    @k2.setter
    def k2(self, k2):
        self._k2 = k2

    @property
    def FM_type(self):
        return self._FM_type

    @FM_type.setter
    def FM_type(self, FM_type):
        if FM_type.lower() not in ['classic', 'icp', 'zoomout']:
            raise ValueError(f'FM_type can only be set to "classic", "icp" or "zoomout", not {FM_type}')
        self._FM_type = FM_type

    def change_FM_type(self, FM_type):
        self.FM_type = FM_type

    @property
    def FM(self):
        return {'classic': self._FM_base, 'icp': self._FM_icp, 'zoomout': self._FM_zo}[self.FM_type.lower()]

    @FM.setter
    def FM(self, FM):
        self._FM_base = FM

    @property
    def preprocessed(self):
        test_descr = (self.descr1 is not None) and (self.descr2 is not None)
        test_evals = (self.mesh1.eigenvalues is not None) and (self.mesh2.eigenvalues is not None)
        test_evects = (self.mesh1.eigenvectors is not None) and (self.mesh2.eigenvectors is not None)
        return test_descr and test_evals and test_evects

    @property
    def fitted(self):
        return self.FM is not None

    def _get_lmks(self, landmarks, verbose=False):
        # (p,) / (p,1): the same vertex indices on both meshes; (p,2): one column per mesh      functional.py:253-262
        lm = np.asarray(landmarks)
        if lm.squeeze().ndim == 1:
            return lm.squeeze(), lm.squeeze().copy()
        return lm[:, 0], lm[:, 1]

    # ---------------------------------------------------------------- preprocess (functional.py:264-350)
    def preprocess(self, n_ev=(50, 50), n_descr=100, descr_type='WKS', landmarks=None, subsample_step=1, k_process=None,
                   verbose=False, descr1=None, descr2=None):
        self.k1, self.k2 = n_ev
        if k_process is None:
            k_process = 1
        use_lm = landmarks is not None and len(landmarks) > 0
        # (functional.py:300-301: mesh1.process, mesh2.process; here the two eigensolves share one batched call)
        ks = [max(self.k1, k_process), max(self.k2, k_process)]
        if all(hasattr(m, "_assemble_laplacian") for m in (self.mesh1, self.mesh2)):
            type(self.mesh1).process_many([self.mesh1, self.mesh2], ks, robust=True, verbose=verbose)
        else:                                                                    # (duck-typed meshes bring their own process)
            self.mesh1.process(ks[0], verbose=verbose, robust=True, intrinsic=False)
            self.mesh2.process(ks[1], verbose=verbose, robust=True, intrinsic=False)
        if use_lm:
            lmks1, lmks2 = self._get_lmks(landmarks)
        if descr1 is not None and descr2 is not None:
            self.descr1, self.descr2 = descr1, descr2
        elif descr_type in ('HKS', 'WKS'):                                       # functional.py:308-329
            sig = sg.mesh_HKS if descr_type == 'HKS' else sg.mesh_WKS
            self.descr1 = sig(self.mesh1, n_descr, k=self.k1)                    # (N1, n_descr)
            self.descr2 = sig(self.mesh2, n_descr, k=self.k2)
            if use_lm:
                self.descr1 = np.hstack([self.descr1, sig(self.mesh1, n_descr, landmarks=lmks1, k=self.k1)])
                self.descr2 = np.hstack([self.descr2, sig(self.mesh2, n_descr, landmarks=lmks2, k=self.k2)])
        else:
            raise ValueError(f'Descriptor type "{descr_type}" not implemented')
        if subsample_step != 1:                                                  # (step 1 selects every column: the arrays as they are)
            self.descr1 = self.descr1[:, np.arange(0, self.descr1.shape[1], subsample_step)]      # functional.py:333-334
            self.descr2 = self.descr2[:, np.arange(0, self.descr2.shape[1], subsample_step)]
        return self                                                                          # no normalisation (:336-344)

    # ---------------------------------------------------------------- fit (functional.py:352-487)
    def fit(self, w_descr=1e-1, w_lap=1e-3, w_dcomm=1, w_orient=0, w_area=0, w_conformal=0, w_p2p=0, w_stochastic=0, w_ent=0,
            w_range01=0, w_sumto1=0, w_area_difference=0, w_mumford_shah=0, mumford_shah_var=0.1, w_eta_entropy=0,
            orient_reversing=False, optinit='zeros', verbose=False, maxiter=1000000, device=None, stopping="reference", driver="device"):
        """reference functional.py:352-487.  With only w_descr / w_lap > 0 the minimiser (what the reference's L-BFGS-B
        converges to, first column pinned) is obtained in closed form on the GPU.  With any of
        w_dcomm, w_orient, w_area, w_conformal, w_p2p, w_stochastic, w_ent, w_range01, w_sumto1 > 0 the reference's own scheme
        runs: limited-memory BFGS with L-BFGS-B's line search and stopping tests (scipy.optimize.minimize, :477) from
        get_x0(optinit), on the device, energy and gradient in float64 (maps up to 32 x 32 with the notebook's kind of terms:
        dm_fmap_fit_fused, one launch per evaluation; everything else: dm_fmap_fit_steps -> dm_fmap_energy_grad + dm_lbfgs_advance).
        Only the area-difference, Mumford-Shah and eta-entropy terms are not on the path (NotImplementedError).
        stopping = "reference" (default): SciPy's default rule, i.e. what the reference's call runs with (ftol 2.2e-9, gtol 1e-5):
        the fit ends where the reference's ends, a few 1e-4 short of the minimiser, and the drop-in agrees best with the
        reference's own outputs (INTEGRATION.md has the per-slot table); "tight": ftol 1e-12, the float64 minimiser to 1e-5."""
        from ..engine import default_engine
        if optinit not in ['random', 'identity', 'zeros']:
            raise ValueError(f"optinit arg should be 'random', 'identity' or 'zeros', not {optinit}")
        if self.optimizer not in _OPTIMIZERS:
            raise ValueError(f"Unknown solver {self.optimizer}")
        if self.partial:
            raise NotImplementedError()                                        # functional.py:480
        off_path = dict(w_area_difference=w_area_difference, w_mumford_shah=w_mumford_shah, w_eta_entropy=w_eta_entropy)
        live = [n for n, v in off_path.items() if v > 0]
        if live:
            raise NotImplementedError(f"energy terms {live} are not on the accelerated path; pass 0")
        general = dict(w_dcomm=w_dcomm, w_p2p=w_p2p, w_stochastic=w_stochastic, w_ent=w_ent, w_range01=w_range01, w_sumto1=w_sumto1,
                       w_orient=w_orient, w_area=w_area, w_conformal=w_conformal)
        iterative = any(v > 0 for v in general.values())
        if not (w_descr > 0 or w_lap > 0 or iterative):
            raise ValueError("every energy weight is 0")                       # base_functions.py:534,639 would fail too
        # The closed form solves k2 systems of order k1 - 1 in on-chip memory: maps up to 200 columns.  Wider maps (the reference has no
        # cap, functional.py:352) take the reference's own scheme instead -- L-BFGS on the two quadratic terms, on the device, float64,
        # run to ftol 1e-12 (the float64 minimiser to 1e-5) unless the caller asked for the reference's stopping rule explicitly.
        if not self.preprocessed:
            self.preprocess()
        wide = (not iterative) and self.mesh1.eigenvectors.shape[1] > CLOSED_FORM_MAX_K1
        eng = default_engine()
        m1, m2 = self.mesh1, self.mesh2
        # like the reference (functional.py:412-413) fit uses every stored eigenvector column
        Phi1 = np.ascontiguousarray(m1.eigenvectors, dtype=np.float32)[None]
        Phi2 = np.ascontiguousarray(m2.eigenvectors, dtype=np.float32)[None]
        a1 = np.ascontiguousarray(m1.A.diagonal(), dtype=np.float32)[None]
        a2 = np.ascontiguousarray(m2.A.diagonal(), dtype=np.float32)[None]
        d1, d2 = np.asarray(self.descr1), np.asarray(self.descr2)
        fdt = np.float16 if (d1.dtype == np.float16 and d2.dtype == np.float16) else np.float32
        batch = {"Phi1": Phi1, "Phi2": Phi2, "a1": a1, "a2": a2,
                 "lam1": np.asarray(m1.eigenvalues, dtype=np.float64)[None], "lam2": np.asarray(m2.eigenvalues, dtype=np.float64)[None],
                 "F1": np.ascontiguousarray(d1, dtype=fdt)[None], "F2": np.ascontiguousarray(d2, dtype=fdt)[None]}
        dev = {n: eng._dev(v, {np.float16: __import__("torch").float16, np.float32: __import__("torch").float32,
                               np.float64: __import__("torch").float64}[v.dtype.type], n) for n, v in batch.items()}
        if iterative or wide:
            weights = dict(w_descr=w_descr, w_lap=w_lap, **general)
            if wide:
                stopping = "tight"
            if stopping not in ("tight", "reference"):
                raise ValueError("stopping must be 'tight' or 'reference'")
            x0 = self.get_x0(optinit=optinit)
            orient_ops = None
            if w_orient > 0:
                # functional.py:432-456: the orientation operators, and the weight rescaled by (energy of the other terms at x0) /
                # (orientation energy at x0).  The reference rescales with the NumPy operators of compute_orientation_op
                # (reversing honoured there) and then OPTIMISES with the operators energy_func_std rebuilds itself
                # (base_functions.py:567-597: rows divided by diag(A), never reversed) -- both restated as they are.
                resc = self.compute_orientation_op(reversing=orient_reversing)
                o1 = np.stack([a for a, _ in resc])[None]
                o2 = np.stack([b for _, b in resc])[None]
                w_native = dict(weights, w_orient=0.0)
                e_native = eng.fit_energy(dev, w_native, x0[None])
                e_orient = eng.fit_energy(dev, dict(w_orient=1.0), x0[None], orient_ops=(o1, o2))
                w_orient = w_orient * float(e_native[0]) / float(e_orient[0])
                weights["w_orient"] = w_orient
                self.w_orient_rescaled = w_orient
                fit_ops = self.compute_orientation_op(reversing=False, area="mass")
                orient_ops = (np.stack([a for a, _ in fit_ops])[None], np.stack([b for _, b in fit_ops])[None])
            self._verbose_terms(eng, dev, weights, x0, orient_ops, "x0")
            C, res = eng.fit_general(dev, weights, x0[None], maxiter=maxiter,
                                     lbfgs_options=(LBFGS_WIDE if wide else LBFGS_OPTIONS) if stopping == "tight" else None, driver=driver,
                                     orient_ops=orient_ops)
            self.FM = np.asarray(C[0], dtype=np.float64)
            self.fit_result = res
            self._verbose_terms(eng, dev, weights, self.FM, orient_ops, "solution")
            if verbose:
                print(f"\tTask funcall : {res.nfev}, nit : {res.nit}, warnflag : {res.message}")
        else:
            A = eng.project(dev["Phi1"], dev["a1"], dev["F1"])
            B = eng.project(dev["Phi2"], dev["a2"], dev["F2"])
            # pinned entry from the float64 spectrum and masses (get_x0 is float64 host code in the reference, :654-658)
            c00 = eng.c00(np.ascontiguousarray(m1.eigenvectors)[None], np.ascontiguousarray(m2.eigenvectors)[None],
                        np.ascontiguousarray(m1.A.diagonal())[None], np.ascontiguousarray(m2.A.diagonal())[None])
            C = eng.fmap_solve(A, B, dev["lam1"], dev["lam2"], c00, w_descr, w_lap, check=True)
            self.FM = C[0].cpu().numpy()
        self.eta = np.ones(m2.eigenvectors.shape[0])                           # functional.py:483
        self._dev = dev

    # the reference's per-term printout (base_functions.py:27-29, 538-636: `VERBOSE` in the environment prints every live term's
    # weighted loss at every energy evaluation).  The optimiser runs on the device here, so the same lines are printed where the
    # host sees the map: at the start point and at the solution.
    _VERBOSE_LABELS = (("w_descr", "descr loss:"), ("w_lap", "lap loss:"), ("w_dcomm", "descr comm loss:"), ("w_orient", "orient loss:"),
                       ("w_area", "area loss:"), ("w_conformal", "conformal loss:"), ("w_p2p", "p2p loss:"),
                       ("w_stochastic", "stochastic loss:"), ("w_ent", "entropy loss:"), ("w_range01", "range01 loss:"),
                       ("w_sumto1", "sumto1 loss:"))

    def _verbose_terms(self, eng, dev, weights, x, orient_ops, where):
        import os
        if not os.environ.get("VERBOSE", False):
            return
        print(f"energy terms at the {where}:")
        for name, label in self._VERBOSE_LABELS:
            w = weights.get(name, 0)
            if w > 0:
                e = eng.fit_energy(dev, {name: w}, np.asarray(x)[None], orient_ops=orient_ops if name == "w_orient" else None)
                print(label, float(e[0]))

    def compute_orientation_op(self, reversing=False, normalize=False, area="vertex"):
        """functional.py:686-728: per descriptor the pair (pinv1 O1 Phi1, +-pinv2 O2 Phi2) of orientation operators in the reduced
        bases, O = TriMesh.orientation_op(gradient of the descriptor).  area = "vertex": rows divided by the mesh's vertex_areas (the
        reference's method); "mass": by diag(A) (what energy_func_std builds, base_functions.py:573).  Host arithmetic (sparse
        products per descriptor), as in the reference."""
        out = []
        sides = []
        for mesh, descr, k in ((self.mesh1, self.descr1, self.k1), (self.mesh2, self.descr2, self.k2)):
            ev = np.asarray(mesh.eigenvectors[:, :k], dtype=np.float64)
            pinv = ev.T @ mesh.A
            pva = None if area == "vertex" else np.asarray(mesh.A.diagonal())
            d = np.asarray(descr, dtype=np.float64)
            sides.append([np.asarray(pinv @ (mesh.orientation_op(mesh.gradient(d[:, i], normalize=normalize), per_vert_area=pva) @ ev))
                          for i in range(d.shape[1])])
        for a, b in zip(*sides):
            out.append((a, -b if reversing else b))
        return out

    def get_x0(self, optinit="zeros"):
        """functional.py:629-660"""
        if optinit == 'random':
            x0 = np.random.random((self.k2, self.k1))
            x0 = x0 / x0.sum()
        elif optinit == 'identity':
            x0 = np.eye(self.k2, self.k1)
        else:
            x0 = np.zeros((self.k2, self.k1))
        ev_sign = np.sign(self.mesh1.eigenvectors[0, 0] * self.mesh2.eigenvectors[0, 0])
        area_ratio = np.sqrt(self.mesh2.area / self.mesh1.area)
        x0[:, 0] = np.zeros(self.k2)
        x0[0, 0] = ev_sign * area_ratio
        return x0

    # ---------------------------------------------------------------- maps and refinement
    def get_p2p(self, use_adj=False, n_jobs=1):
        """functional.py:201-219: returns the kd-tree maps (p2p_21, p2p_12) and sets self.mapped_indicator"""
        p2p_21, p2p_12, self.mapped_indicator = spectral.mesh_FM_to_p2p(self.FM, self.mesh1, self.mesh2, use_adj=use_adj,
                                                                       n_jobs=n_jobs)
        return p2p_21, p2p_12

    def get_precise_map(self, precompute_dmin=True, use_adj=True, batch_size=None, n_jobs=1, verbose=False):
        """functional.py:221-251: (n2, n1) sparse precise map from mesh2 to mesh1"""
        if not self.fitted:
            raise ValueError('Model should be fit and fit to obtain p2p map')
        return spectral.mesh_FM_to_p2p_precise(self.FM, self.mesh1, self.mesh2, precompute_dmin=precompute_dmin, use_adj=use_adj,
                                               batch_size=batch_size, n_jobs=n_jobs, verbose=verbose)

    def _precise_map_device(self):
        """get_precise_map().toarray() kept on the GPU (compute_surface_map feeds it to the assignment kernel)"""
        from ..engine import default_engine
        k2, k1 = self.FM.shape
        from .spectral.convert import _basis, _real_dtype
        dt = _real_dtype(self.mesh1.eigenvectors, self.mesh2.eigenvectors)
        return default_engine().precise_map(_basis(self.mesh1.eigenvectors, k1, dt), _basis(self.mesh2.eigenvectors, k2, dt),
                                            np.asarray(self.FM, dtype=np.float64)[None],
                                            np.ascontiguousarray(self.mesh1.facelist, dtype=np.int32)[None], dense=True)[2][0]

    def icp_refine(self, nit=10, tol=None, use_adj=False, overwrite=True, verbose=False, n_jobs=1):
        """functional.py:564-586"""
        if not self.fitted:
            raise ValueError("The Functional map must be fit before refining it")
        self._FM_icp = refine.mesh_icp_refine(self.FM, self.mesh1, self.mesh2, nit=nit, tol=tol, return_p2p=False,
                                              use_adj=use_adj, n_jobs=n_jobs, verbose=verbose)
        if overwrite:
            self.FM_type = 'icp'

    def zoomout_refine(self, nit=10, step=1, subsample=None, overwrite=True, verbose=False):
        """functional.py:588-617"""
        if not self.fitted:
            raise ValueError("The Functional map must be fit before refining it")
        if subsample is None or (np.issubdtype(type(subsample), np.integer) and subsample == 0):   # functional.py:607-610
            sub = None
        else:
            sub = subsample                                                     # int: farthest point sampling of that size; or (sub1, sub2)
        self._FM_zo = refine.mesh_zoomout_refine(self.FM, self.mesh1, self.mesh2, nit, step=step, subsample=sub, verbose=verbose)
        if overwrite:
            self.FM_type = 'zoomout'

    # ---------------------------------------------------------------- small helpers (functional.py:730-831)
    def project(self, func, k=None, mesh_ind=1):
        if mesh_ind == 1:
            return self.mesh1.project(func, k=self.k1 if k is None else k)
        elif mesh_ind == 2:
            return self.mesh2.project(func, k=self.k2 if k is None else k)
        raise ValueError(f'Only indices 1 or 2 are accepted, not {mesh_ind}')

    def decode(self, encoded_func, mesh_ind=2):
        if mesh_ind == 1:
            return self.mesh1.decode(encoded_func)
        elif mesh_ind == 2:
            return self.mesh2.decode(encoded_func)
        raise ValueError(f'Only indices 1 or 2 are accepted, not {mesh_ind}')

    def transport(self, encoded_func, reverse=False):
        if not self.preprocessed:
            raise ValueError("The Functional map must be fit before transporting a function")
        return np.linalg.pinv(self.FM) @ encoded_func if reverse else self.FM @ encoded_func

    def transfer(self, func, reverse=False):
        if not reverse:
            return self.decode(self.transport(self.project(func)))
        return self.decode(self.transport(self.project(func, mesh_ind=2), reverse=True), mesh_ind=1)

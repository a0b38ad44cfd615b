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


def _assign_early(matrix, slot):
    """_assign started NOW on a side stream (its own MatchEngine), while the caller goes on with the main stream: returns the
    function that waits for it and hands back (row_ind, col_ind).  compute_surface_map's three assignments do not depend on each
    other (functional_map.py:57, 66, 78), and the longest -- the fitted map's indicator, 20 ms on one workgroup -- can start as soon
    as the fit is done, beside the precise map, ICP and the vertex maps of the refined map."""
    import torch
    from .engine import default_engine
    dev_index = torch.cuda.current_device()
    main = torch.cuda.current_stream(dev_index)
    side = _side_streams(dev_index, slot + 1)[slot]
    dev = matrix.device_tensor() if hasattr(matrix, "device_tensor") else matrix       # (formed on the main stream)
    if dev.dim() == 2:
        dev = dev[None]
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dev.record_stream(side)
        done = default_engine().linear_sum_assignment(dev, maximize=True, defer=True)

    def finish():
        with torch.cuda.stream(side):
            col = done()[0].cpu().numpy().astype(np.int64)
        rows = np.nonzero(col >= 0)[0]
        return rows, col[rows]
    return finish


EARLY_ASSIGNMENTS = True          # False: compute_surface_map runs its three assignments as one batched call at the end (bench.py's per-stage timing)


_ASSIGN_STACK_LIMIT = 16 << 30     # bytes of dense matrices copied side by side for one assignment launch (288 GB of HBM on the part)


def _assign_many(matrices):
    """the assignments of several (n2, n1) matrices in ONE batched call: one workgroup per matrix, so three assignments take
    the time of the longest instead of their sum.  Mapped indicators of small maps (k <= 32) go in by their FACTORS
    (dm_lsa_indicator: the kernel evaluates their rows itself, the dense matrices are not formed); everything else dense."""
    import torch
    from .engine import default_engine
    eng = default_engine()
    # From the factors a search step costs 15 fused multiply-adds per column where the dense row costs one load: measured on one pair,
    # 24 ms against 15.5 ms for the fitted map's indicator (tools/lsa_batch_spread.py) -- so a few matrices go in dense, and the factor
    # form serves where the dense copies would not fit (and compute_surface_map_batch, whose launch lasts as long as its slowest
    # matrix either way: 91 ms against 93 for 192 matrices, without 4 GB of indicators).
    devs_bytes = sum(int(np.prod(m.shape)) * 8 for m in matrices)
    ind = [q for q, m in enumerate(matrices) if hasattr(m, "_args") and getattr(m, "_dev", None) is None] if devs_bytes > _ASSIGN_STACK_LIMIT else []
    if ind:
        shapes = {(tuple(matrices[q]._args[0].shape[1:]), tuple(matrices[q]._args[1].shape[1:]), tuple(matrices[q]._args[3].shape[1:]),
                   str(matrices[q]._args[0].dtype), str(getattr(matrices[q]._args[2], "dtype", None))) for q in ind}
        P1, P2, _, C0 = matrices[ind[0]]._args
        if len(shapes) != 1 or not eng.lsa_indicator_ok(P1.shape[1], P2.shape[1], C0.shape[2], C0.shape[1]):
            ind = []
    if ind:
        rest = [q for q in range(len(matrices)) if q not in ind]
        dev = [matrices[q].device_tensor() if hasattr(matrices[q], "device_tensor") else matrices[q] for q in rest]
        dev = [d if d.dim() == 2 else d[0] for d in dev]
        cat = lambda z: torch.cat([torch.as_tensor(matrices[q]._args[z]).to(eng.device) for q in ind], dim=0)      # (masses may arrive as NumPy arrays)
        col = eng.lsa_indicator(cat(0), cat(1), cat(2), cat(3), dense=torch.stack(dev) if dev else None, maximize=True).cpu().numpy().astype(np.int64)
        order = ind + rest
        col = col[np.argsort(order)]
    else:
        devs = [m.device_tensor() if hasattr(m, "device_tensor") else m for m in matrices]
        devs = [d if d.dim() == 2 else d[0] for d in devs]
        # torch.stack copies every matrix next to its original (which the caller's MappedIndicator keeps alive): beyond the limit
        # the matrices are assigned one after the other, as the reference does
        if len(devs) > 1 and sum(d.numel() * d.element_size() for d in devs) > _ASSIGN_STACK_LIMIT:
            col = np.stack([eng.linear_sum_assignment(d[None], maximize=True)[0].cpu().numpy() for d in devs]).astype(np.int64)
        else:
            col = eng.linear_sum_assignment(torch.stack(devs), maximize=True).cpu().numpy().astype(np.int64)
    out = []
    for c in col:
        rows = np.nonzero(c >= 0)[0]
        out.append((rows, c[rows]))
    return out


class _robust_backend_for_call:
    """robust_backend='restated' | 'wheel' for the duration of one call (None: whatever laplacian.set_robust_backend / the environment say)"""
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        from .pyFM.mesh import laplacian
        self.prev = laplacian.robust_backend()
        if self.name is not None:
            laplacian.set_robust_backend(self.name)

    def __exit__(self, *exc):
        from .pyFM.mesh import laplacian
        laplacian.set_robust_backend(self.prev)
        return False


def compute_surface_map(mesh1_t, mesh2_t, c1, c2, n_ev=50, compute_extra=False, optimizer="fmin_l_bfgs_b", descr_type="neural",
                        maxiter=100000, optimize_p2p=False, fit_params=None, robust_backend=None):
    '''
    robust_backend (not in the reference): the Laplacians are built with robust=True like the reference's (functional.py:294-295), which
        calls the `robust_laplacian` wheel.  Where that wheel is not installed the call raises ImportError -- unless
        robust_backend="restated" (or laplacian.set_robust_backend / DENSEMATCHER_AMD_ROBUST_LAPLACIAN) opts into this package's own
        implementation of the same construction, whose parity with the wheel is unpinned.
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
    with _robust_backend_for_call(robust_backend):
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
    # three depends on another one's result, and a matrix's shortest-augmenting-path search is sequential (one workgroup, tens of
    # milliseconds): each starts on its own stream as soon as its matrix exists, beside whatever the main stream does next.
    start_s = time.time()
    early = None
    if compute_extra:
        # (the two assignments that are ready now start on side streams; TIMEIT keeps the reference's printed stages whole)
        overlap = EARLY_ASSIGNMENTS and not timing
        hung_plain = _assign_early(model.mapped_indicator, 0) if overlap else None         # functional_map.py:57
        precise = model._precise_map_device()                                   # functional_map.py:62 (get_precise_map().toarray())
        hung_prec = _assign_early(precise, 1) if overlap else None              # functional_map.py:66
        early = (hung_plain, hung_prec) if overlap else None
        ind_plain = None if overlap else model.mapped_indicator
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
    if compute_extra and early is not None:
        hungarian_icp = _assign_many([model.mapped_indicator])[0]               # functional_map.py:78
        hungarian, hungarian_precise = early[0](), early[1]()
    elif compute_extra:
        hungarian, hungarian_precise, hungarian_icp = _assign_many([ind_plain, precise, model.mapped_indicator])   # :57, :66, :78
    else:
        hungarian, hungarian_precise = None, None
        hungarian_icp = _assign_many([model.mapped_indicator])[0]               # functional_map.py:78
    if timing:
        print("Hungarian for vanilla, precise and icp took", time.time() - start_s, "seconds")
    model.mesh1.release_device_rows(); model.mesh2.release_device_rows()       # (views of the assembly's batch arrays: not kept alive by a result)
    return (p2p_21, p2p_12, hungarian, hungarian_precise, p2p_21_icp, p2p_12_icp, hungarian_icp, model, model.mesh1, model.mesh2,
            p2p_21_adjoint, p2p_12_adjoint, p2p_21_icp_adjoint, p2p_12_icp_adjoint)


def _on_side_stream(side, tensors, launch):
    """`launch()` (a deferred MatchEngine call) on the stream `side`, which first waits for the current stream; `tensors` (made on the
    current stream) are marked as in use there.  Returns what launch returned (the function that waits and hands back the result)."""
    import torch
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for t_ in tensors:
            if t_ is not None:
                t_.record_stream(side)
        return launch()


def _batch_chunk(models, idx, out, n_ev, compute_extra, fit_params, after_eigenbases=None, lsa_streams=None):
    """the pairs `idx` of a batch on the CURRENT stream's engine: eigenbases of their 2 len(idx) meshes in one batched solve, then per group
    of equal sizes one device L-BFGS, one call per map stage, all assignments of the group side by side"""
    import torch
    from .engine import default_engine
    from .pyFM.spectral.convert import MappedIndicator, _real_dtype
    eng = default_engine()
    groups = {}
    for i in idx:
        model = models[i]
        key = (model.mesh1.n_vertices, model.mesh2.n_vertices, model.mesh1.facelist.shape[0], model.descr1.shape[1],
               str(model.descr1.dtype), str(model.descr2.dtype))
        groups.setdefault(key, []).append(i)
    # The descriptors do not wait for the eigenbases: a helper thread stacks them (64 MiB per side for 32 pairs of 2048 x 512 fp16)
    # and uploads them on a stream of its own while this thread drives the eigensolver -- 28 ms of a 64-pair call that the first
    # chunk otherwise spends between its eigensolve and its fit with the GPU idle (rocprofv3 timeline, tools/batch_timeline.py).
    import threading
    tdt = {np.float16: torch.float16, np.float32: torch.float32, np.float64: torch.float64}
    main_stream = torch.cuda.current_stream()
    up_stream = _upload_stream(main_stream)
    staged, stage_err = {}, []

    def stage_descriptors():
        try:
            with torch.cuda.stream(up_stream):
                for gno_, (key_, gidx_) in enumerate(groups.items()):
                    g_ = [models[i] for i in gidx_]
                    fdt_ = np.float16 if (g_[0].descr1.dtype == np.float16 and g_[0].descr2.dtype == np.float16) else np.float32
                    pair = []
                    for side in ("descr1", "descr2"):
                        # (into a page-locked buffer kept from call to call: a fresh 64 MiB array per call is 16 k page faults, 20-50 ms
                        #  of host time that also stalls the other threads' allocations -- the calls alternated between 310 and 375 ms)
                        first = np.asarray(getattr(g_[0], side))
                        host = _pinned_buffer((main_stream.cuda_stream, side, key_), (len(g_),) + first.shape, tdt[fdt_])
                        hv = host.numpy()
                        for q_, m in enumerate(g_):
                            hv[q_] = getattr(m, side)            # (converts to fdt_ where the caller's dtype differs)
                        pair.append(eng.scratch("%s_%d" % (side, gno_), tuple(host.shape), tdt[fdt_]).copy_(host, non_blocking=True))
                    staged[key_] = (fdt_, pair[0], pair[1])
                up_stream.synchronize()
        except BaseException as e:                      # (re-raised by the chunk's thread)
            stage_err.append(e)
    helper = threading.Thread(target=stage_descriptors)
    helper.start()
    try:
        # ---- eigenbases: every mesh of the chunk in one batched solve (FunctionalMapping.preprocess: functional.py:300-301)
        all_meshes = [m for i in idx for m in (models[i].mesh1, models[i].mesh2)]
        type(all_meshes[0]).process_many(all_meshes, [n_ev] * len(all_meshes), robust=True)
    finally:
        if after_eigenbases is not None:
            after_eigenbases()
        helper.join()
    if stage_err:
        raise stage_err[0]
    for key, gidx in groups.items():
        g = [models[i] for i in gidx]
        nb = len(g)
        rdt = _real_dtype(g[0].mesh1.eigenvectors, g[0].mesh2.eigenvectors)
        st = lambda f, dt: np.ascontiguousarray(np.stack([f(m) for m in g]), dtype=dt)
        Phi1, Phi2 = st(lambda m: m.mesh1.eigenvectors[:, :n_ev], rdt), st(lambda m: m.mesh2.eigenvectors[:, :n_ev], rdt)
        a1, a2 = st(lambda m: m.mesh1.vertex_masses, rdt), st(lambda m: m.mesh2.vertex_masses, rdt)
        lam1, lam2 = st(lambda m: m.mesh1.eigenvalues[:n_ev], np.float64), st(lambda m: m.mesh2.eigenvalues[:n_ev], np.float64)
        fdt, F1d, F2d = staged[key]
        F1d.record_stream(main_stream); F2d.record_stream(main_stream)
        # ---- fit (FunctionalMapping.fit: the fp32 view of the bases like the reference's fit, functional.py:412-413)
        dev = {"Phi1": eng._dev(Phi1.astype(np.float32), torch.float32, "Phi1"), "Phi2": eng._dev(Phi2.astype(np.float32), torch.float32, "Phi2"),
               "a1": eng._dev(a1.astype(np.float32), torch.float32, "a1"), "a2": eng._dev(a2.astype(np.float32), torch.float32, "a2"),
               "lam1": eng._dev(lam1, torch.float64, "lam1"), "lam2": eng._dev(lam2, torch.float64, "lam2"),
               "F1": eng._dev(F1d, tdt[fdt], "F1"), "F2": eng._dev(F2d, tdt[fdt], "F2")}
        fp = dict(w_descr=1e-1, w_lap=1e-3, w_dcomm=1, w_p2p=0, w_stochastic=0, w_ent=0, w_range01=0, w_sumto1=0, w_area=0, w_conformal=0,
                  optinit="zeros", maxiter=1000000, stopping="reference")
        fp.update({k_: v for k_, v in fit_params.items() if k_ in fp})
        general = {n: fp[n] for n in ("w_dcomm", "w_p2p", "w_stochastic", "w_ent", "w_range01", "w_sumto1", "w_area", "w_conformal")}
        from .pyFM.functional import CLOSED_FORM_MAX_K1
        wide = n_ev > CLOSED_FORM_MAX_K1              # (FunctionalMapping.fit: maps wider than the closed form's solvers take the iterative scheme, tight)
        if any(v > 0 for v in general.values()) or wide:
            from .pyFM.functional import LBFGS_OPTIONS, LBFGS_WIDE
            x0 = np.stack([m.get_x0(optinit=fp["optinit"]) for m in g])
            C0, res = eng.fit_general(dev, dict(w_descr=fp["w_descr"], w_lap=fp["w_lap"], **general), x0, maxiter=fp["maxiter"],
                                      lbfgs_options=LBFGS_WIDE if wide else (LBFGS_OPTIONS if fp["stopping"] == "tight" else None))
            C0 = np.asarray(C0, dtype=np.float64)
        else:
            res = None
            A = eng.project(dev["Phi1"], dev["a1"], dev["F1"])
            Bm = eng.project(dev["Phi2"], dev["a2"], dev["F2"])
            c00 = eng.c00(Phi1, Phi2, a1, a2)
            C0 = eng.fmap_solve(A, Bm, dev["lam1"], dev["lam2"], c00, fp["w_descr"], fp["w_lap"], check=True).cpu().numpy()
        # ---- vertex maps of the fitted map, precise map, ICP, vertex maps of the ICP map: one call each for the group
        _, P1, P2, A1d = eng._reals(Phi1, Phi2, a1)
        C0d = eng._dev(C0, torch.float64, "C")
        maps0 = eng.fm_to_p2p(P1, P2, A1d, C0d)
        lr = eng.lsa_indicator_ok(P1.shape[1], P2.shape[1], C0d.shape[2], C0d.shape[1])     # assignments from the indicators' factors
        M0 = eng.mapped_indicator(P1, P2, A1d, C0d) if (compute_extra and not lr) else None
        # A call that runs as ONE chunk has nothing beside which its assignments could run.  Its searches go out as early as their
        # matrices exist, the long ones first: the fitted maps' on a stream of their own as soon as the fit is done, the ICP maps'
        # (an ICP that drifted gives the slowest matrix of a batch: 82 ms) on another right after ICP -- which runs BEFORE the precise
        # map here; the precise maps' (18 ms) close the chunk on this stream (lsa_streams; compute_surface_map does the same for one pair).
        early = EARLY_ASSIGNMENTS and lr and compute_extra and lsa_streams is not None
        early_plain = early_icp = None
        if early:
            early_plain = _on_side_stream(lsa_streams[0], (P1, P2, A1d, C0d),
                                          lambda: default_engine().lsa_indicator(P1, P2, A1d, C0d, maximize=True, defer=True))

        def run_icp():
            Ci_, resid, info = eng.icp(P1, P2, C0d, nit=10, return_resid=True)
            bad = np.nonzero((info.cpu().numpy() != 0) | ~(resid.cpu().numpy() <= 1e-8))[0]
            if bad.size:
                # rank-deficient least-squares maps: those pairs re-run with the reference's lstsq + SVD (pyFM/refine/icp.py: icp_host_svd)
                import warnings
                from .pyFM.refine.icp import icp_host_svd
                warnings.warn(f"ICP: the polar iteration did not converge for pairs {bad.tolist()[:8]} of the group; they run the "
                              "reference's lstsq + SVD on the host")
                for q_ in bad:
                    Ci_[q_] = torch.as_tensor(icp_host_svd(C0[q_], g[q_].mesh1.eigenvectors[:, :n_ev], g[q_].mesh2.eigenvectors[:, :n_ev], 10),
                                              dtype=torch.float64).to(Ci_.device)
            return Ci_
        Ci = None
        if early:
            Ci = run_icp()
            early_icp = _on_side_stream(lsa_streams[1], (P1, P2, A1d, Ci),
                                        lambda: default_engine().lsa_indicator(P1, P2, A1d, Ci, maximize=True, defer=True))
        prec = None
        if compute_extra:
            faces = np.ascontiguousarray(np.stack([m.mesh1.facelist for m in g]), dtype=np.int32)
            prec = eng.precise_map(P1, P2, C0d, faces, dense=True, scratch=True)[2]       # (never leaves this function)
        if Ci is None:
            Ci = run_icp()
        mapsi = eng.fm_to_p2p(P1, P2, A1d, Ci)
        # ---- every assignment of the group in one launch (a matrix is one workgroup): rows [plain | precise | ICP] when compute_extra
        # (Measured, r05, same box, alternating: with two chunks, every chunk's assignments started early 396-432 ms against 342 -- the
        #  searches sit on the compute units the other chunk's fit is balanced over --, only the last chunk's 356-391 against 339-373;
        #  one chunk: 380-383 against 399-404.  So: one chunk only.)
        if early:
            cp_ = eng.linear_sum_assignment(prec, maximize=True).cpu().numpy().astype(np.int64)
            with torch.cuda.stream(lsa_streams[0]):
                c0_ = early_plain().cpu().numpy().astype(np.int64)
            with torch.cuda.stream(lsa_streams[1]):
                ci_ = early_icp().cpu().numpy().astype(np.int64)
            cols = np.concatenate([c0_, cp_, ci_])
        elif lr:
            # the indicators by their factors (no N2 x N1 matrices: dm_lsa_indicator), the precise maps dense, in the same launch
            if compute_extra:
                c_ = eng.lsa_indicator(torch.cat([P1, P1]), torch.cat([P2, P2]), torch.cat([A1d, A1d]), torch.cat([C0d, Ci]), dense=prec,
                                       maximize=True).cpu().numpy().astype(np.int64)
                cols = np.concatenate([c_[:nb], c_[2 * nb:], c_[nb:2 * nb]])
            else:
                cols = eng.lsa_indicator(P1, P2, A1d, Ci, maximize=True).cpu().numpy().astype(np.int64)
        else:
            Mi = eng.mapped_indicator(P1, P2, A1d, Ci)
            mats = ([M0, prec] if compute_extra else []) + [Mi]
            # (unless the concatenation, a COPY next to the originals, would pass the same limit _assign_many has: then kind by kind)
            if sum(m_.numel() * m_.element_size() for m_ in mats) > _ASSIGN_STACK_LIMIT and len(mats) > 1:
                cols = np.concatenate([eng.linear_sum_assignment(m_, maximize=True).cpu().numpy() for m_ in mats]).astype(np.int64)
            else:
                cols = eng.linear_sum_assignment(torch.cat(mats, dim=0) if len(mats) > 1 else mats[0], maximize=True).cpu().numpy().astype(np.int64)

        def assignment(c):
            rows = np.nonzero(c >= 0)[0]
            return rows, c[rows]
        h = {n: v.cpu().numpy().astype(np.int64) for n, v in maps0.items()}
        hi = {n: v.cpu().numpy().astype(np.int64) for n, v in mapsi.items()}
        Ci_h = Ci.cpu().numpy()
        for q, i in enumerate(gidx):
            model = g[q]
            model.FM = C0[q]
            model._FM_icp = Ci_h[q]
            model.FM_type = "icp"
            model.eta = np.ones(model.mesh2.n_vertices)
            if res is not None:
                import types
                model.fit_result = types.SimpleNamespace(nit=res.nit[q:q + 1], nfev=res.nfev[q:q + 1], fun=res.fun[q:q + 1],
                                                         status=res.status[q:q + 1], message=res.message[q:q + 1])
            model.mapped_indicator = MappedIndicator(eng, P1[q:q + 1], P2[q:q + 1], A1d[q:q + 1], Ci[q:q + 1], hi["ind21"][q], hi["ind12"][q])
            hung = assignment(cols[q]) if compute_extra else None
            hung_p = assignment(cols[nb + q]) if compute_extra else None
            hung_i = assignment(cols[(2 * nb if compute_extra else 0) + q])
            out[i] = (h["ind21"][q], h["ind12"][q], hung, hung_p, hi["ind21"][q], hi["ind12"][q], hung_i, model, model.mesh1, model.mesh2,
                      h["knn21"][q], h["knn12"][q], hi["knn21"][q], hi["knn12"][q])


def compute_surface_map_batch(meshes1_t, meshes2_t, c1s, c2s, n_ev=50, compute_extra=False, optimizer="fmin_l_bfgs_b", descr_type="neural",
                              maxiter=100000, optimize_p2p=False, fit_params=None, streams=None, robust_backend=None):
    """compute_surface_map for a list of mesh pairs: returns the list of the 14-tuples compute_surface_map returns, each equal to the
    single call's (same kernels, and every kernel's result for a pair is independent of the batch it is in).  No counterpart in
    the reference (it matches one pair per call, functional_map.py:9-81): this is its documented call with the batch dimension the
    GPU path has everywhere -- one batched eigensolve for the meshes, one device L-BFGS over the maps, the 2 x 4 vertex maps,
    precise maps and ICP of all pairs in one library call each, and all linear assignments side by side (one workgroup per
    matrix; a single call leaves 253 of 256 CUs idle there).  Pairs are grouped by (vertex counts, face count of mesh 1): the
    batched kernels take one size per call; a group of one runs the same path with a batch of one.
    streams: the batch is cut into that many contiguous chunks, each run by its own host thread on its own HIP stream (its own
    MatchEngine: context, workspace).  The stages of a chunk depend on each other, the chunks do not: while one chunk's iterative fit
    fills the vector ALUs, another's eigensolver (a chain of a thousand small launches), linear assignments (a workgroup per matrix,
    latency bound) and host-side bookkeeping run beside it.  Default: one chunk per 64 pairs, at most four (r05, float64 fit: 500 ms on one stream, 455 on two, 495 on three, 580 on four; r06, fp32 element loop and a one-chunk call's assignments started early: 260 / 255-297 / 267-308 ms on one / two / three).  A pair's results do
    not depend on the chunking.
    robust_backend: as for compute_surface_map.  Two host threads may call concurrently when each calls on its own HIP stream."""
    with _robust_backend_for_call(robust_backend):
        out = _compute_surface_map_batch(meshes1_t, meshes2_t, c1s, c2s, n_ev, compute_extra, optimizer, descr_type, maxiter, optimize_p2p,
                                         fit_params, streams)
    for t in out:
        if t is not None:
            t[8].release_device_rows(); t[9].release_device_rows()
    return out


def _compute_surface_map_batch(meshes1_t, meshes2_t, c1s, c2s, n_ev, compute_extra, optimizer, descr_type, maxiter, optimize_p2p, fit_params, streams):
    import torch
    assert descr_type == "neural", "the batched call takes network descriptors (descr_type='neural')"
    B = len(meshes1_t)
    assert len(meshes2_t) == B and len(c1s) == B and len(c2s) == B
    fit_params = dict(fit_params or {})
    fit_params.pop("verbose", None)
    known = {"w_descr", "w_lap", "w_dcomm", "w_p2p", "w_stochastic", "w_ent", "w_range01", "w_sumto1", "w_area", "w_conformal", "optinit",
             "maxiter", "stopping", "w_orient", "w_area_difference", "w_mumford_shah", "mumford_shah_var", "w_eta_entropy", "orient_reversing",
             "device", "driver"}
    unknown = set(fit_params) - known
    if unknown:
        raise TypeError(f"fit() got unexpected keyword arguments {sorted(unknown)}")
    if any(fit_params.get(n, 0) > 0 for n in ("w_area_difference", "w_mumford_shah", "w_eta_entropy")):
        raise NotImplementedError("area-difference / Mumford-Shah / eta-entropy terms are not on the accelerated path; pass 0")
    if fit_params.get("w_orient", 0) > 0:
        raise NotImplementedError("w_orient: the orientation operators are built per pair on the host (FunctionalMapping.fit); "
                                  "use compute_surface_map for it")
    if fit_params.get("stopping", "reference") not in ("tight", "reference"):
        raise ValueError("stopping must be 'tight' or 'reference'")
    timing = os.environ.get("TIMEIT", False)
    if timing:
        compute_extra = True
    models = []
    for i in range(B):
        m1 = TriMesh(_np(meshes1_t[i].verts_list()[0]), _np(meshes1_t[i].faces_list()[0]))
        m2 = TriMesh(_np(meshes2_t[i].verts_list()[0]), _np(meshes2_t[i].faces_list()[0]))
        model = FunctionalMapping(m1, m2, partial=False, optimizer=optimizer)
        model.k1, model.k2 = n_ev, n_ev
        model.descr1, model.descr2 = _np(c1s[i]), _np(c2s[i])
        models.append(model)
    out = [None] * B
    if streams is None:
        # r06 (the fit's element loop in fp32: 1.7 x faster): at 64 pairs one chunk -- its assignments started early on side streams --
        # takes 259-264 ms, two chunks 255-297, three 267-308 (tools/batch_stage_times.py --streams=N): a chunk per 64 pairs
        streams = min(4, -(-B // 64))
    streams = max(1, min(int(streams), B))
    dev_index = torch.cuda.current_device()
    if streams == 1:
        lsa_side = _side_streams(dev_index, 2)                                  # the two assignment streams
        _batch_chunk(models, list(range(B)), out, n_ev, compute_extra, fit_params, lsa_streams=lsa_side)
        return out
    from concurrent.futures import ThreadPoolExecutor
    from .shard import block_range
    caller = torch.cuda.current_stream(dev_index)
    side = _side_streams(dev_index, streams)

    # The chunks run STAGGERED by one stage: chunk c starts its eigensolve when chunk c - 1 has finished its own.  Started together they
    # would move in lockstep -- every stream in the same stage, competing for the same resource -- and gain little (measured, 64 pairs:
    # 500 ms on one stream, 462 on two, 565 on four); staggered, one chunk's eigensolver (launch-latency bound) runs beside the previous
    # chunk's fit (vector-ALU bound), whose linear assignments (one workgroup per matrix) run beside the next chunk's fit.
    import threading
    eig_done = [threading.Event() for _ in range(streams)]

    def run(c):
        try:
            lo, hi = block_range(B, c, streams)
            torch.cuda.set_device(dev_index)
            if c > 0:
                eig_done[c - 1].wait()
            side[c].wait_stream(caller)
            with torch.cuda.stream(side[c]):
                _batch_chunk(models, list(range(lo, hi)), out, n_ev, compute_extra, fit_params, after_eigenbases=eig_done[c].set)
            side[c].synchronize()
        finally:
            eig_done[c].set()                                             # (a chunk that failed must not leave the next one waiting)
    with ThreadPoolExecutor(max_workers=streams) as ex:
        for f in [ex.submit(run, c) for c in range(streams)]:
            f.result()                                                    # (re-raises a chunk's exception here)
    return out


_PINNED = {}


def _pinned_buffer(key, shape, dtype):
    """a page-locked host tensor of this shape, kept for the process (one per chunk stream, side and group key; the previous call's upload
    from it was waited for before that call returned)"""
    import torch
    buf = _PINNED.get(key)
    if buf is None or tuple(buf.shape) != tuple(shape) or buf.dtype != dtype:
        buf = torch.empty(shape, dtype=dtype, pin_memory=True)
        if len(_PINNED) > 16:
            _PINNED.clear()
        _PINNED[key] = buf
    return buf


_UPLOAD_STREAMS = {}


def _upload_stream(main_stream):
    """the stream a chunk's descriptors are uploaded on (one per chunk stream, created once)"""
    import torch
    key = (main_stream.device.index, main_stream.cuda_stream)
    if key not in _UPLOAD_STREAMS:
        _UPLOAD_STREAMS[key] = torch.cuda.Stream(device=main_stream.device)
    return _UPLOAD_STREAMS[key]


_SIDE_STREAMS = {}


def _side_streams(dev_index, n, caller=None):
    """the chunk / assignment streams of compute_surface_map[_batch], created once per (device, caller's stream) (every stream brings a
    MatchEngine with its own workspace arena: default_engine() keys on the stream).  Keyed on the caller's stream as well (ADVICE
    r05): two host threads that call on their own streams get their own side streams, upload streams (keyed on the chunk stream) and
    page-locked buffers (ditto) -- one dm_ctx is never driven from two threads.  Two threads on the SAME stream are not supported (nor
    are they by the one-engine-per-(device, stream) design anywhere else in the package)."""
    import torch
    if caller is None:
        caller = torch.cuda.current_stream(dev_index)
    have = _SIDE_STREAMS.setdefault((dev_index, caller.cuda_stream), [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=dev_index))
    return have[:n]

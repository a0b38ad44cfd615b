"""
MatchEngine: batched matching hot path on one MI355X, bound to libdensematch through ctypes.

PyTorch is used for device memory and streams only: every method takes / returns
torch tensors that live on the engine's GPU and hands their `data_ptr()` to the
C ABI (include/densematch.h).  All arithmetic of the path runs in the HIP kernels.

Batch layout (B pairs; pairs are independent):
    Phi1 (B,N1,ld1) f32|f64  Phi2 (B,N2,ld2) f32|f64   eigenvectors (first k columns are used)
    lam1 (B,k1) f64          lam2 (B,k2) f64           eigenvalues
    a1 (B,N1) f32|f64        a2 (B,N2) f32|f64         lumped masses
  (float64 eigenvectors / masses -- the reference's own dtype, pyFM/mesh/trimesh.py:118 -- take the *_f64 entry points
   of the vertex-map, refinement and conversion calls; the projection consumes fp32 like the reference's fit does)
    F1 (B,N1,D) f16|f32   F2 (B,N2,D) f16|f32    descriptors
    C  (B,k2,k1) f64                              functional maps
    maps int32 on the device
"""
import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class _KeepAlive(tuple):
    """an argument tuple that also holds the tensors its raw pointers refer to"""


class MatchEngine:
    def __init__(self, device=None, lib_path=None):
        if not torch.cuda.is_available():
            raise RuntimeError("MatchEngine needs a ROCm GPU (gfx950); there is no CPU fallback")
        self.lib = _lib.load(lib_path)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else
                                   (device if isinstance(device, int) else torch.device(device).index or 0))
        self.stream = torch.cuda.current_stream(self.device)
        ctx = C.c_void_p()
        rc = self.lib.dm_create(self.device.index, C.c_void_p(self.stream.cuda_stream), C.byref(ctx))
        if rc != 0:
            raise _lib.DenseMatchError(f"dm_create failed with status {rc}: {self.lib.dm_last_error(None).decode()}")
        self.ctx = ctx

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.dm_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _chk(self, rc):
        _lib.raise_for(rc, self.lib, self.ctx)

    def _dev(self, t, dtype, name):
        # The context launches on the stream it was created on; temporaries and outputs come from torch's caching
        # allocator, which recycles a block as soon as the *current* stream is done with it.  Using an engine under
        # another stream would let blocks be reused while our kernels still run: refuse it.
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self.stream.cuda_stream:
            raise RuntimeError("MatchEngine is bound to the stream it was created on; create one engine per stream "
                               "(default_engine() does) instead of calling it under torch.cuda.stream(other)")
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t)
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous():
            t = t.to(device=self.device, dtype=dtype).contiguous()
        return t

    def _reals(self, *arrays):
        """Eigenvector / mass arrays of one call on the device in ONE dtype: float64 if any of them is float64 (then the
        *_f64 entry points run: the reference's arithmetic on its own inputs), else float32.  None entries pass through.
        Returns (suffix, arrays...) with suffix "" or "_f64"."""
        def dt(a):
            return a.dtype if isinstance(a, torch.Tensor) else getattr(a, "dtype", None)
        f64 = any(a is not None and str(dt(a)).endswith("float64") for a in arrays)
        tdt = torch.float64 if f64 else torch.float32
        return ("_f64" if f64 else "",) + tuple(None if a is None else self._dev(a, tdt, "real") for a in arrays)

    def synchronize(self):
        self.stream.synchronize()

    OPTION_DEFAULTS = {"simnn_pipe": 1, "simnn_persist": 1, "knn_split": 1, "p2p_split": 2, "solve_packed": 0, "solve_reg": 1, "simnn_band": 4, "lsa_reg": 2, "simnn_big": 0, "energy_keep_gram": 0,
                       "p2pfm_direct": 1, "zoomout_fused": 1, "proj_onepass": 1, "fit_f32": 0, "fit_mfma": 1, "basis_stats": 1, "solve_pcg": 1}

    def set_option(self, name, value):
        """Choose between equivalent code paths of the library (include/densematch.h: dm_set_option); every setting
        returns the same, exact results."""
        self._chk(self.lib.dm_set_option(self.ctx, name.encode(), int(value)))

    def reset_options(self):
        for k, v in self.OPTION_DEFAULTS.items():
            self.set_option(k, v)

    @staticmethod
    def split_depth(k):
        """fp16 contraction depth of the split features for a float64 depth k (dm_knnsplit.hip: 3 entries per index + 8
        bias slots, padded to the 32-wide stage, at least 96)."""
        return max(96, -(-(8 + 3 * k) // 32) * 32)

    def p2p_split_active(self, N2, N1, k):
        """0 when fm_to_p2p runs the float64 kernel for these sizes, 1: two passes of the two-key fp16 tile kernel, 2: one
        pass in both directions (bench.py names its dominant kernel accordingly)."""
        return int(self.lib.dm_fm_to_p2p_uses_split(self.ctx, int(N2), int(N1), int(k)))

    def last_requeued_rows(self):
        """rows of the last fm_to_p2p / simnn call that took the exact float64 path, per reduction (knn21, ind21, knn12, ind12;
        -1: not applicable) -- include/densematch.h: dm_last_requeued_rows"""
        out = (C.c_int * 4)()
        self._chk(self.lib.dm_last_requeued_rows(self.ctx, out))
        return [int(x) for x in out]

    def workspace_bytes(self):
        return int(self.lib.dm_workspace_bytes(self.ctx))

    def profile_kernel(self, name):
        self._chk(self.lib.dm_profile_kernel(self.ctx, name.encode() if name else None))

    def profile_read(self):
        n, ms = C.c_int(0), C.c_double(0.0)
        self._chk(self.lib.dm_profile_read(self.ctx, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    PEAK_PROBES = {"mfma_f16_zero_operands": 0, "mfma_f16_random_operands": 1, "mfma_f64": 2, "hbm_copy": 3}

    def measure_peak(self, which):
        """on-box peak probe (include/densematch.h: dm_measure_peak): FLOP/s, or bytes/s for "hbm_copy" """
        v = C.c_double(0.0)
        self._chk(self.lib.dm_measure_peak(self.ctx, self.PEAK_PROBES[which], C.byref(v)))
        return v.value

    def profile_report(self):
        """after profile_kernel("*"): {kernel name: (launches, total ms)} of every launch since, in order of first launch"""
        buf = C.create_string_buffer(1 << 16)
        self._chk(self.lib.dm_profile_report(self.ctx, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split("\t")
            out[name] = (int(n), float(ms))
        return out

    # ------------------------------------------------------------------ ops
    def simnn(self, Ftgt, Fsrc, return_scores=False):
        """nn[b,i] = argmax_j <Ftgt[b,i], Fsrc[b,j]> (float64-exact, lowest index on ties)."""
        Ftgt = self._dev(Ftgt, torch.float16, "Ftgt")
        Fsrc = self._dev(Fsrc, torch.float16, "Fsrc")
        if Ftgt.dim() != 3 or Fsrc.dim() != 3 or Ftgt.shape[0] != Fsrc.shape[0] or Ftgt.shape[2] != Fsrc.shape[2]:
            raise ValueError("simnn expects Ftgt (B,N2,D) and Fsrc (B,N1,D)")
        B, N2, D = Ftgt.shape
        N1 = Fsrc.shape[1]
        nn = torch.empty((B, N2), dtype=torch.int32, device=self.device)
        best = torch.empty((B, N2), dtype=torch.float32, device=self.device) if return_scores else None
        margin = torch.empty((B, N2), dtype=torch.float32, device=self.device) if return_scores else None
        self._chk(self.lib.dm_simnn_f16(self.ctx, B, N2, N1, D, _ptr(Ftgt), _ptr(Fsrc), _ptr(nn), _ptr(best), _ptr(margin)))
        return (nn, best, margin) if return_scores else nn

    def project(self, Phi, mass, F, k=None, out=None, exact=False):
        """Phi[:, :k]^T (mass * F)  ->  (B,k,D) f32.  fp16 descriptors use the fp16 matrix cores (split basis,
        relative error ~1e-6); exact=True or fp32 descriptors use the float64 matrix cores."""
        sfx, Phi, mass = self._reals(Phi, mass)
        if not isinstance(F, torch.Tensor):
            F = torch.as_tensor(F)
        fdt = torch.float16 if F.dtype == torch.float16 else torch.float32
        F = self._dev(F, fdt, "F")
        B, N, ld = Phi.shape
        k = ld if k is None else k
        if F.shape[:2] != (B, N) or mass.shape != (B, N):
            raise ValueError("project: Phi (B,N,ld), mass (B,N), F (B,N,D) do not agree")
        D = F.shape[2]
        if out is None:
            out = torch.empty((B, k, D), dtype=torch.float32, device=self.device)
        self._chk(getattr(self.lib, "dm_project" + sfx)(self.ctx, B, N, D, k, _ptr(Phi), ld, _ptr(mass), _ptr(F),
                                      (_lib.DM_F16 if fdt == torch.float16 else _lib.DM_F32) |
                                      (_lib.DM_PROJECT_F64 if exact else 0), _ptr(out)))
        return out

    def c00(self, Phi1, Phi2, a1, a2):
        sfx, Phi1, Phi2, a1, a2 = self._reals(Phi1, Phi2, a1, a2)
        B, N1, ld1 = Phi1.shape
        _, N2, ld2 = Phi2.shape
        out = torch.empty((B,), dtype=torch.float64, device=self.device)
        self._chk(getattr(self.lib, "dm_fmap_c00" + sfx)(self.ctx, B, N1, N2, _ptr(Phi1), ld1, _ptr(Phi2), ld2, _ptr(a1), _ptr(a2), _ptr(out)))
        return out

    def fmap_fit(self, Phi1, Phi2, a1, a2, F1, F2, lam1, lam2, w_descr, w_lap, k1=None, k2=None, check=True):
        """project(mesh 1) + project(mesh 2) + c00 + fmap_solve in one library call (dm_fmap_fit): the same C bit for bit; the
        projected descriptors are not returned.  fp16 descriptors."""
        sfx, Phi1, Phi2, a1, a2 = self._reals(Phi1, Phi2, a1, a2)
        F1 = self._dev(F1, torch.float16, "F1")
        F2 = self._dev(F2, torch.float16, "F2")
        lam1 = self._dev(lam1, torch.float64, "lam1")
        lam2 = self._dev(lam2, torch.float64, "lam2")
        B, N1, ld1 = Phi1.shape
        _, N2, ld2 = Phi2.shape
        k1 = lam1.shape[1] if k1 is None else k1
        k2 = lam2.shape[1] if k2 is None else k2
        D = F1.shape[2]
        if (F1.shape[:2] != (B, N1) or F2.shape != (B, N2, D) or a1.shape != (B, N1) or a2.shape != (B, N2) or lam1.shape != (B, k1)
                or lam2.shape != (B, k2)):
            raise ValueError("fmap_fit: shapes do not agree")
        Cm = torch.empty((B, k2, k1), dtype=torch.float64, device=self.device)
        info = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._chk(getattr(self.lib, "dm_fmap_fit" + sfx)(self.ctx, B, N1, N2, D, k1, k2, _ptr(Phi1), ld1, _ptr(Phi2), ld2, _ptr(a1), _ptr(a2),
                                                         _ptr(F1), _ptr(F2), _ptr(lam1), _ptr(lam2), float(w_descr), float(w_lap),
                                                         _ptr(Cm), _ptr(info)))
        if check:
            bad = torch.nonzero(info).flatten()
            if bad.numel():
                raise _lib.DenseMatchError(
                    f"functional-map system not positive definite for pairs {bad.tolist()[:8]} "
                    f"(row {int(info[bad[0]]) - 1}): descriptors are rank deficient in the basis and w_lap cannot fix it")
        return Cm

    def fmap_solve(self, A, Bm, lam1, lam2, c00, w_descr, w_lap, check=True):
        """Closed-form minimiser of the w_descr / w_lap energy -> C (B,k2,k1) f64."""
        A = self._dev(A, torch.float32, "A")
        Bm = self._dev(Bm, torch.float32, "Bm")
        lam1 = self._dev(lam1, torch.float64, "lam1")
        lam2 = self._dev(lam2, torch.float64, "lam2")
        c00 = self._dev(c00, torch.float64, "c00")
        B, k1, D = A.shape
        k2 = Bm.shape[1]
        if Bm.shape[0] != B or Bm.shape[2] != D or lam1.shape != (B, k1) or lam2.shape != (B, k2) or c00.shape != (B,):
            raise ValueError("fmap_solve: shapes do not agree")
        Cm = torch.empty((B, k2, k1), dtype=torch.float64, device=self.device)
        info = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._chk(self.lib.dm_fmap_solve(self.ctx, B, k1, k2, D, _ptr(A), _ptr(Bm), _ptr(lam1), _ptr(lam2), _ptr(c00),
                                         float(w_descr), float(w_lap), _ptr(Cm), _ptr(info)))
        if check:
            bad = torch.nonzero(info).flatten()
            if bad.numel():
                raise _lib.DenseMatchError(
                    f"functional-map system not positive definite for pairs {bad.tolist()[:8]} "
                    f"(row {int(info[bad[0]]) - 1}): descriptors are rank deficient in the basis and w_lap cannot fix it")
        return Cm

    WEIGHT_ORDER = ("w_descr", "w_lap", "w_dcomm", "w_p2p", "w_stochastic", "w_ent", "w_range01", "w_sumto1", "w_area", "w_conformal")

    def descr_ops(self, Phi, mass, F, k=None):
        """Multiplication operators of the descriptors in the reduced basis, Phi^T diag(mass * f_d) Phi -> (B,D,k,k) f64
        (reference base_functions.py:550-555)."""
        Phi = self._dev(Phi, torch.float32, "Phi")
        mass = self._dev(mass, torch.float32, "mass")
        if not isinstance(F, torch.Tensor):
            F = torch.as_tensor(F)
        fdt = torch.float16 if F.dtype == torch.float16 else torch.float32
        F = self._dev(F, fdt, "F")
        B, N, ld = Phi.shape
        k = ld if k is None else k
        D = F.shape[2]
        ops = torch.empty((B, D, k, k), dtype=torch.float64, device=self.device)
        self._chk(self.lib.dm_fmap_descr_ops(self.ctx, B, N, D, k, _ptr(Phi), ld, _ptr(mass), _ptr(F),
                                             _lib.DM_F16 if fdt == torch.float16 else _lib.DM_F32, _ptr(ops)))
        return ops

    def energy_grad(self, Cm, A, Bm, lam1, lam2, weights, Phi1=None, Phi2=None, a1=None, ops1=None, ops2=None):
        """Energy (B,) and gradient (B,k2,k1) of the functional-map objective for the weights in `weights` (dict with keys
        of WEIGHT_ORDER; reference energy_func_std / grad_energy_std, base_functions.py:480-763)."""
        ev = self._energy_grad_args(Cm, A, Bm, lam1, lam2, weights, Phi1, Phi2, a1, ops1, ops2)
        Cm = ev[-1]
        B, k2, k1 = Cm.shape
        energy = torch.empty((B,), dtype=torch.float64, device=self.device)
        grad = torch.empty((B, k2, k1), dtype=torch.float64, device=self.device)
        self._chk(self.lib.dm_fmap_energy_grad(self.ctx, *ev[:-1], _ptr(Cm), _ptr(energy), _ptr(grad)))
        return energy, grad

    def _energy_grad_args(self, Cm, A, Bm, lam1, lam2, weights, Phi1=None, Phi2=None, a1=None, ops1=None, ops2=None):
        """the argument list dm_fmap_energy_grad and dm_fmap_fit_steps share (B ... weights), followed by the map tensor; the tensors
        behind the pointers are kept alive by the tuple"""
        Cm = self._dev(Cm, torch.float64, "C")
        A = self._dev(A, torch.float32, "A")
        Bm = self._dev(Bm, torch.float32, "Bm")
        lam1 = self._dev(lam1, torch.float64, "lam1")
        lam2 = self._dev(lam2, torch.float64, "lam2")
        B, k2, k1 = Cm.shape
        D = A.shape[2]
        if A.shape != (B, k1, D) or Bm.shape != (B, k2, D) or lam1.shape != (B, k1) or lam2.shape != (B, k2):
            raise ValueError("energy_grad: shapes do not agree")
        unknown = set(weights) - set(self.WEIGHT_ORDER)
        if unknown:
            raise ValueError(f"energy_grad: unknown weights {sorted(unknown)}")
        w = (C.c_double * 10)(*[float(weights.get(n, 0.0)) for n in self.WEIGHT_ORDER])
        N1 = N2 = ld1 = ld2 = 1
        if Phi1 is not None:
            Phi1 = self._dev(Phi1, torch.float32, "Phi1")
            Phi2 = self._dev(Phi2, torch.float32, "Phi2")
            a1 = self._dev(a1, torch.float32, "a1")
            _, N1, ld1 = Phi1.shape
            _, N2, ld2 = Phi2.shape
        n_ops = 0
        if ops1 is not None:
            ops1 = self._dev(ops1, torch.float64, "ops1")
            ops2 = self._dev(ops2, torch.float64, "ops2")
            n_ops = ops1.shape[1]
            if ops1.shape != (B, n_ops, k1, k1) or ops2.shape != (B, n_ops, k2, k2):
                raise ValueError("energy_grad: descriptor operators must be (B,D,k1,k1) and (B,D,k2,k2)")
        keep = (Phi1, Phi2, a1, A, Bm, lam1, lam2, ops1, ops2, w)
        args = _KeepAlive((B, N1, N2, k1, k2, D, _ptr(Phi1), ld1, _ptr(Phi2), ld2, _ptr(a1), _ptr(A), _ptr(Bm), _ptr(lam1), _ptr(lam2),
                           _ptr(ops1), _ptr(ops2), n_ops, C.cast(w, C.c_void_p), Cm))
        args.keep = keep
        return args

    # SciPy's L-BFGS-B defaults, i.e. what the reference's `minimize(..., options={'maxiter': maxiter})` runs with (functional.py:477)
    LBFGS_REFERENCE = {"maxcor": 10, "ftol": 2.220446049250313e-09, "gtol": 1e-5, "maxfun": 15000, "maxls": 20}
    LBFGS_STATUS = {0: "running", 1: "CONVERGENCE: NORM OF PROJECTED GRADIENT <= PGTOL", 2: "CONVERGENCE: REL_REDUCTION_OF_F <= FACTR*EPSMCH",
                    3: "STOP: TOTAL NO. of ITERATIONS REACHED LIMIT", 4: "STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT",
                    5: "ABNORMAL_TERMINATION_IN_LNSRCH (or a non-finite energy / gradient)"}

    def _fit_inputs(self, batch, weights, k, orient_ops):
        """projected descriptors, spectra, operator lists and the weight dictionary dm_fmap_energy_grad takes: the orientation term is
        a commutation term like w_dcomm (base_functions.py:567-600), its operator pairs ride in the same lists scaled by
        sqrt(w_orient / w_dcomm) (w_dcomm = 1 when only w_orient is set)"""
        k1 = k if k is not None else batch["lam1"].shape[1]
        k2 = k if k is not None else batch["lam2"].shape[1]
        Phi1, Phi2 = batch["Phi1"], batch["Phi2"]
        A = self.project(Phi1, batch["a1"], batch["F1"], k1)
        Bm = self.project(Phi2, batch["a2"], batch["F2"], k2)
        lam1 = self._dev(batch["lam1"], torch.float64, "lam1")[:, :k1].contiguous()
        lam2 = self._dev(batch["lam2"], torch.float64, "lam2")[:, :k2].contiguous()
        w = {n: float(v) for n, v in weights.items() if n != "w_orient"}
        w_orient, w_dcomm = float(weights.get("w_orient", 0.0)), float(weights.get("w_dcomm", 0.0))
        ops1 = ops2 = None
        if w_dcomm > 0:
            ops1 = self.descr_ops(Phi1, batch["a1"], batch["F1"], k1)
            ops2 = self.descr_ops(Phi2, batch["a2"], batch["F2"], k2)
        if w_orient > 0:
            if orient_ops is None:
                raise ValueError("w_orient > 0 needs the orientation operators (FunctionalMapping.compute_orientation_op)")
            sc = (w_orient / w_dcomm) ** 0.5 if w_dcomm > 0 else w_orient ** 0.5
            o1 = self._dev(orient_ops[0], torch.float64, "orient_ops1") * sc
            o2 = self._dev(orient_ops[1], torch.float64, "orient_ops2") * sc
            ops1 = o1 if ops1 is None else torch.cat([ops1, o1], dim=1).contiguous()
            ops2 = o2 if ops2 is None else torch.cat([ops2, o2], dim=1).contiguous()
            if w_dcomm <= 0:
                w["w_dcomm"] = 1.0
        P1 = self._dev(Phi1, torch.float32, "Phi1")[:, :, :k1].contiguous()
        P2 = self._dev(Phi2, torch.float32, "Phi2")[:, :, :k2].contiguous()
        a1 = self._dev(batch["a1"], torch.float32, "a1")
        return A, Bm, lam1, lam2, w, P1, P2, a1, ops1, ops2, k1, k2

    def fit_energy(self, batch, weights, x, k=None, orient_ops=None):
        """energy (B,) of the fit's objective at the maps x (B,k2,k1) -- what FunctionalMapping.fit evaluates at x0 to rescale the
        orientation weight (functional.py:448-456)"""
        A, Bm, lam1, lam2, w, P1, P2, a1, ops1, ops2, k1, k2 = self._fit_inputs(batch, weights, k, orient_ops)
        w.setdefault("w_descr", 0.0)
        e, _ = self.energy_grad(self._dev(x, torch.float64, "x"), A, Bm, lam1, lam2, w, P1, P2, a1, ops1, ops2)
        return e.cpu().numpy()

    def _weight_array(self, weights):
        return (C.c_double * 10)(*[float(weights.get(n, 0.0)) for n in self.WEIGHT_ORDER])

    def fit_fused_ok(self, k1, k2, weights, ops1=None):
        """does dm_fmap_fit_fused take this fit (maps up to 32 x 32; w_descr, w_lap, w_p2p, w_ent, w_range01, w_sumto1 only)?"""
        w = self._weight_array(weights)
        return bool(self.lib.dm_fmap_fit_fused_ok(int(k1), int(k2), C.cast(w, C.c_void_p), 0 if ops1 is None else int(ops1.shape[1])))

    def _fit_fused(self, A, Bm, lam1, lam2, weights, P1, P2, a1, x0, opts, maxiter, precision="auto"):
        """the whole fit in one library call (dm_fmap_fit_fused: one launch per evaluation, the optimiser inside the kernel).
        precision: the element loop over the N2 x N1 entries of the mapped indicator (product, entropy / range terms, back-product):
        "f32" = the reference's precision (pyFM/functional.py:379-383 moves every tensor to float32), "f64", or "auto" (default): fp32
        under SciPy's stopping rule -- what the reference runs: the same iterations, the map within 3e-6 of the float64 loop's, 1.7 x
        faster per evaluation -- and float64 when the caller asks for a tighter one (ftol < 1e-10 is below the fp32 noise floor of the
        energy: the line search then ends in ABNORMAL_TERMINATION, like SciPy's does on a noisy function).  Everything around the
        element loop (Phi2 C, partial sums, quadratic and sum-to-one terms, L-BFGS) is float64 either way."""
        import types
        import numpy as np
        if precision not in ("auto", "f32", "f64"):
            raise ValueError("precision must be 'auto', 'f32' or 'f64'")
        f32 = precision == "f32" or (precision == "auto" and float(opts["ftol"]) >= 1e-10)
        B, k1, D = A.shape
        k2 = Bm.shape[1]
        _, N1, ld1 = P1.shape
        _, N2, ld2 = P2.shape
        w = self._weight_array(dict(weights, w_dcomm=0.0))      # (reached only without operator lists: the commutativity term is off)
        x0d = self._dev(x0, torch.float64, "x0")
        xo = torch.empty((B, k2, k1), dtype=torch.float64, device=self.device)
        fo = torch.empty((B,), dtype=torch.float64, device=self.device)
        info = torch.empty((B, 4), dtype=torch.int32, device=self.device)
        nev = C.c_int(0)
        self.set_option("fit_f32", 1 if f32 else 0)
        try:
            self._chk(self.lib.dm_fmap_fit_fused(self.ctx, B, N1, N2, k1, k2, D, _ptr(P1), ld1, _ptr(P2), ld2, _ptr(a1), _ptr(A), _ptr(Bm), _ptr(lam1),
                                                 _ptr(lam2), C.cast(w, C.c_void_p), int(opts["maxcor"]), _ptr(x0d), float(opts["ftol"]), float(opts["gtol"]),
                                                 int(maxiter), int(opts["maxfun"]), int(opts["maxls"]), _ptr(xo), _ptr(fo), _ptr(info), C.c_void_p(0),
                                                 C.byref(nev)))
        finally:
            self.set_option("fit_f32", 0)
        info = info.cpu().numpy()
        info[:, 0] = np.where(info[:, 0] == 0, 4, info[:, 0])
        res = types.SimpleNamespace(x=xo.cpu().numpy(), fun=fo.cpu().numpy(), status=info[:, 0].copy(), nit=info[:, 1].copy(),
                                    nfev=info[:, 2].copy(), message=[self.LBFGS_STATUS.get(int(q), "?") for q in info[:, 0]],
                                    success=bool(np.all((info[:, 0] == 1) | (info[:, 0] == 2))), evaluations=int(nev.value), path="fused",
                                    element_loop="f32" if f32 else "f64")
        return res.x, res

    def energy_grad_fused(self, Cm, A, Bm, lam1, lam2, weights, Phi1, Phi2, a1, precision="f64"):
        """energy (B,) and gradient (B,k2,k1) at the maps Cm through the arithmetic of dm_fmap_fit_fused (its single-evaluation mode):
        what the fused fit minimises, for tests and diagnostics.  precision "f64" | "f32": the element loop's (see _fit_fused)"""
        Cm = self._dev(Cm, torch.float64, "C")
        A = self._dev(A, torch.float32, "A")
        Bm = self._dev(Bm, torch.float32, "Bm")
        lam1 = self._dev(lam1, torch.float64, "lam1")
        lam2 = self._dev(lam2, torch.float64, "lam2")
        P1 = self._dev(Phi1, torch.float32, "Phi1")
        P2 = self._dev(Phi2, torch.float32, "Phi2")
        a1 = self._dev(a1, torch.float32, "a1")
        B, k2, k1 = Cm.shape
        D = A.shape[2]
        _, N1, ld1 = P1.shape
        _, N2, ld2 = P2.shape
        w = self._weight_array(dict(weights, w_dcomm=0.0))
        energy = torch.empty((B,), dtype=torch.float64, device=self.device)
        grad = torch.empty((B, k2, k1), dtype=torch.float64, device=self.device)
        self.set_option("fit_f32", 1 if precision == "f32" else 0)
        try:
            self._chk(self.lib.dm_fmap_fit_fused(self.ctx, B, N1, N2, k1, k2, D, _ptr(P1), ld1, _ptr(P2), ld2, _ptr(a1), _ptr(A), _ptr(Bm), _ptr(lam1),
                                                 _ptr(lam2), C.cast(w, C.c_void_p), 0, _ptr(Cm), 0.0, 0.0, 0, 0, 0, C.c_void_p(0), _ptr(energy), C.c_void_p(0),
                                                 _ptr(grad), None))
        finally:
            self.set_option("fit_f32", 0)
        return energy, grad

    def fit_general(self, batch, weights, x0, k=None, maxiter=15000, lbfgs_options=None, driver="device", check_every=4, orient_ops=None, fused=True,
                    precision="auto"):
        """FunctionalMapping.fit for any of the implemented energy terms, a whole batch at once (reference: L-BFGS-B through
        scipy.optimize.minimize, one pair per call, functional.py:477).  Every pair runs its OWN limited-memory BFGS iteration --
        history, step length, stopping test -- so a pair's result does not depend on the batch it is in; the optimiser state
        lives on the device (dm_lbfgs_*), energy and gradient of all pairs come from one dm_fmap_energy_grad call per
        evaluation, and the host reads the status words every `check_every` evaluations.
        lbfgs_options: maxcor, ftol, gtol, maxfun, maxls with SciPy's names; default = SciPy's defaults (LBFGS_REFERENCE).
        driver="scipy" (single pair only): the same evaluations driven by scipy.optimize.minimize on the host, as the reference.
        x0 (B,k2,k1): start, its first column is the pinned one.  Returns (C (B,k2,k1) numpy, result) with result.x, .fun (B),
        .status (B), .nit (B), .nfev (B), .message (list)."""
        import types
        import numpy as np
        A, Bm, lam1, lam2, weights, P1, P2, a1, ops1, ops2, k1, k2 = self._fit_inputs(batch, weights, k, orient_ops)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        B = x0.shape[0]
        opts = dict(self.LBFGS_REFERENCE)
        opts.update(lbfgs_options or {})
        if driver == "scipy":
            import scipy.optimize
            if B != 1:
                raise ValueError("driver='scipy' optimises one pair per call (the summed energy of a batch would couple the pairs)")
            cache = {}

            def fg(x):
                key = x.tobytes()
                if key not in cache:
                    cache.clear()
                    e, g = self.energy_grad(torch.from_numpy(x.reshape(B, k2, k1)), A, Bm, lam1, lam2, weights, P1, P2, a1, ops1, ops2)
                    cache[key] = (float(e.sum().item()), g.cpu().numpy().ravel())
                return cache[key]
            res = scipy.optimize.minimize(lambda x: fg(x)[0], x0.ravel(), jac=lambda x: fg(x)[1], method="L-BFGS-B",
                                          options={"maxiter": maxiter, **opts})
            return res.x.reshape(B, k2, k1), res
        n, m = k2 * k1, int(opts["maxcor"])
        if self.fit_fused_ok(k1, k2, weights, ops1) and fused:
            return self._fit_fused(A, Bm, lam1, lam2, weights, P1, P2, a1, x0, opts, maxiter, precision=precision)
        nbytes = int(self.lib.dm_lbfgs_state_bytes(B, n, m))
        state = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=self.device)
        xt = torch.empty((B, k2, k1), dtype=torch.float64, device=self.device)
        x0d = self._dev(x0, torch.float64, "x0")
        self._chk(self.lib.dm_lbfgs_init(self.ctx, B, n, m, _ptr(x0d), _ptr(state), _ptr(xt)))
        # the per-pair integer state (status first) sits behind the float64 blocks of the state buffer
        n_f64 = 3 * B * n + 2 * B * m * n + 2 * B * m + 16 * B
        ints = state[n_f64:n_f64 + (8 * B * 4 + 7) // 8].view(torch.int32)[:8 * B].view(B, 8)
        nev, maxfun = 0, int(opts["maxfun"])
        # the projected descriptors A, Bm are fixed during the fit: their Gram blocks are computed by the first evaluation only
        self.set_option("energy_keep_gram", 1)
        # the evaluation loop itself runs behind the ABI, `check_every` evaluations per call (dm_fmap_fit_steps): a Python round per
        # evaluation cost more host time than the evaluation takes on the device for one small pair
        ev = self._energy_grad_args(xt, A, Bm, lam1, lam2, weights, P1, P2, a1, ops1, ops2)
        energy = torch.empty((B,), dtype=torch.float64, device=self.device)
        grad = torch.empty((B, k2, k1), dtype=torch.float64, device=self.device)
        try:
            while True:
                self._chk(self.lib.dm_fmap_fit_steps(self.ctx, int(check_every), *ev[:-1], m, _ptr(state), _ptr(xt), _ptr(energy), _ptr(grad),
                                                     float(opts["ftol"]), float(opts["gtol"]), int(maxiter), maxfun, int(opts["maxls"])))
                nev += int(check_every)
                if bool((ints[:, 0] != 0).all()) or nev > maxfun + 2:        # (one host synchronisation per check_every evaluations)
                    break
        finally:
            self.set_option("energy_keep_gram", 0)
        xo = torch.empty((B, k2, k1), dtype=torch.float64, device=self.device)
        fo = torch.empty((B,), dtype=torch.float64, device=self.device)
        info = torch.empty((B, 4), dtype=torch.int32, device=self.device)
        self._chk(self.lib.dm_lbfgs_result(self.ctx, B, n, m, _ptr(state), _ptr(xo), _ptr(fo), _ptr(info)))
        info = info.cpu().numpy()
        info[:, 0] = np.where(info[:, 0] == 0, 4, info[:, 0])      # (the host loop ended at the evaluation limit: report it as such)
        res = types.SimpleNamespace(x=xo.cpu().numpy(), fun=fo.cpu().numpy(), status=info[:, 0].copy(), nit=info[:, 1].copy(),
                                    nfev=info[:, 2].copy(), message=[self.LBFGS_STATUS.get(int(q), "?") for q in info[:, 0]],
                                    success=bool(np.all((info[:, 0] == 1) | (info[:, 0] == 2))), evaluations=nev)
        return res.x, res

    def fm_to_p2p(self, Phi1, Phi2, a1, Cm, k1=None, k2=None, knn=True, ind=True):
        """Returns dict with knn21, knn12 (kd-tree maps of the reference) and ind21, ind12 (indicator arg-max)."""
        sfx, Phi1, Phi2, a1 = self._reals(Phi1, Phi2, a1)
        Cm = self._dev(Cm, torch.float64, "C")
        B, N1, ld1 = Phi1.shape
        _, N2, ld2 = Phi2.shape
        k2_, k1_ = Cm.shape[1], Cm.shape[2]
        if (k1 is not None and k1 != k1_) or (k2 is not None and k2 != k2_) or Cm.shape[0] != B:
            raise ValueError("fm_to_p2p: C must be (B,k2,k1)")
        if k1_ > ld1 or k2_ > ld2:
            raise AssertionError(f"At least {k1_}/{k2_} eigenvectors should be provided, here only {ld1}/{ld2} are given")
        mk = lambda n: torch.empty((B, n), dtype=torch.int32, device=self.device)
        out = {"knn21": mk(N2) if knn else None, "knn12": mk(N1) if knn else None,
               "ind21": mk(N2) if ind else None, "ind12": mk(N1) if ind else None}
        self._chk(getattr(self.lib, "dm_fm_to_p2p" + sfx)(self.ctx, B, N1, N2, k1_, k2_, _ptr(Phi1), ld1, _ptr(Phi2), ld2, _ptr(a1), _ptr(Cm),
                                        _ptr(out["knn21"]), _ptr(out["knn12"]), _ptr(out["ind21"]), _ptr(out["ind12"])))
        return out

    def p2p_to_fm(self, p21, Phi1, Phi2, a2, k1, k2):
        sfx, Phi1, Phi2, a2 = self._reals(Phi1, Phi2, a2)
        p21 = self._dev(p21, torch.int32, "p21")
        B, N1, ld1 = Phi1.shape
        _, N2, ld2 = Phi2.shape
        if p21.shape != (B, N2):
            raise ValueError("p2p_to_fm: p21 must be (B,N2)")
        Cm = torch.empty((B, k2, k1), dtype=torch.float64, device=self.device)
        self._chk(getattr(self.lib, "dm_p2p_to_fm" + sfx)(self.ctx, B, N1, N2, k1, k2, _ptr(p21), _ptr(Phi1), ld1, _ptr(Phi2), ld2, _ptr(a2), _ptr(Cm)))
        return Cm

    def tufted_covers(self, meshes, mollify_factor=1e-5, n_threads=0):
        """The tufted intrinsic-Delaunay cover of every (verts, faces) in `meshes` (dm_tufted_cover_batch: host C++ on a thread pool,
        like the reference's robust_laplacian wheel).  Returns [(T (2 nf, 3) int32, L (2 nf, 3) f64, flips, converged, eps)]."""
        import numpy as np
        cnt = len(meshes)
        V = [np.ascontiguousarray(v, dtype=np.float64) for v, _ in meshes]
        F = [np.ascontiguousarray(f, dtype=np.int32) for _, f in meshes]
        for v, f in zip(V, F):
            if v.ndim != 2 or v.shape[1] != 3 or f.ndim != 2 or f.shape[1] != 3 or f.shape[0] == 0:
                raise ValueError("tufted_covers: meshes are (verts (n, 3), faces (m, 3)) with at least one face")
        T = [np.empty((2 * f.shape[0], 3), np.int32) for f in F]
        L = [np.empty((2 * f.shape[0], 3), np.float64) for f in F]
        info = np.zeros((cnt, 2), np.int32)
        eps = np.zeros(cnt, np.float64)
        n = np.array([v.shape[0] for v in V], np.int32)
        nf = np.array([f.shape[0] for f in F], np.int32)
        ptrs = lambda arrs: C.cast((C.c_void_p * cnt)(*[a.ctypes.data for a in arrs]), C.c_void_p)
        rc = self.lib.dm_tufted_cover_batch(cnt, n.ctypes.data, nf.ctypes.data, ptrs(V), ptrs(F), float(mollify_factor), ptrs(T), ptrs(L),
                                            info.ctypes.data, eps.ctypes.data, int(n_threads))
        if rc != 0:
            raise ValueError("tufted cover: face indices must lie in [0, n)")
        return [(T[i], L[i], int(info[i, 0]), bool(info[i, 1]), float(eps[i])) for i in range(cnt)]

    def laplacian_ell(self, tris, lens=None, verts=None, scale=1.0, want_w=True):
        """Cotangent Laplacians of a batch of meshes assembled on the device (dm_laplacian_rows + dm_laplacian_ell).
        tris: list of (nt_b, 3) int arrays; lens: list of (nt_b, 3) intrinsic side lengths, or None with verts: list of (n_b, 3)
        coordinates (the reference's cotangent_weights / dia_area_mat arithmetic); meshes of different sizes are padded to the
        largest (padding vertices become decoupled rows at the top of the spectrum).  n_verts of a mesh = len(verts[b]) or, with
        lens, max index + 1 unless `verts` gives it.
        Returns dict(cols (B,N,nnz) int32, vals (B,N,nnz) f64 [A^-1/2 W A^-1/2], mass32 (B,N), w (B,N,nnz) f64 [W] or None,
        mass64 (B,N), nnz, n_verts (list))."""
        import numpy as np
        Bn = len(tris)
        if lens is None and verts is None:
            raise ValueError("laplacian_ell: pass intrinsic lengths or vertex coordinates")
        n_verts = [int(np.asarray(v).shape[0]) for v in verts] if verts is not None else [int(np.max(t)) + 1 for t in tris]
        N, nt = max(n_verts), max(int(np.asarray(t).shape[0]) for t in tris)
        tri_h = np.full((Bn, nt, 3), -1, np.int32)
        for b, t in enumerate(tris):
            tri_h[b, :len(t)] = t
        tri_d = torch.as_tensor(tri_h).to(self.device)
        len_d = vert_d = None
        if lens is not None:
            len_h = np.ones((Bn, nt, 3), np.float64)
            for b, l in enumerate(lens):
                len_h[b, :len(l)] = l
            len_d = torch.as_tensor(len_h).to(self.device)
        else:
            vert_h = np.zeros((Bn, N, 3), np.float64)
            for b, v in enumerate(verts):
                vert_h[b, :len(v)] = v
            vert_d = torch.as_tensor(vert_h).to(self.device)
        ragged = min(n_verts) < N
        nv_d = torch.as_tensor(np.asarray(n_verts, np.int32)).to(self.device) if ragged else None
        rows = torch.empty(int(self.lib.dm_laplacian_rows_bytes(Bn, N, nt)), dtype=torch.uint8, device=self.device)
        mx = C.c_int(0)
        self._chk(self.lib.dm_laplacian_rows(self.ctx, Bn, N, nt, _ptr(tri_d), _ptr(len_d), _ptr(vert_d), float(scale), _ptr(nv_d), _ptr(rows), C.byref(mx)))
        nnz = int(mx.value)
        cols = torch.empty((Bn, N, nnz), dtype=torch.int32, device=self.device)
        vals = torch.empty((Bn, N, nnz), dtype=torch.float64, device=self.device)
        mass32 = torch.empty((Bn, N), dtype=torch.float32, device=self.device)
        w = torch.empty((Bn, N, nnz), dtype=torch.float64, device=self.device) if want_w else None
        mass64 = torch.empty((Bn, N), dtype=torch.float64, device=self.device)
        self._chk(self.lib.dm_laplacian_ell(self.ctx, Bn, N, nt, _ptr(rows), nnz, _ptr(nv_d), _ptr(cols), _ptr(vals), _ptr(mass32), _ptr(w), _ptr(mass64)))
        return {"cols": cols, "vals": vals, "mass32": mass32, "w": w, "mass64": mass64, "nnz": nnz, "n_verts": n_verts}

    def eigenbasis(self, W_list, mass, k, guard=None, degree=30, tol=1e-9, max_rounds=12, seed=0, ell=None):
        """k smallest eigenpairs of W phi = lambda A phi for a batch of meshes (reference TriMesh.process ->
        laplacian_spectrum: ARPACK on the host, one mesh at a time).
        ell = the dict laplacian_ell returns (operands already on the device; W_list / mass are ignored), or
        W_list: sparse stiffness matrices (host, SciPy) with mass (B,N) lumped masses, or a list of 1-D arrays when the vertex
        counts differ (Phi is then padded to the largest) -- the ELL layout of A^-1/2 W A^-1/2 is then built on the host.
        The iteration runs on the GPU (dm_eigenbasis) until max_j |L x_j - lam_j x_j| <= tol * lam_k.  Meshes too small for the
        filtered subspace iteration (2 (k + guard) > N) take the dense route: the whole space as the subspace, one Rayleigh-Ritz
        step (a Jacobi eigensolve of the N x N operator, N <= 2048: one workgroup per mesh, ~0.3 s at N = 1024).
        Returns (lam (B,k) f64, Phi (B,N,k) f64, resid (B,), rounds)."""
        import numpy as np
        import scipy.sparse as sp
        # (dm_eigenbasis takes the lumped masses as fp32: the problem solved is the one with the ROUNDED masses, and Phi^T A Phi = I
        #  holds for those.  The float64 entry points downstream receive the caller's unrounded mesh.A: the basis is orthonormal
        #  for a mass vector that differs from theirs by <= 6e-8 relative -- far inside the 1e-4 bar on C, and the reason the
        #  eigenbasis tests compare with SciPy on the rounded masses.)
        # Meshes of DIFFERENT vertex counts share a call too: the smaller ones are padded with decoupled vertices whose only entry
        # is a diagonal one AT the Gershgorin bound of the mesh's own operator (max_i sum_q |L_iq| >= lambda_max: the upper end of
        # the interval the Chebyshev filter damps), so the spurious eigenvalue can never fall inside the wanted part of the
        # spectrum.  Their rows of Phi come back ~0 and the caller slices them off.
        if guard is None:
            # guard vectors: 12 .. 32, so that the block k + guard is a multiple of 32 columns where that is possible -- the sparse
            # product gathers whole rows of the block, and 32 doubles are two aligned 128-byte lines (measured, 128 meshes of 2048
            # vertices, k = 20: 79 ms with 32 guard vectors = 52 columns, 46 ms with 12 = 32 columns, one more round, same eigenpairs
            # to 1e-15: tools/eig_sweep.py)
            guard = next((g for g in range(12, 33) if (k + g) % 32 == 0), 32)
        if ell is not None:
            cols_d, vals_d, mass_d, nnz = ell["cols"], ell["vals"], ell["mass32"], int(ell["nnz"])
            B, N = mass_d.shape
            nmin = min(ell["n_verts"])
        else:
            masses = [np.ascontiguousarray(a, dtype=np.float32).astype(np.float64).ravel() for a in mass]
            B, N = len(masses), max(a.shape[0] for a in masses)
            if any(np.any(a <= 0) for a in masses):
                raise ValueError("eigenbasis: every vertex needs a positive lumped mass (isolated or degenerate vertices?)")
            mats = []
            for b in range(B):
                d = sp.diags(1.0 / np.sqrt(masses[b]))
                mats.append((d @ sp.csr_matrix(W_list[b]) @ d).tocsr())
            nnz = max(int(np.diff(Lm.indptr).max()) for Lm in mats)
            cols = np.tile(np.arange(N, dtype=np.int32)[None, :, None], (B, 1, nnz))
            vals = np.zeros((B, N, nnz))
            mass_h = np.ones((B, N))
            for b, Lm in enumerate(mats):
                nb = Lm.shape[0]
                cnt = np.diff(Lm.indptr)
                pos = np.arange(Lm.nnz) - np.repeat(Lm.indptr[:-1], cnt)
                rows = np.repeat(np.arange(nb), cnt)
                cols[b, rows, pos] = Lm.indices
                vals[b, rows, pos] = Lm.data
                mass_h[b, :nb] = masses[b]
                if nb < N:
                    vals[b, nb:, 0] = float(abs(Lm).sum(axis=1).max())
            nmin = min(a.shape[0] for a in masses)
            cols_d = torch.as_tensor(cols).to(self.device)
            vals_d = torch.as_tensor(vals).to(self.device)
            mass_d = torch.as_tensor(mass_h.astype(np.float32)).to(self.device)
        lam = torch.empty((B, k), dtype=torch.float64, device=self.device)
        Phi = torch.empty((B, N, k), dtype=torch.float64, device=self.device)
        resid = torch.empty((B,), dtype=torch.float64, device=self.device)
        if k > nmin:
            raise ValueError(f"eigenbasis: {k} eigenpairs asked of a mesh with {nmin} vertices")
        if 2 * (k + guard) > nmin:
            # (measured: with the wanted range reaching into the upper half of a spectrum the filtered block loses rank and the Ritz step
            #  returns spurious zero pairs whose residual looks converged.)  Dense route: X = I, no filter -- the Rayleigh-Ritz step IS the
            # eigendecomposition of L; the padding rows of a ragged batch sit at their Gershgorin bound, above every wanted pair.
            if N > 2048:
                raise ValueError(f"eigenbasis: k + guard = {k + guard} vectors need a mesh of at least {2 * (k + guard)} vertices "
                                 f"(the smallest has {nmin}); the dense route takes meshes up to 2048 vertices, this batch has {N}")
            X = torch.eye(N, dtype=torch.float64, device=self.device).repeat(B, 1, 1).contiguous()
            self._chk(self.lib.dm_eigenbasis(self.ctx, B, N, nnz, _ptr(cols_d), _ptr(vals_d), _ptr(mass_d), k, N - k, 1, 2, 2,
                                             _ptr(X), _ptr(lam), _ptr(Phi), _ptr(resid)))
            return lam, Phi, resid, 0
        m = min(k + guard, N)
        # A mesh's result must not depend on the batch it is solved in: its start block is drawn for ITS vertex count (one draw per
        # distinct count, the same seed; padding rows start at zero), and its outputs are latched the first time its own residual
        # passes -- the batch iterates on until the last mesh has, but a mesh that was done keeps what it had.
        nvs = ell["n_verts"] if ell is not None else [a.shape[0] for a in masses]
        X = torch.zeros((B, N, m), dtype=torch.float64, device=self.device)
        draws = {}
        for b, nb in enumerate(nvs):
            if nb not in draws:
                g = torch.Generator(device=self.device).manual_seed(seed)
                draws[nb] = torch.randn((nb, m), dtype=torch.float64, device=self.device, generator=g)
            X[b, :nb] = draws[nb]
        lam_w, Phi_w, resid_w = torch.empty_like(lam), torch.empty_like(Phi), torch.empty_like(resid)
        done = torch.zeros((B,), dtype=torch.bool, device=self.device)
        rounds = 0
        for rounds in range(1, max_rounds + 1):
            n_iter = 5 if rounds == 1 else 2
            self._chk(self.lib.dm_eigenbasis(self.ctx, B, N, nnz, _ptr(cols_d), _ptr(vals_d), _ptr(mass_d), k, m - k, n_iter, degree,
                                             0 if rounds == 1 else 1, _ptr(X), _ptr(lam_w), _ptr(Phi_w), _ptr(resid_w)))
            scale = torch.clamp(lam_w[:, -1].abs(), min=1e-300)
            ok = resid_w <= tol * scale
            take = ~done if rounds == max_rounds else (ok & ~done)        # (the last round: whatever is there, the caller judges the residual)
            if B == 1:
                if bool(take[0]):
                    lam, Phi, resid = lam_w, Phi_w, resid_w
            else:
                lam[take], Phi[take], resid[take] = lam_w[take], Phi_w[take], resid_w[take]
            done |= ok
            if bool(done.all()):
                break
        return lam, Phi, resid, rounds

    def scratch(self, name, shape, dtype):
        """a device tensor that belongs to this engine and is handed out again by the next call with the same name, shape and dtype: for
        large intermediates of a call that never reach the user (a gigabyte of precise maps per chunk: allocated per call, the caching
        allocator of a process with other tensors around kept going back to hipMalloc, which stalls every stream)"""
        cache = self.__dict__.setdefault("_scratch", {})
        t = cache.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(tuple(shape), dtype=dtype, device=self.device)
            cache[name] = t
        return t

    def precise_map(self, Phi1, Phi2, Cm, faces1, dense=False, scratch=False):
        """Barycentric projection of every vertex of mesh 2 onto the faces of mesh 1 in the spectral embedding (reference
        get_precise_map, functional.py:221-251).  Returns (face_match (B,N2) int32, bary (B,N2,3) f64[, dense (B,N2,N1) f64])."""
        sfx, Phi1, Phi2 = self._reals(Phi1, Phi2)
        Cm = self._dev(Cm, torch.float64, "C")
        faces1 = self._dev(faces1, torch.int32, "faces1")
        B, N1, ld1 = Phi1.shape
        _, N2, ld2 = Phi2.shape
        k2, k1 = Cm.shape[1], Cm.shape[2]
        nf = faces1.shape[1]
        if faces1.shape != (B, nf, 3) or Cm.shape[0] != B:
            raise ValueError("precise_map: faces1 must be (B,nf,3) and C (B,k2,k1)")
        if nf and (int(faces1.min()) < 0 or int(faces1.max()) >= N1):
            raise ValueError("precise_map: face indices must lie in [0, N1)")
        fm = torch.empty((B, N2), dtype=torch.int32, device=self.device)
        bary = torch.empty((B, N2, 3), dtype=torch.float64, device=self.device)
        # (scratch=True: the dense matrices live in this engine's scratch and are overwritten by its next such call)
        M = (self.scratch("precise_dense", (B, N2, N1), torch.float64) if scratch else
             torch.empty((B, N2, N1), dtype=torch.float64, device=self.device)) if dense else None
        info = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._chk(getattr(self.lib, "dm_precise_map" + sfx)(self.ctx, B, N1, N2, k1, k2, nf, _ptr(Phi1), ld1, _ptr(Phi2), ld2, _ptr(Cm), _ptr(faces1),
                                          _ptr(fm), _ptr(bary), _ptr(M), _ptr(info)))
        return (fm, bary, M) if dense else (fm, bary)

    def linear_sum_assignment(self, cost, maximize=False, defer=False):
        """Optimal assignment of every matrix of the batch `cost` (B,nr,nc) f64 -> col_of_row (B,nr) int32, -1 = unassigned
        (identical to scipy.optimize.linear_sum_assignment, reference functional_map.py:57,66,78).
        defer=True: the launches go out and a function is returned that waits for them, raises SciPy's errors and hands back the result
        (a search is a single workgroup busy for milliseconds: a caller with other work for the GPU queues it on another stream meanwhile)."""
        cost = self._dev(cost, torch.float64, "cost")
        if cost.dim() != 3:
            raise ValueError("linear_sum_assignment expects (B,nr,nc)")
        B, nr, nc = cost.shape
        out = torch.empty((B, nr), dtype=torch.int32, device=self.device)
        info = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._chk(self.lib.dm_linear_sum_assignment(self.ctx, B, nr, nc, _ptr(cost), 1 if maximize else 0, _ptr(out), _ptr(info)))

        def finish(keep=cost):                               # (`keep`: the matrix stays alive until the search has read it)
            worst = int(info.max())                          # (the caller reads the result next: this synchronisation is not extra)
            if worst == 2:
                raise ValueError("matrix contains invalid numeric entries")      # SciPy's messages
            if worst == 1:
                raise ValueError("cost matrix is infeasible")
            return out
        return finish if defer else finish()

    def lsa_indicator_ok(self, N1, N2, k1, k2):
        return bool(self.lib.dm_lsa_indicator_ok(self.ctx, int(N1), int(N2), int(k1), int(k2)))

    def lsa_indicator(self, Phi1, Phi2, a1, Cm, dense=None, maximize=True, defer=False):
        """Optimal assignments of the mapped indicators Phi2 C Phi1^T diag(a1) given by their factors (no N2 x N1 matrix is formed: the
        kernel evaluates cost rows from the factors with dm_mapped_indicator's arithmetic, bit for bit) and of the dense (n, N2, N1)
        matrices `dense` in the same launch.  Returns col_of_row (n_ind + n_dense, N2) int32, the indicators first.
        (reference functional_map.py:57, 66, 78: scipy.optimize.linear_sum_assignment(..., maximize=True))"""
        sfx, Phi1, Phi2, a1 = self._reals(Phi1, Phi2, a1)
        Cm = self._dev(Cm, torch.float64, "C")
        B, N1, ld1 = Phi1.shape
        _, N2, ld2 = Phi2.shape
        k2, k1 = Cm.shape[1], Cm.shape[2]
        nd = 0
        if dense is not None:
            dense = self._dev(dense, torch.float64, "dense")
            if dense.dim() != 3 or dense.shape[1:] != (N2, N1):
                raise ValueError("lsa_indicator: dense matrices must be (n, N2, N1)")
            nd = dense.shape[0]
        out = torch.empty((B + nd, N2), dtype=torch.int32, device=self.device)
        info = torch.empty((B + nd,), dtype=torch.int32, device=self.device)
        self._chk(getattr(self.lib, "dm_lsa_indicator" + sfx)(self.ctx, B, N1, N2, k1, k2, _ptr(Phi1), ld1, _ptr(Phi2), ld2, _ptr(a1), _ptr(Cm), nd, _ptr(dense),
                                                             1 if maximize else 0, _ptr(out), _ptr(info)))

        def finish(keep=(Phi1, Phi2, a1, Cm, dense)):        # (defer=True: see linear_sum_assignment)
            worst = int(info.max())
            if worst == 2:
                raise ValueError("matrix contains invalid numeric entries")
            if worst == 1:
                raise ValueError("cost matrix is infeasible")
            return out
        return finish if defer else finish()

    def p2p_to_fm_lstsq(self, p21, Phi1, Phi2, k1, k2):
        """argmin_X |Phi2[:, :k2] X - Phi1[p21, :k1]|_F (reference convert.py:51, no mass matrix) -> (B,k2,k1) f64."""
        sfx, Phi1, Phi2 = self._reals(Phi1, Phi2)
        p21 = self._dev(p21, torch.int32, "p21")
        B, N1, ld1 = Phi1.shape
        _, N2, ld2 = Phi2.shape
        if p21.shape != (B, N2):
            raise ValueError("p2p_to_fm_lstsq: p21 must be (B,N2)")
        Cm = torch.empty((B, k2, k1), dtype=torch.float64, device=self.device)
        info = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._chk(getattr(self.lib, "dm_p2p_to_fm_lstsq" + sfx)(self.ctx, B, N1, N2, k1, k2, _ptr(p21), _ptr(Phi1), ld1, _ptr(Phi2), ld2, _ptr(Cm),
                                              _ptr(info)))
        bad = torch.nonzero(info).flatten()
        if bad.numel():
            raise _lib.DenseMatchError(f"least-squares map: Phi2^T Phi2 is not positive definite for pairs {bad.tolist()[:8]}")
        return Cm

    def zoomout(self, Phi1, Phi2, a2, C0, nit, step=1, return_p2p=False):
        sfx, Phi1, Phi2, a2 = self._reals(Phi1, Phi2, a2)
        C0 = self._dev(C0, torch.float64, "C0")
        B, N1, ld1 = Phi1.shape
        _, N2, ld2 = Phi2.shape
        if C0.dim() != 3 or C0.shape[0] != B or C0.shape[1] != C0.shape[2]:
            raise ValueError("zoomout: C0 must be (B,k0,k0)")
        k0 = C0.shape[1]
        kf = k0 + nit * step
        assert kf <= ld1, f"Not enough eigenvectors on source : {kf} are needed when {ld1} are provided"
        assert kf <= ld2, f"Not enough eigenvectors on target : {kf} are needed when {ld2} are provided"
        Cout = torch.empty((B, kf, kf), dtype=torch.float64, device=self.device)
        p21 = torch.empty((B, N2), dtype=torch.int32, device=self.device) if return_p2p else None
        self._chk(getattr(self.lib, "dm_zoomout" + sfx)(self.ctx, B, N1, N2, k0, nit, step, _ptr(Phi1), ld1, _ptr(Phi2), ld2, _ptr(a2),
                                      _ptr(C0), _ptr(Cout), _ptr(p21)))
        return (Cout, p21) if return_p2p else Cout

    def icp(self, Phi1, Phi2, C0, nit=10, return_resid=False):
        """Spectral ICP (reference pyFM/refine/icp.py) -> C (B,k2,k1) f64 with orthonormal columns."""
        sfx, Phi1, Phi2 = self._reals(Phi1, Phi2)
        C0 = self._dev(C0, torch.float64, "C0")
        B, N1, ld1 = Phi1.shape
        _, N2, ld2 = Phi2.shape
        k2, k1 = C0.shape[1], C0.shape[2]
        Cout = torch.empty_like(C0)
        resid = torch.empty((B,), dtype=torch.float64, device=self.device)
        info = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._chk(getattr(self.lib, "dm_icp" + sfx)(self.ctx, B, N1, N2, k1, k2, _ptr(Phi1), ld1, _ptr(Phi2), ld2, _ptr(C0), int(nit), _ptr(Cout),
                                  _ptr(resid), _ptr(info)))
        return (Cout, resid, info) if return_resid else Cout

    def knn_query(self, X, Y):
        """For every row of Y the index of the nearest row of X (float64, exact, lowest index on ties)."""
        X = self._dev(X, torch.float64, "X")
        Y = self._dev(Y, torch.float64, "Y")
        if X.dim() != 3 or Y.dim() != 3 or X.shape[0] != Y.shape[0] or X.shape[2] != Y.shape[2]:
            raise ValueError("knn_query expects X (B,nx,p) and Y (B,ny,p)")
        B, nx, p = X.shape
        ny = Y.shape[1]
        out = torch.empty((B, ny), dtype=torch.int32, device=self.device)
        self._chk(self.lib.dm_knn_query_f64(self.ctx, B, nx, ny, p, _ptr(X), _ptr(Y), _ptr(out)))
        return out

    def knn_query_topk(self, X, Y, k):
        """the k nearest rows of X for every row of Y, nearest first -> (idx (B,ny,k) int32, dist (B,ny,k) f64)"""
        X = self._dev(X, torch.float64, "X")
        Y = self._dev(Y, torch.float64, "Y")
        B, nx, p = X.shape
        ny = Y.shape[1]
        idx = torch.empty((B, ny, k), dtype=torch.int32, device=self.device)
        dist = torch.empty((B, ny, k), dtype=torch.float64, device=self.device)
        self._chk(self.lib.dm_knn_query_topk_f64(self.ctx, B, nx, ny, p, k, _ptr(X), _ptr(Y), _ptr(idx), _ptr(dist)))
        return idx, dist

    def mapped_indicator(self, Phi1, Phi2, a1, Cm):
        """Dense (B,N2,N1) float64 indicator ((Phi2 C) Phi1^T) * a1 -- only for callers that want the matrix."""
        sfx, Phi1, Phi2, a1 = self._reals(Phi1, Phi2, a1)
        Cm = self._dev(Cm, torch.float64, "C")
        B, N1, ld1 = Phi1.shape
        _, N2, ld2 = Phi2.shape
        k2, k1 = Cm.shape[1], Cm.shape[2]
        M = torch.empty((B, N2, N1), dtype=torch.float64, device=self.device)
        self._chk(getattr(self.lib, "dm_mapped_indicator" + sfx)(self.ctx, B, N1, N2, k1, k2, _ptr(Phi1), ld1, _ptr(Phi2), ld2, _ptr(a1),
                                               _ptr(Cm), _ptr(M)))
        return M

    # ------------------------------------------------------------------ the hot path, one batch
    def match(self, batch, k=None, w_descr=1e4, w_lap=1e3, knn=True, ind=True, check=False):
        """project -> pinned column -> solve -> vertex maps for a batch of pairs (BASELINE config 2).

        `batch` is a dict with Phi1, Phi2, lam1, lam2, a1, a2, F1, F2 (device tensors).
        Returns dict(C, knn21, knn12, ind21, ind12).  No host synchronisation inside
        (unless check=True)."""
        Phi1, Phi2 = batch["Phi1"], batch["Phi2"]
        k1 = k if k is not None else batch["lam1"].shape[1]
        k2 = k if k is not None else batch["lam2"].shape[1]
        lam1 = batch["lam1"][:, :k1].contiguous()
        lam2 = batch["lam2"][:, :k2].contiguous()
        if batch["F1"].dtype == torch.float16 and batch["F2"].dtype == torch.float16:
            Cm = self.fmap_fit(Phi1, Phi2, batch["a1"], batch["a2"], batch["F1"], batch["F2"], lam1, lam2, w_descr, w_lap, k1, k2, check=check)
        else:
            A = self.project(Phi1, batch["a1"], batch["F1"], k1)
            Bm = self.project(Phi2, batch["a2"], batch["F2"], k2)
            c00 = self.c00(Phi1, Phi2, batch["a1"], batch["a2"])
            Cm = self.fmap_solve(A, Bm, lam1, lam2, c00, w_descr, w_lap, check=check)
        out = self.fm_to_p2p(Phi1, Phi2, batch["a1"], Cm, knn=knn, ind=ind)
        out["C"] = Cm
        return out


_default_engines = {}


def default_engine(device=None):
    """One engine per (device, current stream), created on first use."""
    if not torch.cuda.is_available():
        raise RuntimeError("densematcher_amd needs a ROCm GPU (gfx950); there is no CPU fallback")
    dev = torch.cuda.current_device() if device is None else (device if isinstance(device, int) else torch.device(device).index or 0)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _default_engines:
        _default_engines[key] = MatchEngine(dev)
    return _default_engines[key]

/*
 * densematch.h -- C ABI of libdensematch (MI355X / gfx950 native).
 *
 * The drop-in boundary for DenseMatcher's matching hot path.  The reference
 * has no FFI on this path: the seam is the Python call level
 * (densematcher/functional_map.py:9 `compute_surface_map` and the
 * densematcher/pyFM functions it calls).  Each entry point below replaces the
 * arithmetic of one of those reference functions; the Python mirror in
 * `densematcher_amd/` keeps the reference signatures and binds these symbols
 * with ctypes (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.
 *   - every array argument is a DEVICE pointer owned by the caller (hipMalloc /
 *     torch tensor.data_ptr()), row-major, batch-major, contiguous unless an
 *     `ld` (row stride in elements) is given.  The library never frees or
 *     retains caller pointers past return.
 *   - B = number of mesh pairs in the batch.  Pairs are independent.
 *   - mesh 1 = source, mesh 2 = target.  Phi1 (N1 x ld1) / Phi2 (N2 x ld2) are
 *     mass-orthonormal Laplace-Beltrami eigenvectors, mass1/mass2 the diagonals
 *     of the lumped mass matrices, C is (k2 x k1) and maps basis-1
 *     coefficients to basis 2 (reference convention, pyFM/functional.py:482).
 *   - eigenvectors and masses are fp32 in the plain entry points and fp64 in the `*_f64` ones (same argument lists,
 *     `const double*` for Phi1 / Phi2 / mass): float64 is what the reference holds (pyFM/mesh/trimesh.py:118 float64
 *     vertices -> float64 spectrum) and what its FM_to_p2p / p2p_to_FM / ICP / ZoomOut consume
 *     (pyFM/spectral/convert.py:134-144), so the integer outputs of the `*_f64` forms equal the reference's on ITS inputs;
 *     fp32 halves the operand traffic and is exact for bases that were rounded to fp32 anyway.
 *   - integer maps are int32 on the device (the Python layer widens to int64).
 *   - calls are asynchronous on the context's stream; the caller synchronises
 *     the stream before reading results.  A context is not thread-safe;
 *     contexts are independent of one another.
 *   - return value: DM_OK or a negative dm_status; no exceptions cross the ABI.
 */
#ifndef DENSEMATCH_H
#define DENSEMATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dm_ctx dm_ctx;

typedef enum dm_status {
    DM_OK = 0,
    DM_EINVAL = -1,     /* bad argument (shape, null pointer, unsupported size)  -> Python ValueError   */
    DM_ENOMEM = -2,     /* workspace allocation failed                           -> Python MemoryError  */
    DM_EHIP = -3,       /* a HIP runtime call failed (see dm_last_error)         -> Python RuntimeError */
    DM_ESINGULAR = -4   /* a linear system was not positive definite (info[] says which pair) */
} dm_status;

/* feature dtypes for dm_project; OR in DM_PROJECT_F64 to force the float64 matrix-core path */
enum { DM_F16 = 0, DM_F32 = 1, DM_PROJECT_F64 = 0x10 };

/* ---- context ---------------------------------------------------------- */
/* One context per (device, stream).  hip_stream may be NULL (default stream). */
int dm_create(int device, void* hip_stream, dm_ctx** out);
int dm_destroy(dm_ctx* ctx);
/* ctx-owned string describing the last failure; valid until the next call on ctx. */
const char* dm_last_error(const dm_ctx* ctx);
const char* dm_version(void);
/* bytes of ctx-owned scratch currently allocated (grown lazily, never shrunk). */
size_t dm_workspace_bytes(const dm_ctx* ctx);
/* Choose between equivalent code paths (every value of every option returns the same, exact results; the
 * defaults are the fast paths).  No reference counterpart: the reference has one CPU path.  Options:
 *   "simnn_pipe"    1 | 0   feature-similarity tiles: LDS-DMA ring kernel | bounds-checked register-staged kernel
 *   "simnn_persist" 1 | 0 | n>1   one persistent workgroup per CU walking its tiles | one workgroup per tile | exactly n workgroups
 *   "knn_split"     1 | 0   knn21 of dm_zoomout / dm_icp / dm_knn_query_f64: fp16-split first pass | float64 G kernel
 *   "p2p_split" 2 | 3 | 4 | 1 | 0 dm_fm_to_p2p: one fp16 pass reducing in both directions, tile shape by size (4 waves, 128 x 256
 *                           tiles, two workgroups per CU while a pair's operands fit an XCD's L2, else 8 waves, 256 x 256) |
 *                           always the 4-wave shape | always the 8-wave shape | two two-key passes | float64 G kernel
 *                           (all + exact float64 re-evaluation of the ambiguous rows; identical results)
 *   "solve_packed"  0 | 1   dm_fmap_solve: blocked LDS Cholesky when it fits | packed-storage solver always
 *   "simnn_band"    4 | n   tile order of the similarity kernels: bands of n tile rows, column-major inside (0: row-major)
 *   "simnn_big"     0 | 1   tile kernels as four waves of 128 x 128 (accumulators in AGPRs) instead of eight waves of 128 x 64
 *   "lsa_reg"       2 | 1 | 0   dm_linear_sum_assignment: column state in registers, started from a column reduction (kept per
 *                           matrix only when its optimum is provably unique, else redone) | the same in SciPy's order | LDS state
 *   "solve_reg"     1 | 0   dm_fmap_solve, k1 <= 129: register-resident solver (one wave per system) | the LDS-resident blocked one
 *   "energy_keep_gram" 0 | 1   dm_fmap_energy_grad: 1 = keep what does not depend on C (A A^T, B A^T, the mass-weighted column sums
 *                           of the bases) from the next call and reuse it while the same A, B, Phi1, Phi2, mass1 pointers and sizes are
 *                           passed; the caller promises not to change their CONTENTS meanwhile (an optimiser's evaluations of one fit).  Setting the option (to any value) drops what is kept.
 *   "zoomout_fused" 1 | 0   dm_zoomout on meshes of at least 256 vertices, maps up to 208: five launches per iteration (embedding + split
 *                           rows, biased-key search, merge, exact, p2p_to_FM) | six (seven with the reduce) through K-major copies
 *   "p2pfm_direct"  1 | 0   dm_p2p_to_fm (and the p2p_to_FM steps of dm_zoomout / dm_icp): register-resident tiles, operands straight
 *                           from global memory, fixed-order in-workgroup reduction | LDS-staged 64 x 64 tiles + split-K partials + reduce
 *   "simnn1_wt"     4 | 2   tile shape of the fused ZoomOut search: 8 waves, 256 x 256 | 4 waves, 128 x 256 (two workgroups per CU)
 *   "solve_pcg"     1 | 0   dm_fmap_solve / dm_fmap_fit, 66 <= k1 <= 200 (any batch: the choice follows the sizes only, a pair's result
 *                           does not depend on its batch): the k2 systems of a pair by a batched Jacobi-preconditioned
 *                           conjugate-gradient iteration on the float64 matrix cores (stops at a 1e-11 relative reduction: C within 1e-9 of
 *                           the direct solution; ill-conditioned pairs leave it after six steps and take the direct solver) | direct
 *                           solvers only.  The two settings agree to 1e-9, not bit for bit.
 *   "basis_stats"   1 | 0   dm_fm_to_p2p: keep the maximum of |Phi2| per basis tensor between calls (a checked hint: the target operand's
 *                           fp16 rows are then written while the basis streams through the second embedding) | every call takes its own
 *                           pass over the basis.  Same results.
 *   "fit_f32"       0 | 1   dm_fmap_fit_fused: the element loop over the N2 x N1 entries of the mapped indicator in float64 | in fp32,
 *                           the precision the reference evaluates these terms in (pyFM/functional.py:379-383).  The ONE option whose two
 *                           settings differ in the result: energy / gradient within 1e-6 relative, the fitted map within 3e-6 under
 *                           SciPy's stopping rule.  The Python layer chooses it from the stopping rule (engine.py: _fit_fused).
 *   "fit_mfma"      1 | 0   the fp32 element loop for maps up to 16 x 16: the two 16-deep products of an entry (the entry itself and the
 *                           back-product of its derivative) on v_mfma_f32_16x16x4_f32, the element-wise terms alone on the vector ALU | both
 *                           on the packed vector FMA.  Different summation orders of the same fp32 arithmetic: both within 1e-7 (energy) /
 *                           1e-6 (gradient) of the float64 oracle, fitted maps within 1e-6 of each other.
 * Unknown names return DM_EINVAL.  The library never reads environment variables. */
int dm_set_option(dm_ctx* ctx, const char* name, int value);
/* Diagnostics of the LAST dm_fm_to_p2p[_f64] / dm_simnn_f16 call on ctx (synchronises the stream): out[q] = rows of reduction q
 * whose fp32 margin fell inside the error bound and were re-evaluated in float64 (q = knn21 | ind21 | knn12 | ind12 for the four
 * maps, q = 0 for dm_simnn_f16; -1 where the call ran the float64 kernel).  bench.py reports them for its hard input
 * distributions: the cost of the exact pass depends on the data.  No reference counterpart. */
int dm_last_requeued_rows(dm_ctx* ctx, int out[4]);

/* ---- kernel timing (HIP events on the ctx stream) ----------------------- */
/* Bracket every launch of the kernel called `name` (see DESIGN.md for names)
 * with a hipEvent pair.  Pass NULL to stop.  dm_profile_read synchronises the
 * stream and returns the number of launches and their summed duration since
 * the last dm_profile_kernel call. */
int dm_profile_kernel(dm_ctx* ctx, const char* name);
int dm_profile_read(dm_ctx* ctx, int* launches, double* total_ms);
/* dm_profile_kernel(ctx, "*") brackets EVERY launch; dm_profile_report then writes one line per kernel name,
 * "name\tlaunches\ttotal_ms\n" in order of first launch, into buf (NUL-terminated; DM_EINVAL if cap is too small) and
 * resets the record.  (The event pairs add a few microseconds between launches: bench.py times its K steps with one kernel
 * bracketed and collects the per-kernel table in a separate, untimed pass.) */
int dm_profile_report(dm_ctx* ctx, char* buf, size_t cap);

/* ---- on-box peak probes (bench.py: measured peak beside the spec peak of the roofline block) -------------------
 * value = FLOP/s (matrix-core probes) or bytes/s read + written (copy).  The fp16 probe comes in two flavours because the
 * part is power-limited: all-zero operands run at the full 2.4 GHz (the instruction-issue ceiling, ~2.48 PFLOP/s), N(0,1)
 * operands pull the shader clock to ~1.7 GHz (~1.72 PFLOP/s): the ceiling of a kernel that multiplies real data. */
enum { DM_PEAK_MFMA_F16_ZERO = 0, DM_PEAK_MFMA_F16_RANDOM = 1, DM_PEAK_MFMA_F64 = 2, DM_PEAK_HBM_COPY = 3 };
int dm_measure_peak(dm_ctx* ctx, int which, double* value);

/* ---- config 3: feature-similarity nearest neighbour ---------------------
 * nn21[b,i] = argmax_j <Ftgt[b,i,:], Fsrc[b,j,:]>   (lowest j on ties)
 * No reference symbol; defined by oracle/dm_oracle.py:simnn.
 * The products are exact (fp16 x fp16 in fp32), accumulated in fp32 on the
 * matrix cores; rows whose top-2 margin is within the fp32 accumulation bound
 * are re-evaluated in float64, so nn21 equals the float64 argmax.
 * Ftgt (B,N2,D), Fsrc (B,N1,D) fp16.  best/margin (B,N2) fp32 are optional
 * (fp32-accumulated best score and best - second best). */
int dm_simnn_f16(dm_ctx* ctx, int B, int N2, int N1, int D,
                 const void* Ftgt, const void* Fsrc,
                 int32_t* nn21, float* best /*nullable*/, float* margin /*nullable*/);

/* ---- spectral projection ------------------------------------------------
 * Ared[b] = Phi[b][:, :k]^T (mass[b] * F[b])        (k x D), fp32 out.
 * Replaces pyFM/optimize/base_functions.py:526-532 (descr{1,2}_red) and
 * pyFM/mesh/trimesh.py:533-556 (TriMesh.project).
 * Phi (B,N,ld) fp32, mass (B,N) fp32, F (B,N,D) fp16 or fp32 (f_dtype).
 * fp16 descriptors run on the fp16 matrix cores with the basis split into two fp16 pieces (relative error
 * ~1e-6, the class of the reference's own fp32 projection); fp32 descriptors, or f_dtype | DM_PROJECT_F64,
 * run on the float64 matrix cores (error = the final fp32 rounding only). */
int dm_project(dm_ctx* ctx, int B, int N, int D, int k,
               const float* Phi, int ld, const float* mass,
               const void* F, int f_dtype, float* Ared /* B*k*D */);
/* float64 basis and masses: rounded to fp32 as they are loaded by the fp16-split path (exactly what the reference's fit does
 * before it projects, pyFM/functional.py:410-414: same result as converting first, without the extra pass over Phi); the
 * DM_PROJECT_F64 path uses them unrounded (TriMesh.project, pyFM/mesh/trimesh.py:533-556, is float64 NumPy). */
int dm_project_f64(dm_ctx* ctx, int B, int N, int D, int k,
                   const double* Phi, int ld, const double* mass,
                   const void* F, int f_dtype, float* Ared /* B*k*D */);

/* ---- pinned first column ---------------------------------------------------
 * c00[b] = sign(Phi1[b][0,0] * Phi2[b][0,0]) * sqrt(sum(mass2[b]) / sum(mass1[b]))
 * Replaces FunctionalMapping.get_x0 (pyFM/functional.py:654-658; mesh.area =
 * A.sum(), pyFM/mesh/trimesh.py:206-221): the only non-zero entry of column 0
 * of the initial map, which the optimiser never changes (base_functions.py:759). */
int dm_fmap_c00(dm_ctx* ctx, int B, int N1, int N2,
                const float* Phi1, int ld1, const float* Phi2, int ld2,
                const float* mass1, const float* mass2, double* c00 /* B */);
int dm_fmap_c00_f64(dm_ctx* ctx, int B, int N1, int N2,
                    const double* Phi1, int ld1, const double* Phi2, int ld2,
                    const double* mass1, const double* mass2, double* c00 /* B */);

/* ---- functional-map solve -----------------------------------------------
 * Minimiser of  w_descr/2 |C A - Bm|^2 + w_lap/2 sum C_ij^2 ev_ij  with column
 * 0 pinned to (c00, 0, ..., 0)^T, ev_ij = ((lam1_j - lam2_i)/max(lam))^2.
 * Replaces the L-BFGS-B loop of pyFM/functional.py:352-487 over
 * pyFM/optimize/base_functions.py:480-763 (w_descr / w_lap terms), i.e.
 * FunctionalMapping.fit; c00 is get_x0()[0,0] (functional.py:654-658).
 * A (B,k1,D), Bm (B,k2,D) fp32; lam1 (B,k1), lam2 (B,k2), c00 (B) fp64;
 * C (B,k2,k1) fp64 out; info (B) int32 out, 0 = ok, r+1 = row r not SPD. */
int dm_fmap_solve(dm_ctx* ctx, int B, int k1, int k2, int D,
                  const float* A, const float* Bm,
                  const double* lam1, const double* lam2, const double* c00,
                  double w_descr, double w_lap, double* C, int32_t* info);

/* The same fit in ONE call: FunctionalMapping.fit with w_descr, w_lap > 0 and every other weight 0
 * (pyFM/functional.py:352-487) = the two projections (dm_project), the pinned column (dm_fmap_c00) and dm_fmap_solve.
 * Same C, bit for bit, as the three calls; the projected descriptors are not returned, which lets the library skip their
 * split-K reduction pass (the Gram kernel adds the chunks up as it reads them).  F1 (B,N1,D), F2 (B,N2,D) fp16. */
int dm_fmap_fit(dm_ctx* ctx, int B, int N1, int N2, int D, int k1, int k2,
                const float* Phi1, int ld1, const float* Phi2, int ld2,
                const float* mass1, const float* mass2, const void* F1, const void* F2,
                const double* lam1, const double* lam2, double w_descr, double w_lap,
                double* C, int32_t* info);
int dm_fmap_fit_f64(dm_ctx* ctx, int B, int N1, int N2, int D, int k1, int k2,
                    const double* Phi1, int ld1, const double* Phi2, int ld2,
                    const double* mass1, const double* mass2, const void* F1, const void* F2,
                    const double* lam1, const double* lam2, double w_descr, double w_lap,
                    double* C, int32_t* info);

/* ---- general functional-map energy and gradient -----------------------------------
 * What scipy's L-BFGS-B evaluates in FunctionalMapping.fit when energy terms beyond w_descr / w_lap are switched on:
 * replaces energy_func_std / grad_energy_std (pyFM/optimize/base_functions.py:480-763) for the terms
 *   weights[0..9] = w_descr, w_lap, w_dcomm, w_p2p, w_stochastic, w_ent, w_range01, w_sumto1, w_area, w_conformal   (HOST array of 10 doubles)
 *   (w_area 1/2 |C^T C - I|^2, w_conformal 1/2 |C^T D2 C - D1|^2 with D = diag(lam / max lam): base_functions.py:228-294.  The
 *    orientation term w_orient, base_functions.py:567-600, is a commutation term like w_dcomm: the host appends its operator pairs,
 *    scaled by sqrt(w_orient / w_dcomm), to ops1 / ops2 -- densematcher_amd/pyFM/functional.py.)
 * i.e. descr_preservation :31, LB_commutation :79, oplist_commutation :168 (descriptor multiplication operators),
 * p2p :296, doubly_stochastic :324, entropy :363, range01 :374, sumto1 :387 (eta == 1, v = None).
 * energy (B) fp64 = the reference's energy value; grad (B,k2,k1) fp64 = its gradient with column 0 zeroed (:759).
 * A (B,k1,D), Bm (B,k2,D) fp32 = the projections (dm_project); ops1 (B,n_ops,k1,k1), ops2 (B,n_ops,k2,k2) fp64 =
 * dm_fmap_descr_ops of the two meshes (needed only when w_dcomm > 0; n_ops = number of descriptors).  The terms in the
 * mapped indicator (last five weights) need Phi1, Phi2, mass1; the indicator itself is never stored (its tiles are formed twice on
 * the float64 matrix cores, statistics then derivative): workspace O(B N k), k1 <= 256. */
int dm_fmap_energy_grad(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, int D,
                        const float* Phi1, int ld1, const float* Phi2, int ld2, const float* mass1,
                        const float* A, const float* Bm, const double* lam1, const double* lam2,
                        const double* ops1 /*nullable*/, const double* ops2 /*nullable*/, int n_ops,
                        const double* weights /*host, 10*/, const double* C, double* energy, double* grad);

/* ---- batched L-BFGS on the device --------------------------------------------------------------------------------
 * The optimiser FunctionalMapping.fit runs for the non-quadratic terms: replaces scipy.optimize.minimize(method = "L-BFGS-B")
 * of pyFM/functional.py:477 (no bounds are set there: plain limited-memory BFGS with L-BFGS-B's line-search constants and
 * stopping tests).  The state of every pair (iterate, gradient, direction, m history pairs, line-search bracket, counters)
 * lives in `state` (dm_lbfgs_state_bytes bytes of device memory owned by the caller); pairs are independent of one another.
 *   dm_lbfgs_init     x0 (B,n) -> state, x_trial = x0
 *   dm_lbfgs_advance  energy (B) and grad (B,n) of x_trial (dm_fmap_energy_grad) -> the pairs' next trial points in x_trial;
 *                     ftol: stop when (f_k - f_k+1) <= ftol max(|f_k|, |f_k+1|, 1) (SciPy: factr * eps = 2.2e-9), pgtol: max |g_i|
 *                     (SciPy 1e-5), maxiter, maxfun (SciPy 15000), maxls (SciPy 20).  Finished pairs keep x_trial at their result.
 *   dm_lbfgs_result   x (B,n), f (B), info (B,4) = status (0 running, 1 gradient, 2 energy decrease, 3 maxiter, 4 maxfun, 5 line
 *                     search failed), iterations, evaluations, history length
 * The host alternates dm_fmap_energy_grad and dm_lbfgs_advance and reads info every few evaluations. */
size_t dm_lbfgs_state_bytes(int B, int n, int m);
int dm_lbfgs_init(dm_ctx* ctx, int B, int n, int m, const double* x0, void* state, double* x_trial);
int dm_lbfgs_advance(dm_ctx* ctx, int B, int n, int m, void* state, const double* energy, const double* grad, double* x_trial,
                     double ftol, double pgtol, int maxiter, int maxfun, int maxls);
int dm_lbfgs_result(dm_ctx* ctx, int B, int n, int m, const void* state, double* x, double* f, int32_t* info);
/* `nsteps` rounds of (dm_fmap_energy_grad at x_trial -> dm_lbfgs_advance) without returning to the caller in between: the loop
 * of one scipy.optimize.minimize call (pyFM/functional.py:477), nsteps evaluations at a time.  `energy` (B) and `grad` (B,k2,k1)
 * are scratch the caller owns; x_trial (B,k2,k1) is the trial point dm_lbfgs_init / the last advance left.  Pairs that have
 * stopped are not moved by further steps.  (A Python loop around the two calls costs more host time per evaluation than the
 * evaluation takes on the device for one small pair.) */
int dm_fmap_fit_steps(dm_ctx* ctx, int nsteps, int B, int N1, int N2, int k1, int k2, int D,
                      const float* Phi1, int ld1, const float* Phi2, int ld2, const float* mass1,
                      const float* A, const float* Bm, const double* lam1, const double* lam2,
                      const double* ops1 /*nullable*/, const double* ops2 /*nullable*/, int n_ops,
                      const double* weights /*host, 10*/, int m, void* state, double* x_trial, double* energy, double* grad,
                      double ftol, double pgtol, int maxiter, int maxfun, int maxls);

/* The WHOLE iterative fit of small maps in one call: FunctionalMapping.fit (pyFM/functional.py:352-487) for maps up to 32 x 32 whose
 * energy has, besides w_descr / w_lap, only element-wise indicator terms and the sum-to-one term (w_p2p, w_ent, w_range01, w_sumto1:
 * base_functions.py:296, 363, 374, 387) -- the notebook's call (example.ipynb cell 11).  One launch per energy evaluation: the
 * workgroup that delivers a pair's last partial sum also adds up, evaluates the O(k^3) terms and advances that pair's L-BFGS
 * (same optimiser, constants and stopping rules as dm_lbfgs_advance); the status words are read every 16 launches.  A pair's
 * result does not depend on the batch it is in (the additions follow a tree fixed by N1, N2 alone).
 *   dm_fmap_fit_fused_ok  1 when the sizes / weights (host array of 10, order of dm_fmap_energy_grad) are taken, else 0 (the caller
 *                         then runs dm_fmap_fit_steps)
 *   x0 (B,k2,k1) start (first column = the pinned one); x_out (B,k2,k1), f_out (B), info_out (B,4) as dm_lbfgs_result;
 *   evaluations_out (host, nullable): launches issued.
 *   maxfun <= 0: ONE evaluation at x0, no optimiser: f_out (B) energy, grad_out (B,k2,k1) gradient (x_out / info_out unused).
 *   dm_fmap_fit_fused takes no operator lists, so it has no commutativity term: weights[2] (w_dcomm) must be 0, DM_EINVAL otherwise
 *   (a fit with operators goes through dm_fmap_fit_steps; dm_fmap_fit_fused_ok(..., n_ops > 0) says so). */
int dm_fmap_fit_fused_ok(int k1, int k2, const double* weights /*host, 10*/, int n_ops);
int dm_fmap_fit_fused(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, int D,
                      const float* Phi1, int ld1, const float* Phi2, int ld2, const float* mass1,
                      const float* A, const float* Bm, const double* lam1, const double* lam2,
                      const double* weights /*host, 10*/, int m, const double* x0,
                      double ftol, double pgtol, int maxiter, int maxfun, int maxls,
                      double* x_out, double* f_out, int32_t* info_out, double* grad_out /*nullable*/, int* evaluations_out /*host, nullable*/);

/* ops[b][d] = Phi[b][:, :k]^T diag(mass[b] * F[b][:, d]) Phi[b][:, :k]   (B, D, k, k) fp64: the multiplication operator
 * of descriptor d in the reduced basis.  Replaces commute_left / commute_right of base_functions.py:550-555
 * (pinv @ (descr[:, i, None] * evects), pinv = evects^T A, pyFM/functional.py:416-417).  B * D <= 65535 per call. */
int dm_fmap_descr_ops(dm_ctx* ctx, int B, int N, int D, int k, const float* Phi, int ld, const float* mass,
                      const void* F, int f_dtype, double* ops);

/* ---- functional map -> vertex maps ---------------------------------------
 * With G = Phi2[:, :k2] C Phi1[:, :k1]^T (never materialised):
 *   knn21[i] = argmin_j |C Phi1_j|^2 - 2 G_ij      (pyFM/spectral/convert.py:138-140)
 *   knn12[j] = argmin_i |Phi2_i C|^2 - 2 G_ij      (pyFM/spectral/convert.py:134-136)
 *   ind21[i] = argmax_j G_ij mass1_j               (convert.py:144 + functional_map.py:49)
 *   ind12[j] = argmax_i G_ij mass1_j               (convert.py:144 + functional_map.py:50)
 * The returned indices are those of the float64 arithmetic above, lowest index on ties.  When all four maps are
 * asked for, N1, N2 >= 256 (any values: operands are padded to whole tiles internally) and k2 >= 65, they come from one pass
 * on the fp16 matrix cores over split operands with a rigorous error bound, every row inside the bound re-evaluated in
 * float64; otherwise from a fused kernel on the f64 matrix cores ("p2p_split" option).
 * Any of the four outputs may be NULL.  knn21/ind21 (B,N2); knn12/ind12 (B,N1). */
int dm_fm_to_p2p(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2,
                 const float* Phi1, int ld1, const float* Phi2, int ld2,
                 const float* mass1, const double* C,
                 int32_t* knn21, int32_t* knn12, int32_t* ind21, int32_t* ind12);
/* float64 eigenvectors and masses: the reference's own inputs (convert.py:134-144 runs on float64 evects and a float64
 * sparse A1).  The fp16 first pass splits the float64 values directly; its error bound and the exact float64
 * re-evaluation are the same, the masses enter the re-evaluation unrounded. */
int dm_fm_to_p2p_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2,
                     const double* Phi1, int ld1, const double* Phi2, int ld2,
                     const double* mass1, const double* C,
                     int32_t* knn21, int32_t* knn12, int32_t* ind21, int32_t* ind12);

/* The "p2p_split" mode (1, 2, 3) dm_fm_to_p2p would take for these sizes with the context's options when all four maps
 * are requested, else 0 (the float64 G kernel).  Informational (bench.py reports the dominant kernel of the step); no
 * reference counterpart. */
int dm_fm_to_p2p_uses_split(const dm_ctx* ctx, int N2, int N1, int k);

/* ---- exact nearest neighbour, k = 1 -----------------------------------------
 * out[b,i] = argmin_j |X[b,j,:] - Y[b,i,:]|^2 (lowest j on ties); X (B,nx,p), Y (B,ny,p) fp64.
 * Replaces pyFM/spectral/nn_utils.py:4-38 (knn_query: sklearn kd-tree, k = 1).
 * Exact: a first pass on the fp16 matrix cores with a rigorous error bound, every row inside the bound
 * re-evaluated in float64 from the original operands (also the search inside dm_zoomout / dm_icp;
 * dm_set_option("knn_split", 0) selects the float64 matrix-core kernel for all of it instead). */
int dm_knn_query_f64(dm_ctx* ctx, int B, int nx, int ny, int p,
                     const double* X, const double* Y, int32_t* out /* B*ny */);

/* ---- k nearest neighbours, k > 1 -------------------------------------------------------
 * idx[b,i,r] = index of the r-th nearest row of X[b] to Y[b,i] (ascending distance, lowest index on equal distances),
 * dist (nullable) = the Euclidean distances; idx / dist (B,ny,k).  Replaces pyFM/spectral/nn_utils.py:4-38 for k > 1
 * (sklearn kneighbors).  The k = 1 searches of the matching path use dm_knn_query_f64. */
int dm_knn_query_topk_f64(dm_ctx* ctx, int B, int nx, int ny, int p, int k,
                          const double* X, const double* Y, int32_t* idx, double* dist /*nullable*/);

/* ---- dense mapped indicator --------------------------------------------------
 * M[b] = ((Phi2[:, :k2] C) Phi1[:, :k1]^T) * mass1[None, :]   (B,N2,N1) fp64.
 * Replaces the matrix returned by FM_to_p2p (pyFM/spectral/convert.py:144) for
 * callers that want it; the arg-max maps come from dm_fm_to_p2p without it. */
int dm_mapped_indicator(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2,
                        const float* Phi1, int ld1, const float* Phi2, int ld2,
                        const float* mass1, const double* C, double* M);
int dm_mapped_indicator_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2,
                            const double* Phi1, int ld1, const double* Phi2, int ld2,
                            const double* mass1, const double* C, double* M);

/* ---- vertex map -> functional map -----------------------------------------
 * C[b] = Phi2[b][:, :k2]^T (mass2[b] * Phi1[b][p21[b], :k1])   (k2 x k1) fp64.
 * Replaces pyFM/spectral/convert.py:14-51 (p2p_to_FM, A2 given). */
int dm_p2p_to_fm(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2,
                 const int32_t* p21, const float* Phi1, int ld1, const float* Phi2, int ld2,
                 const float* mass2, double* C);
int dm_p2p_to_fm_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2,
                     const int32_t* p21, const double* Phi1, int ld1, const double* Phi2, int ld2,
                     const double* mass2, double* C);

/* ---- Laplace-Beltrami eigenbasis -----------------------------------------------------
 * The k smallest eigenpairs of  W phi = lambda A phi  (Phi^T A Phi = I) for B meshes of N vertices each.
 * Replaces the eigensolve of TriMesh.process / laplacian_spectrum (pyFM/mesh/trimesh.py:440-531 -> pyFM/mesh/laplacian.py:
 * 143-182: scipy.sparse.linalg.eigsh(W, k, M=A, sigma=-0.01), ARPACK) by a Chebyshev-filtered subspace iteration.
 * Input: L = A^-1/2 W A^-1/2 in ELL format -- ell_cols (B,N,nnz) int32, ell_vals (B,N,nnz) fp64, padded with (col = row,
 * val = 0) -- and mass (B,N) fp32 = diag(A).  X (B,N,k+guard) fp64: in, a random block (warm_start = 0) or the block a
 * previous call returned (warm_start = 1); out, the orthonormal Ritz vectors of L.  n_iter filtered iterations of
 * polynomial degree `degree` (<= 64; a cold start ramps 4, 8, 16, ... up to it).
 * Output: lam (B,k) fp64 ascending, Phi (B,N,k) fp64 (sign: largest entry of a column positive), resid (B) fp64 =
 * max_j |L x_j - lam_j x_j| over the k wanted pairs -- the caller iterates (warm_start = 1) until it is small enough.
 * ARPACK's output is not reproducible either (random start vector, arbitrary basis of multiple eigenvalues): parity is
 * stated on eigenvalues and invariant subspaces, never on bits.  k + guard <= min(N, 512).
 * warm_start = 2: the dense route for small meshes (where ARPACK, laplacian.py:165, works for any k < N): X = the identity (k + guard =
 * N <= 512), no filter, one Rayleigh-Ritz step = the Jacobi eigendecomposition of L; n_iter / degree are ignored. */
int dm_eigenbasis(dm_ctx* ctx, int B, int N, int nnz, const int32_t* ell_cols, const double* ell_vals, const float* mass,
                  int k, int guard, int n_iter, int degree, int warm_start,
                  double* X, double* lam, double* Phi, double* resid);

/* The linear assignments of mapped indicators WITHOUT the dense matrices (functional_map.py:57, 78 call
 * scipy.optimize.linear_sum_assignment(mapped_indicator, maximize=True) on the N2 x N1 matrix of pyFM/spectral/convert.py:144).  For maps
 * up to 32 x 32 the kernel evaluates a cost row from the factors E2 = Phi2 C and Phi1, a1 -- the same arithmetic, bit for bit, as
 * dm_mapped_indicator's entries (csrc/dm_indicator_dev.h), so the result is the assignment of that matrix -- and n_dense dense N2 x N1
 * matrices (the precise map, functional_map.py:66) ride in the same launch.  Phi1 (n_ind,N1,ld1), Phi2 (n_ind,N2,ld2), mass1 (n_ind,N1),
 * C (n_ind,k2,k1) fp64; col_of_row (n_ind + n_dense, N2), info (n_ind + n_dense): the indicators first.  dm_lsa_indicator_ok: 1 when
 * the sizes are taken (k1, k2 <= 32, N2 <= N1 <= 2048). */
int dm_lsa_indicator_ok(dm_ctx* ctx, int N1, int N2, int k1, int k2);
int dm_lsa_indicator(dm_ctx* ctx, int n_ind, int N1, int N2, int k1, int k2, const float* Phi1, int ld1, const float* Phi2, int ld2,
                     const float* mass1, const double* C, int n_dense, const double* dense /*nullable*/, int maximize,
                     int32_t* col_of_row, int32_t* info);
int dm_lsa_indicator_f64(dm_ctx* ctx, int n_ind, int N1, int N2, int k1, int k2, const double* Phi1, int ld1, const double* Phi2, int ld2,
                         const double* mass1, const double* C, int n_dense, const double* dense /*nullable*/, int maximize,
                         int32_t* col_of_row, int32_t* info);

/* ---- Laplace-Beltrami operators ---------------------------------------------------------------------------------------
 * What TriMesh.process assembles before its eigensolve (pyFM/mesh/trimesh.py:440-482).
 *   dm_tufted_cover   HOST function (no device, no context, thread safe): the tufted intrinsic-Delaunay cover of one mesh -- the
 *                     construction behind robust_laplacian.mesh_laplacian(V, F, mollify_factor) (external C++ wheel of the reference,
 *                     trimesh.py:465-470): mollified edge lengths, front / back copy of every face glued around every edge,
 *                     intrinsic edge flips until every cover edge is Delaunay.  T (2 nf, 3) int32 corner vertices and L (2 nf, 3)
 *                     fp64 side lengths (side s runs from corner s to corner s + 1) of the cover's triangles; info = [flips,
 *                     converged]; mollify_eps (nullable) = what was added to every length.  Its Laplacian = dm_laplacian_* of
 *                     (T, L) with scale 1/2.
 *   dm_laplacian_rows cotangent stiffness rows and lumped masses of B meshes of N vertices and nt triangles each, on the device, in a
 *                     fixed order of additions.  tri (B,nt,3) int32; len (B,nt,3) fp64 intrinsic side lengths (Heron) or null: from
 *                     verts (B,N,3) fp64 with the reference's arithmetic (laplacian.py:88-140, 5-40).  n_verts (B, nullable): meshes
 *                     with fewer vertices / triangles ride along padded (triangles (-1,-1,-1); vertices >= n_verts[b] become decoupled
 *                     rows at the top of the spectrum).  rows: dm_laplacian_rows_bytes of device memory; max_row (host): longest row.
 *   dm_laplacian_ell  the operands of dm_eigenbasis from those rows: L = A^-1/2 W A^-1/2 in ELL (ell_cols, ell_vals (B,N,nnz), nnz >=
 *                     max_row, padded with (col = row, val = 0)) and mass32 (B,N) = fp32(diag A); optional w_vals (B,N,nnz) = the
 *                     entries of W itself and mass64 (B,N) fp64 (what TriMesh.W / TriMesh.A hold). */
int dm_tufted_cover(int n, int nf, const double* verts /*host*/, const int32_t* faces /*host*/, double mollify_factor,
                    int32_t* T /*host, 2 nf x 3*/, double* L /*host, 2 nf x 3*/, int32_t* info /*host, 2, nullable*/, double* mollify_eps /*host, nullable*/);
/* dm_tufted_cover for `count` meshes on n_threads host threads (<= 0: one per hardware thread); arrays of per-mesh sizes and pointers */
int dm_tufted_cover_batch(int count, const int32_t* n, const int32_t* nf, const double* const* verts, const int32_t* const* faces,
                          double mollify_factor, int32_t* const* T, double* const* L, int32_t* info /*count x 2*/,
                          double* mollify_eps /*count, nullable*/, int n_threads);
size_t dm_laplacian_rows_bytes(int B, int N, int nt);
int dm_laplacian_rows(dm_ctx* ctx, int B, int N, int nt, const int32_t* tri, const double* len /*nullable*/, const double* verts /*nullable*/,
                      double scale, const int32_t* n_verts /*nullable*/, void* rows, int* max_row /*host*/);
int dm_laplacian_ell(dm_ctx* ctx, int B, int N, int nt, const void* rows, int nnz, const int32_t* n_verts /*nullable*/,
                     int32_t* ell_cols, double* ell_vals, float* mass32, double* w_vals /*nullable*/, double* mass64 /*nullable*/);

/* ---- precise (barycentric) map ----------------------------------------------------
 * For every vertex i of mesh 2 the face of mesh 1 its spectral embedding projects onto and the barycentric coordinates
 * of the projection: face_match (B,N2) int32, bary (B,N2,3) fp64; dense (B,N2,N1) fp64 optional = the same map as a
 * matrix with three weights per row (what get_precise_map().toarray() returns).  faces1 (B,nf,3) int32.
 * Replaces FunctionalMapping.get_precise_map (pyFM/functional.py:221-251 -> pyFM/spectral/convert.py:185-229 with
 * use_adj = True -> pyFM/spectral/projection_utils.py:16-115).  info (B): 1 if some point of the pair had more than 4096
 * candidate faces (such a point re-tests every face instead of walking a list: slower, same result).  Face indices must lie in
 * [0, N1) (the caller checks; the kernel indexes LDS with them).  N1 + k1 <= about 17000. */
int dm_precise_map(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, int nf,
                   const float* Phi1, int ld1, const float* Phi2, int ld2, const double* C,
                   const int32_t* faces1, int32_t* face_match, double* bary, double* dense /*nullable*/, int32_t* info);
int dm_precise_map_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, int nf,
                       const double* Phi1, int ld1, const double* Phi2, int ld2, const double* C,
                       const int32_t* faces1, int32_t* face_match, double* bary, double* dense /*nullable*/, int32_t* info);

/* ---- linear assignment ------------------------------------------------------------
 * col_of_row[b][r] = column assigned to row r of cost[b] (nr x nc fp64, device), -1 for an unassigned row (nr > nc).
 * Replaces scipy.optimize.linear_sum_assignment(mapped_indicator, maximize=True) of densematcher/functional_map.py:57,66,78
 * (the `hungarian`, `hungarian_precise`, `hungarian_icp` outputs).  Same algorithm as SciPy's (Crouse 2016, shortest
 * augmenting paths), same floating-point steps and tie rules: the assignment equals SciPy's, not merely its objective.
 * One workgroup per matrix; matrices of a batch run concurrently.  info (B): 0 ok; 1 the matrix is infeasible (no complete
 * assignment of finite cost: SciPy raises ValueError("cost matrix is infeasible")), 2 it holds NaN or an infinity of the
 * rejected sign (SciPy: "matrix contains invalid numeric entries"); col_of_row of such a matrix is not meaningful. */
int dm_linear_sum_assignment(dm_ctx* ctx, int B, int nr, int nc, const double* cost, int maximize,
                             int32_t* col_of_row /* B*nr */, int32_t* info /* B */);

/* ---- vertex map -> functional map, least squares -------------------------------
 * C[b] = argmin_X |Phi2[b][:, :k2] X - Phi1[b][p21[b], :k1]|_F   (k2 x k1) fp64, no mass matrix.
 * Replaces pyFM/spectral/convert.py:51 (p2p_to_FM with A2 = None: scipy.linalg.lstsq), the form ICP and ZoomOut on
 * subsampled vertices use.  Normal equations (Phi2^T Phi2) C = Phi2^T Phi1[p21] with one step of iterative refinement; info (B):
 * 0 ok, else the Gram matrix was not positive definite / its inverse did not converge.  Needs N2 >= k2 and Phi2 of full column
 * rank, cond(Phi2) up to ~1e4 (the reference's SVD-based lstsq also returns the minimum-norm solution of rank-deficient
 * problems: those are reported here, not solved).  k2 <= 256.  Phi1 rows may be any (N1 x ld1)
 * matrix (a pulled-back basis P Phi1 with p21 = identity gives the sparse-map form of convert.py:39). */
int dm_p2p_to_fm_lstsq(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2,
                       const int32_t* p21, const float* Phi1, int ld1, const float* Phi2, int ld2,
                       double* C, int32_t* info);
int dm_p2p_to_fm_lstsq_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2,
                           const int32_t* p21, const double* Phi1, int ld1, const double* Phi2, int ld2,
                           double* C, int32_t* info);

/* ---- ZoomOut ----------------------------------------------------------------
 * nit times: p21 = knn21(C_k); C_{k+step} = p2p_to_fm(p21) with k+step columns.
 * Replaces pyFM/refine/zoomout.py:7-44,47-115 (with upstream FM_to_p2p
 * semantics).  C0 (B,k0,k0); Cout (B,kf,kf), kf = k0+nit*step;
 * p21_out (B,N2) optional = knn21(Cout).  Needs ld1, ld2 >= kf. */
int dm_zoomout(dm_ctx* ctx, int B, int N1, int N2, int k0, int nit, int step,
               const float* Phi1, int ld1, const float* Phi2, int ld2,
               const float* mass2, const double* C0, double* Cout, int32_t* p21_out /*nullable*/);
int dm_zoomout_f64(dm_ctx* ctx, int B, int N1, int N2, int k0, int nit, int step,
                   const double* Phi1, int ld1, const double* Phi2, int ld2,
                   const double* mass2, const double* C0, double* Cout, int32_t* p21_out /*nullable*/);

/* ---- spectral ICP -------------------------------------------------------------
 * nit times: p21 = knn21(C); Chat = argmin |Phi2[:, :k2] X - Phi1[p21, :k1]|_F (no mass);
 * C = U eye(k2,k1) V^T with U S V^T = svd(Chat), i.e. the orthogonal polar factor of Chat.
 * Replaces pyFM/refine/icp.py:10-40,43-107 (fixed nit; functional.py:564 uses nit = 10).
 * C0, Cout (B,k2,k1) fp64; resid (B) fp64 optional = max |Cout^T Cout - I| (convergence of the
 * polar iteration); info (B): 0 ok, c+1 = normal equations not SPD.  k1 <= k2 <= 256 (k2 <= 176: in-LDS Cholesky of
 * Phi2^T Phi2; above: Newton-Schulz inverse on the float64 matrix cores). */
int dm_icp(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2,
           const float* Phi1, int ld1, const float* Phi2, int ld2,
           const double* C0, int nit, double* Cout, double* resid /*nullable*/, int32_t* info);
int dm_icp_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2,
               const double* Phi1, int ld1, const double* Phi2, int ld2,
               const double* C0, int nit, double* Cout, double* resid /*nullable*/, int32_t* info);

#ifdef __cplusplus
}
#endif
#endif /* DENSEMATCH_H */

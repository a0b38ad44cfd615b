// Device-resident batched L-BFGS: the optimiser of FunctionalMapping.fit for the non-quadratic energy terms.
//
// The reference drives torch-autograd evaluations from scipy.optimize.minimize(method = "L-BFGS-B") on the host, one pair at
// a time, with a host <-> device round trip per evaluation (pyFM/functional.py:477; base_functions.py:516, 639, 763).  Here
// the optimiser state of EVERY pair of a batch lives on the device: one workgroup per pair advances that pair's own
// limited-memory BFGS iteration (two-loop recursion, strong-Wolfe line search as a state machine) each time a new
// (energy, gradient) pair arrives, and writes the pair's next trial point.  The host only alternates two calls,
//
//        dm_fmap_energy_grad(x_trial) -> dm_lbfgs_advance(state, energy, grad) -> x_trial ...
//
// and looks at the status words every few evaluations.  Pairs are independent: every pair has its own history, step
// length, iteration count and stopping test, so its result does not depend on the batch it is in.
//
// Algorithm = the unconstrained case of L-BFGS-B (Byrd, Lu, Nocedal, Zhu 1995; no bounds are set by the reference):
//   direction   d = -H g, H from the last m pairs (s, y), initial scaling s^T y / y^T y;
//   line search strong Wolfe conditions with SciPy's constants (sufficient decrease 1e-3, curvature 0.9), bracketing then
//               zoom with safeguarded cubic interpolation, at most maxls trials; first step min(1, 1 / |d|) as in lnsrlb;
//   stopping    max |g_i| <= pgtol, or (f_k - f_{k+1}) <= ftol max(|f_k|, |f_{k+1}|, 1), or maxiter / maxfun.
// The first column of C is pinned by a zero gradient there (base_functions.py:759): it never moves.
#include "dm_internal.h"
#include "dm_lbfgs_dev.h"

__global__ __launch_bounds__(256) void lbfgs_advance_kernel(int n, lbfgs_opts o, const double* __restrict__ f_in, const double* __restrict__ g_in,
                                                            double* __restrict__ xt, double* __restrict__ x, double* __restrict__ g,
                                                            double* __restrict__ d, double* __restrict__ S, double* __restrict__ Y,
                                                            double* __restrict__ rho, double* __restrict__ sc, int* __restrict__ ic,
                                                            double* __restrict__ alpha_ws) {
    lb_advance_pair<LB_FAST_M>(blockIdx.x, n, o, f_in, g_in, xt, x, g, d, S, Y, rho, sc, ic, alpha_ws);
}

extern "C" size_t dm_lbfgs_state_bytes(int B, int n, int m) { return lb_state_bytes(B, n, m); }

extern "C" int dm_lbfgs_init(dm_ctx* ctx, int B, int n, int m, const double* x0, void* state, double* x_trial) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && n > 0 && m > 0 && m <= 64, "sizes must be positive, m <= 64");
    DM_REQUIRE(ctx, x0 && state && x_trial, "null pointer");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    lb_layout L = lb_carve(state, B, n, m);
    DM_CHECK_HIP(ctx, hipMemsetAsync(state, 0, dm_lbfgs_state_bytes(B, n, m), ctx->stream));
    DM_CHECK_HIP(ctx, hipMemcpyAsync(L.x, x0, (size_t)B * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    DM_CHECK_HIP(ctx, hipMemcpyAsync(x_trial, x0, (size_t)B * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return DM_OK;
}

extern "C" int dm_lbfgs_advance(dm_ctx* ctx, int B, int n, int m, void* state, const double* energy, const double* grad, double* x_trial,
                                double ftol, double pgtol, int maxiter, int maxfun, int maxls) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && n > 0 && m > 0 && m <= 64, "sizes must be positive, m <= 64");
    DM_REQUIRE(ctx, state && energy && grad && x_trial, "null pointer");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    lb_layout L = lb_carve(state, B, n, m);
    lbfgs_opts o{ftol, pgtol, m, maxiter, maxfun, maxls > 0 ? maxls : 20};
    DM_LAUNCH(ctx, "lbfgs_advance", lbfgs_advance_kernel, dim3(B), dim3(256), 0, n, o, energy, grad, x_trial, L.x, L.g, L.d, L.S, L.Y, L.rho,
              L.sc, L.ic, L.al);
    return DM_OK;
}

// x (B, n): the accepted iterates; f (B): their energies; info (B, 4) int32: status (1 gradient, 2 energy decrease, 3 maxiter,
// 4 maxfun, 5 line search failed, 0 still running), iterations, evaluations, history length
__global__ __launch_bounds__(256) void lbfgs_result_kernel(int n, int B, const double* __restrict__ x, const double* __restrict__ sc,
                                                           const int* __restrict__ ic, double* __restrict__ xo, double* __restrict__ fo,
                                                           int32_t* __restrict__ info) {
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < n; e += 256) xo[(long long)b * n + e] = x[(long long)b * n + e];
    if (threadIdx.x == 0) {
        fo[b] = sc[(long long)b * LS_NSCAL + LS_F];
        const int* icb = ic + (long long)b * LI_NINT;
        info[4 * b] = icb[LI_STATUS]; info[4 * b + 1] = icb[LI_ITER]; info[4 * b + 2] = icb[LI_NFEV]; info[4 * b + 3] = icb[LI_NHIST];
    }
}
extern "C" int dm_lbfgs_result(dm_ctx* ctx, int B, int n, int m, const void* state, double* x, double* f, int32_t* info) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && n > 0 && m > 0, "sizes must be positive");
    DM_REQUIRE(ctx, state && x && f && info, "null pointer");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    lb_layout L = lb_carve(const_cast<void*>(state), B, n, m);
    DM_LAUNCH(ctx, "lbfgs_result", lbfgs_result_kernel, dim3(B), dim3(256), 0, n, B, L.x, L.sc, L.ic, x, f, info);
    return DM_OK;
}

// nsteps x (energy + gradient at the trial points -> advance): the evaluation loop of a fit on the library side of the ABI
extern "C" int dm_fmap_fit_steps(dm_ctx* ctx, int nsteps, int B, int N1, int N2, int k1, int k2, int D, const float* Phi1, int ld1,
                                 const float* Phi2, int ld2, const float* mass1, const float* A, const float* Bm, const double* lam1,
                                 const double* lam2, const double* ops1, const double* ops2, int n_ops, const double* weights, int m,
                                 void* state, double* x_trial, double* energy, double* grad, double ftol, double pgtol, int maxiter,
                                 int maxfun, int maxls) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, nsteps > 0 && state && x_trial && energy && grad, "fit_steps: null pointer or no steps");
    for (int s = 0; s < nsteps; ++s) {
        int rc = dm_fmap_energy_grad(ctx, B, N1, N2, k1, k2, D, Phi1, ld1, Phi2, ld2, mass1, A, Bm, lam1, lam2, ops1, ops2, n_ops, weights,
                                     x_trial, energy, grad);
        if (rc) return rc;
        rc = dm_lbfgs_advance(ctx, B, k2 * k1, m, state, energy, grad, x_trial, ftol, pgtol, maxiter, maxfun, maxls);
        if (rc) return rc;
    }
    return DM_OK;
}

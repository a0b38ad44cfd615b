// Device side of the batched L-BFGS (dm_lbfgs.hip): state layout, line-search state machine and the per-pair advance as a
// device function, shared by the stand-alone advance kernel and the fused evaluation kernel (dm_fitfuse.hip).
#pragma once
#include <math.h>

#include "dm_device.h"

enum { LB_RUN = 0, LB_GTOL = 1, LB_FTOL = 2, LB_MAXITER = 3, LB_MAXFUN = 4, LB_LSFAIL = 5 };
enum { PH_FIRST = 0, PH_BRACKET = 1, PH_ZOOM = 2 };

// per-pair scalar state (doubles): see the accessors below
constexpr int LS_F = 0, LS_DPHI0 = 1, LS_T = 2, LS_TPREV = 3, LS_FPREV = 4, LS_DPREV = 5, LS_TLO = 6, LS_FLO = 7, LS_DLO = 8,
              LS_THI = 9, LS_FHI = 10, LS_DHI = 11, LS_GAMMA = 12, LS_TBEST = 13, LS_FBEST = 14, LS_NSCAL = 16;
// per-pair integer state
constexpr int LI_STATUS = 0, LI_PHASE = 1, LI_ITER = 2, LI_NFEV = 3, LI_NHIST = 4, LI_HEAD = 5, LI_LSIT = 6, LI_NINT = 8;

struct lbfgs_opts { double ftol, pgtol; int m, maxiter, maxfun, maxls; };

__device__ __forceinline__ double lb_block_sum(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ __forceinline__ double lb_block_max(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
__device__ __forceinline__ double lb_dot(const double* a, const double* b, int n, double* sh) {
    double s = 0.0;
    for (int e = threadIdx.x; e < n; e += 256) s = fma(a[e], b[e], s);
    return lb_block_sum(s, sh);
}

// One-barrier block sum for the register-resident direction update.  The additions follow lb_block_sum's tree (partners at
// lane distance 32, 16, 8, 4, 2, 1, then the four waves); the partner's value comes over the vector ALU instead of
// ds_bpermute: half swaps for 32 and 16, row rotations for 8, 4, 2 (after the level above, lanes that differ in a higher
// bit hold equal sums, so "i + d mod 16" is as good as "i xor d"), a quad permute for 1.  (The compiler fuses the product
// into the first addition, so the last bits differ from the generic path's: where the energy is flat to 1e-10 the
// number of iterations before the ftol test fires moves with such bits -- 289 to 440 evaluations on the notebook's fit.)
template <int LEVEL>
__device__ __forceinline__ int lb_partner(int x) {
    if constexpr (LEVEL == 5) {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);     // r[0] = low half twice, r[1] = high half twice
        return (int)((threadIdx.x & 32) ? r[0] : r[1]);
    } else if constexpr (LEVEL == 4) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);     // r[0] = rows 0 0 2 2, r[1] = rows 1 1 3 3
        return (int)((threadIdx.x & 16) ? r[0] : r[1]);
    } else if constexpr (LEVEL == 3) return __builtin_amdgcn_mov_dpp(x, 0x128, 0xf, 0xf, false);    // row_ror:8
    else if constexpr (LEVEL == 2) return __builtin_amdgcn_mov_dpp(x, 0x124, 0xf, 0xf, false);      // row_ror:4
    else if constexpr (LEVEL == 1) return __builtin_amdgcn_mov_dpp(x, 0x122, 0xf, 0xf, false);      // row_ror:2
    else return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, false);                                 // quad_perm [1,0,3,2]
}
template <int LEVEL>
__device__ __forceinline__ double lb_add_level(double v) {
    return v + __hiloint2double(lb_partner<LEVEL>(__double2hiint(v)), lb_partner<LEVEL>(__double2loint(v)));
}
__device__ __forceinline__ double lb_fast_sum(double v, double (*sh2)[4], int& par) {
    v = lb_add_level<5>(v); v = lb_add_level<4>(v); v = lb_add_level<3>(v);
    v = lb_add_level<2>(v); v = lb_add_level<1>(v); v = lb_add_level<0>(v);
    if ((threadIdx.x & 63) == 0) sh2[par][threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = (sh2[par][0] + sh2[par][1]) + (sh2[par][2] + sh2[par][3]);
    par ^= 1;
    return r;
}
constexpr int LB_FAST_M = 32;      // history length the register-resident update holds (one coefficient per thread: n <= 256)
constexpr int LB_FAST_M_FUSED = 12;  // the same inside the fused evaluation kernel (dm_fitfuse.hip): its registers count against the
                                     // occupancy of that kernel's element loop, SciPy's default history is 10; longer ones take the generic path

// minimiser of the cubic through (a, fa, da), (b, fb, db), safeguarded into the inner 80 % of the interval; bisection when
// the cubic has no minimiser there
__device__ __forceinline__ double lb_cubic(double a, double fa, double da, double b, double fb, double db) {
    const double lo = fmin(a, b), hi = fmax(a, b), w = hi - lo;
    double t = 0.5 * (a + b);
    const double d1 = da + db - 3.0 * (fa - fb) / (a - b);
    const double rad = d1 * d1 - da * db;
    if (rad >= 0.0 && isfinite(rad)) {
        const double d2 = copysign(sqrt(rad), b - a);
        const double den = db - da + 2.0 * d2;
        if (den != 0.0) {
            const double c = b - (b - a) * (db + d2 - d1) / den;
            if (isfinite(c)) t = c;
        }
    }
    return fmin(fmax(t, lo + 0.1 * w), hi - 0.1 * w);
}

// One workgroup per pair.  In: the energy f_t and gradient g_t at the pair's current trial point xt.  Out: the next xt.
//   x, g, d (B, n): accepted point, its gradient, search direction;  S, Y (B, m, n): history;  rho (B, m)
// (a device function: the stand-alone kernel below and the fused evaluation kernel of dm_fitfuse.hip, whose last workgroup to finish
//  a pair's evaluation advances that pair, run the same code; 256 threads, every thread of the workgroup must call it)
template <int FASTM>
__device__ __forceinline__ void lb_advance_pair(const int b, int n, lbfgs_opts o, const double* __restrict__ f_in, const double* __restrict__ g_in,
                                                double* __restrict__ xt, double* __restrict__ x, double* __restrict__ g,
                                                double* __restrict__ d, double* __restrict__ S, double* __restrict__ Y,
                                                double* __restrict__ rho, double* __restrict__ sc, int* __restrict__ ic,
                                                double* __restrict__ alpha_ws) {
    __shared__ double sh[4];
    __shared__ int s_act;
    __shared__ double s_tnext;
    const int t = threadIdx.x;
    double* scb = sc + (long long)b * LS_NSCAL;
    int* icb = ic + (long long)b * LI_NINT;
    if (icb[LI_STATUS] != LB_RUN) return;                 // finished pairs keep xt = x: their evaluations are repeated, not used
    const double* gt = g_in + (long long)b * n;
    double* xtb = xt + (long long)b * n;
    double* xb = x + (long long)b * n;
    double* gb = g + (long long)b * n;
    double* db = d + (long long)b * n;
    double* Sb = S + (long long)b * o.m * n;
    double* Yb = Y + (long long)b * o.m * n;
    double* rb = rho + (long long)b * o.m;
    double* al = alpha_ws + (long long)b * o.m;
    const double ft = f_in[b];
    const int phase = icb[LI_PHASE];
    const int nfev = icb[LI_NFEV] + 1;
    bool restart = false;                                 // line search failed with a non-empty history: retry along -g
    // ---------------------------------------------------------------- line-search bookkeeping (thread 0 decides, all follow)
    if (phase != PH_FIRST) {                              // (PH_FIRST = the evaluation at x0: nothing to search yet)
        const double dphit = lb_dot(gt, db, n, sh);
        if (t == 0) {
            const double c1 = 1e-3, c2 = 0.9;
            const double f0 = scb[LS_F], dphi0 = scb[LS_DPHI0], tt = scb[LS_T];
            int act = 0;                                  // 0: next trial at s_tnext, 1: accept this point, 2: line search failed
            double tn = tt;
            const int lsit = icb[LI_LSIT] + 1;
            const bool armijo = ft <= f0 + c1 * tt * dphi0 && isfinite(ft);
            if (armijo && ft < scb[LS_FBEST]) { scb[LS_FBEST] = ft; scb[LS_TBEST] = tt; }
            if (phase == PH_BRACKET) {
                if (!armijo || (lsit > 1 && ft >= scb[LS_FPREV])) {
                    scb[LS_TLO] = scb[LS_TPREV]; scb[LS_FLO] = scb[LS_FPREV]; scb[LS_DLO] = scb[LS_DPREV];
                    scb[LS_THI] = tt; scb[LS_FHI] = ft; scb[LS_DHI] = dphit;
                    icb[LI_PHASE] = PH_ZOOM;
                    tn = isfinite(ft) ? lb_cubic(scb[LS_TLO], scb[LS_FLO], scb[LS_DLO], tt, ft, dphit) : 0.5 * (scb[LS_TLO] + tt);
                } else if (fabs(dphit) <= -c2 * dphi0) {
                    act = 1;
                } else if (dphit >= 0.0) {
                    scb[LS_TLO] = tt; scb[LS_FLO] = ft; scb[LS_DLO] = dphit;
                    scb[LS_THI] = scb[LS_TPREV]; scb[LS_FHI] = scb[LS_FPREV]; scb[LS_DHI] = scb[LS_DPREV];
                    icb[LI_PHASE] = PH_ZOOM;
                    tn = lb_cubic(tt, ft, dphit, scb[LS_THI], scb[LS_FHI], scb[LS_DHI]);
                } else {
                    scb[LS_TPREV] = tt; scb[LS_FPREV] = ft; scb[LS_DPREV] = dphit;
                    tn = 2.5 * tt;                        // extrapolate
                }
            } else {                                      // PH_ZOOM
                if (!armijo || ft >= scb[LS_FLO]) {
                    scb[LS_THI] = tt; scb[LS_FHI] = ft; scb[LS_DHI] = dphit;
                } else {
                    if (fabs(dphit) <= -c2 * dphi0) act = 1;
                    if (dphit * (scb[LS_THI] - scb[LS_TLO]) >= 0.0) { scb[LS_THI] = scb[LS_TLO]; scb[LS_FHI] = scb[LS_FLO]; scb[LS_DHI] = scb[LS_DLO]; }
                    scb[LS_TLO] = tt; scb[LS_FLO] = ft; scb[LS_DLO] = dphit;
                }
                if (!act) {
                    tn = isfinite(scb[LS_FHI]) ? lb_cubic(scb[LS_TLO], scb[LS_FLO], scb[LS_DLO], scb[LS_THI], scb[LS_FHI], scb[LS_DHI])
                                               : 0.5 * (scb[LS_TLO] + scb[LS_THI]);
                    if (fabs(scb[LS_THI] - scb[LS_TLO]) <= 1e-14 * fmax(fabs(scb[LS_TLO]), fabs(scb[LS_THI]))) act = armijo ? 1 : 2;
                }
            }
            if (!act && lsit >= o.maxls) {
                // out of trials: take the best point that gave sufficient decrease (it is the current one if we are on it)
                if (scb[LS_TBEST] > 0.0 && scb[LS_TBEST] == tt) act = 1;
                else if (scb[LS_TBEST] > 0.0) { tn = scb[LS_TBEST]; icb[LI_LSIT] = o.maxls - 1; }   // one more evaluation there, then accepted
                else act = 2;
            } else if (!act) {
                icb[LI_LSIT] = lsit;
            }
            if (!act && nfev >= o.maxfun) act = (armijo ? 1 : 2);
            s_act = act;
            s_tnext = tn;
        }
        __syncthreads();
        if (s_act == 0) {
            const double tn = s_tnext;
            for (int e = t; e < n; e += 256) xtb[e] = fma(tn, db[e], xb[e]);
            if (t == 0) { scb[LS_T] = tn; icb[LI_NFEV] = nfev; }
            return;
        }
        if (s_act == 2) {
            // no acceptable step: with a history, drop it and search again along steepest descent from the accepted point (what
            // L-BFGS-B does once, "refresh the lbfgs memory and restart"); without, give up at the accepted point
            if (icb[LI_NHIST] == 0) {
                for (int e = t; e < n; e += 256) xtb[e] = xb[e];
                if (t == 0) { icb[LI_NFEV] = nfev; icb[LI_STATUS] = LB_LSFAIL; }
                return;
            }
            restart = true;
        }
    }
    // ---------------------------------------------------------------- accepted: the trial point becomes the iterate
    const double fold = scb[LS_F];
    const double fcur = restart ? fold : ft;              // energy at the point the new direction starts from
    int nh = icb[LI_NHIST], head = icb[LI_HEAD], iter = icb[LI_ITER];
    __syncthreads();                                      // (every thread has read the scalars before thread 0 rewrites them)
    if (restart) {
        nh = 0;
    } else {
    if (phase != PH_FIRST) {
        // s = xt - x, y = g_t - g
        double sy = 0.0, yy = 0.0;
        double* Sn = Sb + (long long)head * n;
        double* Yn = Yb + (long long)head * n;
        for (int e = t; e < n; e += 256) {
            const double s_ = xtb[e] - xb[e], y_ = gt[e] - gb[e];
            Sn[e] = s_; Yn[e] = y_;
            sy = fma(s_, y_, sy); yy = fma(y_, y_, yy);
        }
        sy = lb_block_sum(sy, sh);
        yy = lb_block_sum(yy, sh);
        if (sy > 2.2e-16 * yy && yy > 0.0) {              // curvature condition (L-BFGS-B: skip the update otherwise)
            if (t == 0) { rb[head] = 1.0 / sy; scb[LS_GAMMA] = sy / yy; }
            head = (head + 1) % o.m;
            nh = min(nh + 1, o.m);
        }
        iter += 1;
    }
    double gmax = 0.0;
    for (int e = t; e < n; e += 256) {
        const double ge = gt[e];
        xb[e] = xtb[e];
        gb[e] = ge;
        gmax = (ge != ge) ? DM_INF_F64 : fmax(gmax, fabs(ge));    // (fmax drops a NaN: a NaN gradient must not read as "converged")
    }
    gmax = lb_block_max(gmax, sh);
    __syncthreads();
    int status = LB_RUN;
    if (!isfinite(ft) || !isfinite(gmax)) status = LB_LSFAIL;       // energy or gradient not finite at an accepted point: abnormal end
    else if (gmax <= o.pgtol) status = LB_GTOL;
    else if (phase != PH_FIRST && (fold - ft) <= o.ftol * fmax(fmax(fabs(fold), fabs(ft)), 1.0)) status = LB_FTOL;
    else if (iter >= o.maxiter) status = LB_MAXITER;
    else if (nfev >= o.maxfun) status = LB_MAXFUN;
    if (status != LB_RUN) {
        if (t == 0) { scb[LS_F] = ft; icb[LI_STATUS] = status; icb[LI_ITER] = iter; icb[LI_NFEV] = nfev; icb[LI_NHIST] = nh; icb[LI_HEAD] = head; }
        return;                                           // xt == x already
    }
    }
    // ---------------------------------------------------------------- new direction: two-loop recursion, d = -H g
    if (n <= 256 && o.m <= FASTM) {
        // small problems (the notebook's 15 x 15 map: n = 225): the 2 m dependent dot products are the whole cost of a call,
        // so the history rows are fetched up front (independent loads, one latency), the direction stays in a register and
        // a dot product is one wave sum on the vector ALU + one barrier  (49 -> 23 us per call)
        __shared__ double sh2[2][4];
        int par = 0;
        const bool on = t < n;
        double sreg[FASTM], yreg[FASTM], rreg[FASTM], areg[FASTM];
        __syncthreads();                                  // (rho / gamma of the newest pair were written by thread 0)
#pragma unroll
        for (int q = 0; q < FASTM; ++q) {
            const int idx = (head - 1 - q + 2 * o.m) % o.m;
            const bool live = q < nh;
            sreg[q] = (live && on) ? Sb[(long long)idx * n + t] : 0.0;
            yreg[q] = (live && on) ? Yb[(long long)idx * n + t] : 0.0;
            rreg[q] = live ? rb[idx] : 0.0;
        }
        double dr = on ? -gb[t] : 0.0;
#pragma unroll
        for (int q = 0; q < FASTM; ++q) {             // newest to oldest
            if (q < nh) {                                 // (uniform)
                const double a = rreg[q] * lb_fast_sum(sreg[q] * dr, sh2, par);
                areg[q] = a;
                dr = fma(-a, yreg[q], dr);
            }
        }
        dr *= nh > 0 ? scb[LS_GAMMA] : 1.0;
#pragma unroll
        for (int q = FASTM - 1; q >= 0; --q) {        // oldest to newest
            if (q < nh) {
                const double be = rreg[q] * lb_fast_sum(yreg[q] * dr, sh2, par);
                dr = fma(areg[q] - be, sreg[q], dr);
            }
        }
        if (on) db[t] = dr;
        __syncthreads();
    } else {
    for (int e = t; e < n; e += 256) db[e] = -gb[e];
    __syncthreads();
    for (int q = 0; q < nh; ++q) {                        // newest to oldest
        const int idx = (head - 1 - q + 2 * o.m) % o.m;
        const double a = rb[idx] * lb_dot(Sb + (long long)idx * n, db, n, sh);
        if (t == 0) al[idx] = a;
        const double* Yi = Yb + (long long)idx * n;
        for (int e = t; e < n; e += 256) db[e] = fma(-a, Yi[e], db[e]);
        __syncthreads();
    }
    const double gamma = nh > 0 ? scb[LS_GAMMA] : 1.0;
    for (int e = t; e < n; e += 256) db[e] *= gamma;
    __syncthreads();
    for (int q = nh - 1; q >= 0; --q) {                   // oldest to newest
        const int idx = (head - 1 - q + 2 * o.m) % o.m;
        const double be = rb[idx] * lb_dot(Yb + (long long)idx * n, db, n, sh);
        const double a = al[idx];
        const double* Si = Sb + (long long)idx * n;
        for (int e = t; e < n; e += 256) db[e] = fma(a - be, Si[e], db[e]);
        __syncthreads();
    }
    }
    double dphi0 = lb_dot(gb, db, n, sh);
    if (!(dphi0 < 0.0)) {                                 // not a descent direction (numerical breakdown): restart from steepest descent
        for (int e = t; e < n; e += 256) db[e] = -gb[e];
        __syncthreads();
        dphi0 = lb_dot(gb, db, n, sh);
        nh = 0;
    }
    // first trial step: 1, or min(1, 1 / |d|) on the very first iteration (lnsrlb)
    double t0 = 1.0;
    if (iter == 0 || restart) {
        const double dn = sqrt(lb_dot(db, db, n, sh));
        t0 = dn > 0.0 ? fmin(1.0, 1.0 / dn) : 1.0;
    }
    for (int e = t; e < n; e += 256) xtb[e] = fma(t0, db[e], xb[e]);
    if (t == 0) {
        scb[LS_F] = fcur; scb[LS_DPHI0] = dphi0; scb[LS_T] = t0;
        scb[LS_TPREV] = 0.0; scb[LS_FPREV] = fcur; scb[LS_DPREV] = dphi0;
        scb[LS_TBEST] = 0.0; scb[LS_FBEST] = fcur;
        icb[LI_PHASE] = PH_BRACKET; icb[LI_ITER] = iter; icb[LI_NFEV] = nfev; icb[LI_NHIST] = nh; icb[LI_HEAD] = head; icb[LI_LSIT] = 0;
    }
}


struct lb_layout { double *x, *g, *d, *S, *Y, *rho, *al, *sc; int* ic; };
static inline size_t lb_state_bytes(int B, int n, int m) {
    return ((size_t)B * n * 3 + (size_t)B * m * n * 2 + (size_t)B * m * 2 + (size_t)B * LS_NSCAL) * 8 + (size_t)B * LI_NINT * 4 + 1024;
}
static inline lb_layout lb_carve(void* state, int B, int n, int m) {
    lb_layout L;
    double* p = (double*)state;
    L.x = p; p += (size_t)B * n;
    L.g = p; p += (size_t)B * n;
    L.d = p; p += (size_t)B * n;
    L.S = p; p += (size_t)B * m * n;
    L.Y = p; p += (size_t)B * m * n;
    L.rho = p; p += (size_t)B * m;
    L.al = p; p += (size_t)B * m;
    L.sc = p; p += (size_t)B * LS_NSCAL;
    L.ic = (int*)p;
    return L;
}

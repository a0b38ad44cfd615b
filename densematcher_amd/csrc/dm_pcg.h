// The k2 linear systems of a pair by a BATCHED preconditioned conjugate-gradient iteration on the float64 matrix cores (r06).
//
//   row i of C:   (P_ff + diag(dd_i)) x_i = rhs_i,     dd_i[c] = w_l ((lam1[c + 1] - lam2[i]) / scale)^2          (fmap_solve_*: same systems)
//
// All k2 systems of a pair share P_ff = w_d (A A^T)[1:, 1:] and differ by a diagonal.  The direct solvers factor every one of them
// (k2 Cholesky factorisations of order k1 - 1: serial pivot chains, 0.20 of the float64 matrix rate at k = 128).  The
// matrices are well conditioned where the path runs them -- cond(P_ff + diag(dd_i)) has median 13.6 on the config-2 fixture, and the
// Jacobi-preconditioned iteration reaches |x - x_direct| <= 1e-9 in 14 - 19 steps on sigma = 0.1 / 1.0 / smooth descriptors
// (profiles/r06_solver_jacobi_pcg_experiment.txt; r04 measured a SHARED Cholesky preconditioner, 23 - 66 steps, and dropped the idea) --
// so the iteration's one product per step, P_ff times the block of direction vectors of 32 systems, is a GEMM the matrix cores like:
//   * a workgroup = one pair x 32 systems; its waves own two 16-row tiles of P_ff each and keep them IN REGISTERS as A operands of
//     v_mfma_f64_16x16x4_f64 for the whole iteration (128 doubles per lane at k = 128; one wave per SIMD);
//   * the direction block lives in the LDS (the B operand: one ds_read_b64 per two matrix instructions), x / r / p in registers in
//     the accumulator layout, so every vector update is in-lane;
//   * the two inner products per step are column sums: in-lane, two lane swaps across the four row groups of a lane's column,
//     four partials per column through the LDS, added in wave order by everybody (every wave holds every column's total: the
//     stopping test is wave-uniform without a flag).
// Arithmetic per step and pair: 2 n^2 k2 flops against (n^3 / 3 + 2 n^2) k2 of the factorisations: 17 steps = 0.27 of the flops at
// n = 127, all of them matrix instructions.
// A workgroup that does not reach the tolerance in `maxit` steps, or meets a non-positive curvature (the matrix is not positive
// definite), raises the pair's flag: the caller then runs the direct solver on the flagged pairs (it also reports singular systems).
#pragma once
#include "dm_device.h"

constexpr int PCG_NS = 32;        // systems per workgroup: two 16-column tiles

// did the iteration flag pair b (any of its `ng` workgroups)?  -- the direct solvers' fall-back launch
__device__ __forceinline__ bool pcg_flagged(const int32_t* __restrict__ flags, int ng, int b) {
    int f = 0;
    for (int q = 0; q < ng; ++q) f |= flags[(long long)b * ng + q];
    return f != 0;
}
constexpr int PCG_LDP = 48;       // LDS row stride (doubles) of the direction block: the four k-rows of a ds_read_b64 hit disjoint bank halves

static inline size_t pcg_lds_bytes(int NT, int NW) { return ((size_t)NT * 16 * PCG_LDP + 2 * NW * PCG_NS + 64) * sizeof(double); }

// sum over the rows of a column (= over a system's unknowns) of the per-lane partials v[ct]: across the lane's four row groups by
// shuffles, across the waves through `red` (NW x 32 doubles), every lane ends with the totals of its two columns, added in wave order
// sum over the four 16-lane rows of a wave, the same total (and the same order of additions) in every row: (row 0 + row 1) + (row 2 + row 3).
// Two swaps on the vector ALU per step and half -- v_permlane16_swap: odd rows of the first operand <-> even rows of the second;
// v_permlane32_swap: the halves -- instead of two dependent ds_bpermute round trips (__shfl_xor).
__device__ __forceinline__ double pcg_rows4_sum(double v) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double e = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    lo = (unsigned)__double2loint(e); hi = (unsigned)__double2hiint(e);
    const auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)d[0], (int)c[0]) + __hiloint2double((int)d[1], (int)c[1]);
}

template <int NW>
__device__ __forceinline__ void pcg_colsum(double (&v)[2], double* red, int wave, int lane) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) v[ct] = pcg_rows4_sum(v[ct]);
    if (lane < 16) { red[wave * PCG_NS + lane] = v[0]; red[wave * PCG_NS + 16 + lane] = v[1]; }
    __syncthreads();
    const int l15 = lane & 15;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        double s = red[ct * 16 + l15];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += red[w * PCG_NS + ct * 16 + l15];
        v[ct] = s;
    }
}

// NT: 16-row tiles of the matrix (n <= 16 NT); NW waves, each owning NT / NW tiles
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW, 1) void fmap_solve_pcg_kernel(const double* __restrict__ PQ, const double* __restrict__ lam1,
                                                                    const double* __restrict__ lam2, const double* __restrict__ c00,
                                                                    double w_lap, int k1, int k2, int ngroups, double tol2, int maxit,
                                                                    int slow_it, double slow_tol2,
                                                                    double* __restrict__ C, int32_t* __restrict__ fallback, int32_t* __restrict__ info) {
    static_assert(NT % NW == 0, "whole tiles per wave");
    constexpr int RTW = NT / NW, KS = NT * 4;
    extern __shared__ __attribute__((aligned(16))) double pcg_sm[];
    double* pL = pcg_sm;                                  // [16 NT][PCG_LDP] direction block, (unknown, system)
    double* red = pL + NT * 16 * PCG_LDP;                 // [2][NW][32] column partials (two buffers: consecutive sums need no barrier between them)
    int* s_notpd = reinterpret_cast<int*>(red + 2 * NW * PCG_NS);     // a non-positive diagonal met while the systems were set up
    const int b = (int)blockIdx.x / ngroups, g = (int)blockIdx.x - b * ngroups;
    const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = k1 - 1;
    const double* P = PQ + (long long)b * (k1 + k2) * k1;
    const double* Q = P + (long long)k1 * k1;
    const double* l1 = lam1 + (long long)b * k1;
    const double* l2 = lam2 + (long long)b * k2;

    // scale = max(lam1.max(), lam2.max())   (functional.py:404)
    double mx = -DM_INF_F64;
    for (int q = t; q < k1; q += 64 * NW) mx = fmax(mx, l1[q]);
    for (int q = t; q < k2; q += 64 * NW) mx = fmax(mx, l2[q]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if (lane == 0) red[wave] = mx;
    if (t == 0) *s_notpd = 0;
    __syncthreads();
    double scale = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) scale = fmax(scale, red[w]);
    __syncthreads();

    // this wave's tiles of P_ff as A operands: lane l holds P_ff[16 R + (l & 15)][4 s + (l >> 4)]
    double Pf[RTW][KS];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) {
        const int row = 16 * (wave * RTW + rt) + l15;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int col = 4 * s + lg;
            Pf[rt][s] = (row < n && col < n) ? P[(long long)(row + 1) * k1 + col + 1] : 0.0;
        }
    }
    // vectors in the accumulator layout: element (rt, ct, q) = unknown c = 16 (wave RTW + rt) + lg + 4 q of system i = 32 g + 16 ct + l15
    double x[RTW][2][4], r[RTW][2][4], pv[RTW][2][4], minv[RTW][2][4], dd[RTW][2][4];
    double part[2] = {0.0, 0.0};
    bool notpd = false;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int i = g * PCG_NS + ct * 16 + l15;
        const bool iok = i < k2;
        const double l2i = iok ? l2[i] / scale : 0.0;
        const double ci0 = (i == 0) ? c00[b] : 0.0;                  // get_x0: column 0 is (c00, 0, ..., 0)^T
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 16 * (wave * RTW + rt) + lg + 4 * q;
                const bool ok = iok && c < n;
                const int cc = ok ? c : 0, ii = ok ? i : 0;
                const double d = l1[cc + 1] / scale - l2i;
                const double dv = w_lap * (d * d);
                const double m = P[(long long)(cc + 1) * (k1 + 1)] + dv;
                const double bv = Q[(long long)ii * k1 + cc + 1] - P[(long long)(cc + 1) * k1] * ci0;
                if (ok && !(m > 0.0)) notpd = true;
                dd[rt][ct][q] = ok ? dv : 0.0;
                minv[rt][ct][q] = (ok && m > 0.0) ? 1.0 / m : 0.0;
                x[rt][ct][q] = 0.0;
                r[rt][ct][q] = ok ? bv : 0.0;
                pv[rt][ct][q] = r[rt][ct][q] * minv[rt][ct][q];
                part[ct] = fma(r[rt][ct][q], pv[rt][ct][q], part[ct]);
            }
    }
    if (notpd) *s_notpd = 1;                              // (seen by everybody behind the barrier of the sum below: a workgroup-uniform exit)
    pcg_colsum<NW>(part, red, wave, lane);
    notpd = *s_notpd != 0;
    double rz[2] = {part[0], part[1]}, rz0[2] = {part[0], part[1]};
    int rbuf = 1;
    bool done = false;
    int it = 0;
    for (; it < maxit && !done && !notpd; ++it) {
        // direction block -> LDS (rows past n are zero: their minv is)
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) pL[(16 * (wave * RTW + rt) + lg + 4 * q) * PCG_LDP + ct * 16 + l15] = pv[rt][ct][q];
        __syncthreads();
        // Ap = P_ff p on the matrix cores
        f64x4 acc[RTW][2];
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f64x4{0.0, 0.0, 0.0, 0.0};
        // the B operands of a group of four k-steps are requested a whole group (16 matrix instructions) ahead of their use: with one wave
        // per SIMD nothing else covers an LDS round trip (the compiler's own schedule asked for a k-step's pair two instructions ahead and
        // waited lgkmcnt(0) in front of every fourth one)
        constexpr int G = 4, NG = KS / G;
        static_assert(KS % G == 0, "whole groups of k-steps");
        double bq[2][G][2];
#pragma unroll
        for (int u = 0; u < G; ++u) { bq[0][u][0] = pL[(4 * u + lg) * PCG_LDP + l15]; bq[0][u][1] = pL[(4 * u + lg) * PCG_LDP + 16 + l15]; }
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if (gq + 1 < NG) {
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int sn = (gq + 1) * G + u;
                    bq[(gq + 1) & 1][u][0] = pL[(4 * sn + lg) * PCG_LDP + l15];
                    bq[(gq + 1) & 1][u][1] = pL[(4 * sn + lg) * PCG_LDP + 16 + l15];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int s = gq * G + u;
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt) {
                    acc[rt][0] = mfma_f64_16x16x4(Pf[rt][s], bq[gq & 1][u][0], acc[rt][0]);
                    acc[rt][1] = mfma_f64_16x16x4(Pf[rt][s], bq[gq & 1][u][1], acc[rt][1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // + diag(dd) p;  p . Ap
        part[0] = part[1] = 0.0;
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[rt][ct][q] = fma(dd[rt][ct][q], pv[rt][ct][q], acc[rt][ct][q]);
                    part[ct] = fma(pv[rt][ct][q], acc[rt][ct][q], part[ct]);
                }
        pcg_colsum<NW>(part, red + rbuf * NW * PCG_NS, wave, lane);
        rbuf ^= 1;
        double alpha[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            if (rz[ct] > 0.0 && !(part[ct] > 0.0)) notpd = true;       // non-positive curvature along a non-zero direction
            alpha[ct] = (part[ct] > 0.0) ? rz[ct] / part[ct] : 0.0;
        }
        part[0] = part[1] = 0.0;
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x[rt][ct][q] = fma(alpha[ct], pv[rt][ct][q], x[rt][ct][q]);
                    r[rt][ct][q] = fma(-alpha[ct], acc[rt][ct][q], r[rt][ct][q]);
                    const double z = r[rt][ct][q] * minv[rt][ct][q];
                    part[ct] = fma(r[rt][ct][q], z, part[ct]);
                    acc[rt][ct][q] = z;
                }
        pcg_colsum<NW>(part, red + rbuf * NW * PCG_NS, wave, lane);
        rbuf ^= 1;
        bool conv = true;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const double beta = rz[ct] > 0.0 ? part[ct] / rz[ct] : 0.0;
            rz[ct] = part[ct];
            conv = conv && !(rz[ct] > tol2 * rz0[ct]);
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q) pv[rt][ct][q] = fma(beta, pv[rt][ct][q], acc[rt][ct][q]);
        }
        // every wave holds every column's totals (16 columns per tile over the lanes' l15, both tiles per lane): the same answer in all waves
        done = __all(conv) != 0;
        notpd = __any(notpd) != 0;
        if (notpd) break;
        // a pair that is far from done after `slow_it` steps (reduction above `slow_tol2`) is ill conditioned: it goes to the direct solver
        // now rather than after maxit steps (rank-deficient descriptors: 344 us of iteration in front of the 376 us direct solve)
        if (it + 1 == slow_it) {
            bool slow = false;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) slow = slow || (rz[ct] > slow_tol2 * rz0[ct]);
            if (__any(slow)) break;
        }
    }
    // the last direction block may still be read by a slower wave: nothing below writes the LDS
    // one flag per (pair, group), written by every workgroup: nothing to clear before the launch; group 0 also clears the pair's status word
    // for the direct solver behind (which reports singular systems there)
    if (t == 0) { fallback[(long long)b * ngroups + g] = (!done || notpd) ? 1 : 0; if (g == 0) info[b] = 0; }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int i = g * PCG_NS + ct * 16 + l15;
        if (i < k2) {
            double* Crow = C + ((long long)b * k2 + i) * k1;
            if (wave == 0 && lg == 0) Crow[0] = (i == 0) ? c00[b] : 0.0;
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 16 * (wave * RTW + rt) + lg + 4 * q;
                    if (c < n) Crow[c + 1] = x[rt][ct][q];
                }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Systems of order 129 .. 256: P_ff no longer fits the registers of a workgroup (317 KB at n = 199), so the A operands are STREAMED every
// step from a packed image -- fragment (row tile R, k-step s) = 64 consecutive doubles, lane l's value P_ff[16 R + (l & 15)][4 s + (l >> 4)],
// 512-byte coalesced loads, a pair's image (333 KB at n = 199) shared by its workgroups in one XCD's L2 (xcd_remap) -- a few k-steps
// ahead of the matrix instructions that consume them.  Eight waves (two per SIMD), two row tiles each; everything else as above.
constexpr int PCGS_NT = 16;       // row tiles a workgroup covers (n <= 256), PCGS_NW waves x 2
constexpr int PCGS_NW = 8;
constexpr int PCGS_U = 4;         // k-steps per prefetch group (the image's k-steps are padded to a multiple of it)

static inline int pcgs_ksp(int n) { return ((n + 3) / 4 + PCGS_U - 1) / PCGS_U * PCGS_U; }
static inline size_t pcgs_image_bytes(int B, int n) { return (size_t)B * ((n + 15) / 16) * pcgs_ksp(n) * 64 * sizeof(double); }

// image[b][R][s][l] = P_ff[16 R + (l & 15)][4 s + (l >> 4)]  (0 outside the matrix; P_ff = P[1:, 1:])
__global__ __launch_bounds__(256) void pcgs_pack_kernel(const double* __restrict__ PQ, int k1, int k2, int NT, int KSP, double* __restrict__ img) {
    const int b = blockIdx.y, n = k1 - 1;
    const double* P = PQ + (long long)b * (k1 + k2) * k1;
    const long long total = (long long)NT * KSP * 64;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int l = (int)(e & 63);
        const long long rs = e >> 6;
        const int s = (int)(rs % KSP), R = (int)(rs / KSP);
        const int row = 16 * R + (l & 15), col = 4 * s + (l >> 4);
        img[(long long)b * total + e] = (row < n && col < n) ? P[(long long)(row + 1) * k1 + col + 1] : 0.0;
    }
}

__global__ __launch_bounds__(64 * PCGS_NW, 1) void fmap_solve_pcgs_kernel(const double* __restrict__ PQ, const double* __restrict__ img,
                                                                          const double* __restrict__ lam1, const double* __restrict__ lam2,
                                                                          const double* __restrict__ c00, double w_lap, int k1, int k2, int ngroups,
                                                                          int NT, int KSP, double tol2, int maxit, int slow_it, double slow_tol2,
                                                                          double* __restrict__ C, int32_t* __restrict__ fallback, int32_t* __restrict__ info) {
    constexpr int NW = PCGS_NW, RTW = 2, U = PCGS_U;
    extern __shared__ __attribute__((aligned(16))) double pcg_sm[];
    double* pL = pcg_sm;                                  // [16 PCGS_NT][PCG_LDP]
    double* red = pL + PCGS_NT * 16 * PCG_LDP;
    int* s_notpd = reinterpret_cast<int*>(red + 2 * NW * PCG_NS);
    const int vid = xcd_remap(blockIdx.x, gridDim.x);     // the workgroups of a pair share an XCD (its L2 holds the pair's image)
    const int b = vid / ngroups, g = vid - b * ngroups;
    const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = k1 - 1;
    const double* P = PQ + (long long)b * (k1 + k2) * k1;
    const double* Q = P + (long long)k1 * k1;
    const double* l1 = lam1 + (long long)b * k1;
    const double* l2 = lam2 + (long long)b * k2;
    typedef __attribute__((address_space(1))) const double gdouble;
    bool has[RTW];
    gdouble* ap[RTW];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) {
        const int R = wave * RTW + rt;
        has[rt] = R < NT;                                 // (wave-uniform)
        ap[rt] = (gdouble*)img + (((long long)b * NT + (has[rt] ? R : 0)) * KSP) * 64 + lane;
    }

    double mx = -DM_INF_F64;
    for (int q = t; q < k1; q += 64 * NW) mx = fmax(mx, l1[q]);
    for (int q = t; q < k2; q += 64 * NW) mx = fmax(mx, l2[q]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if (lane == 0) red[wave] = mx;
    if (t == 0) *s_notpd = 0;
    __syncthreads();
    double scale = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) scale = fmax(scale, red[w]);
    __syncthreads();

    double x[RTW][2][4], r[RTW][2][4], pv[RTW][2][4], minv[RTW][2][4];
    double l1c[RTW][4], l2c[2];                           // dd = w_lap (l1c - l2c)^2 is recomputed where it is used (registers: 256 per wave here)
    double part[2] = {0.0, 0.0};
    bool notpd = false;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int i = g * PCG_NS + ct * 16 + l15;
        const bool iok = i < k2;
        const double l2i = iok ? l2[i] / scale : 0.0;
        l2c[ct] = l2i;
        const double ci0 = (i == 0) ? c00[b] : 0.0;
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 16 * (wave * RTW + rt) + lg + 4 * q;
                const bool ok = iok && c < n;
                const int cc = ok ? c : 0, ii = ok ? i : 0;
                l1c[rt][q] = l1[(c < n ? c : 0) + 1] / scale;
                const double d = l1c[rt][q] - l2i;
                const double dv = w_lap * (d * d);
                const double m = P[(long long)(cc + 1) * (k1 + 1)] + dv;
                const double bv = Q[(long long)ii * k1 + cc + 1] - P[(long long)(cc + 1) * k1] * ci0;
                if (ok && !(m > 0.0)) notpd = true;
                minv[rt][ct][q] = (ok && m > 0.0) ? 1.0 / m : 0.0;
                x[rt][ct][q] = 0.0;
                r[rt][ct][q] = ok ? bv : 0.0;
                pv[rt][ct][q] = r[rt][ct][q] * minv[rt][ct][q];
                part[ct] = fma(r[rt][ct][q], pv[rt][ct][q], part[ct]);
            }
    }
    if (notpd) *s_notpd = 1;
    pcg_colsum<NW>(part, red, wave, lane);
    notpd = *s_notpd != 0;
    double rz[2] = {part[0], part[1]}, rz0[2] = {part[0], part[1]};
    int rbuf = 1;
    bool done = false;
    int it = 0;
    for (; it < maxit && !done && !notpd; ++it) {
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) pL[(16 * (wave * RTW + rt) + lg + 4 * q) * PCG_LDP + ct * 16 + l15] = pv[rt][ct][q];
        // the first group of A fragments is requested before the barrier: the loads do not depend on the direction block
        double a_cur[U][RTW], a_nxt[U][RTW];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) a_cur[u][rt] = has[rt] ? ap[rt][(long long)u * 64] : 0.0;
        __syncthreads();
        f64x4 acc[RTW][2];
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f64x4{0.0, 0.0, 0.0, 0.0};
        for (int s0 = 0; s0 < KSP; s0 += U) {
            const int sn = (s0 + U < KSP) ? s0 + U : s0;              // (the last group re-reads itself: no branch around the loads)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt) a_nxt[u][rt] = has[rt] ? ap[rt][(long long)(sn + u) * 64] : 0.0;
            // (the B operands are read where they are used: requested a group ahead like the n <= 128 kernel's they cost 27 more spilled
            //  registers here -- 256 per wave -- and 5 % of the kernel, measured)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = s0 + u;
                const double b0 = pL[(4 * s + lg) * PCG_LDP + l15], b1 = pL[(4 * s + lg) * PCG_LDP + 16 + l15];
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt)
                    if (has[rt]) {
                        acc[rt][0] = mfma_f64_16x16x4(a_cur[u][rt], b0, acc[rt][0]);
                        acc[rt][1] = mfma_f64_16x16x4(a_cur[u][rt], b1, acc[rt][1]);
                    }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt) a_cur[u][rt] = a_nxt[u][rt];
        }
        part[0] = part[1] = 0.0;
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double d_ = l1c[rt][q] - l2c[ct];          // (padding entries: pv is 0 there)
                    acc[rt][ct][q] = fma(w_lap * (d_ * d_), pv[rt][ct][q], acc[rt][ct][q]);
                    part[ct] = fma(pv[rt][ct][q], acc[rt][ct][q], part[ct]);
                }
        pcg_colsum<NW>(part, red + rbuf * NW * PCG_NS, wave, lane);
        rbuf ^= 1;
        double alpha[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            if (rz[ct] > 0.0 && !(part[ct] > 0.0)) notpd = true;
            alpha[ct] = (part[ct] > 0.0) ? rz[ct] / part[ct] : 0.0;
        }
        part[0] = part[1] = 0.0;
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x[rt][ct][q] = fma(alpha[ct], pv[rt][ct][q], x[rt][ct][q]);
                    r[rt][ct][q] = fma(-alpha[ct], acc[rt][ct][q], r[rt][ct][q]);
                    const double z = r[rt][ct][q] * minv[rt][ct][q];
                    part[ct] = fma(r[rt][ct][q], z, part[ct]);
                    acc[rt][ct][q] = z;
                }
        pcg_colsum<NW>(part, red + rbuf * NW * PCG_NS, wave, lane);
        rbuf ^= 1;
        bool conv = true;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const double beta = rz[ct] > 0.0 ? part[ct] / rz[ct] : 0.0;
            rz[ct] = part[ct];
            conv = conv && !(rz[ct] > tol2 * rz0[ct]);
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q) pv[rt][ct][q] = fma(beta, pv[rt][ct][q], acc[rt][ct][q]);
        }
        done = __all(conv) != 0;
        notpd = __any(notpd) != 0;
        if (notpd) break;
        if (it + 1 == slow_it) {
            bool slow = false;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) slow = slow || (rz[ct] > slow_tol2 * rz0[ct]);
            if (__any(slow)) break;
        }
    }
    // one flag per (pair, group), written by every workgroup: nothing to clear before the launch; group 0 also clears the pair's status word
    // for the direct solver behind (which reports singular systems there)
    if (t == 0) { fallback[(long long)b * ngroups + g] = (!done || notpd) ? 1 : 0; if (g == 0) info[b] = 0; }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int i = g * PCG_NS + ct * 16 + l15;
        if (i < k2) {
            double* Crow = C + ((long long)b * k2 + i) * k1;
            if (wave == 0 && lg == 0) Crow[0] = (i == 0) ? c00[b] : 0.0;
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 16 * (wave * RTW + rt) + lg + 4 * q;
                    if (c < n) Crow[c + 1] = x[rt][ct][q];
                }
        }
    }
}

// Blocked Cholesky solve of one SPD system held in LDS (shared by dm_fmap.hip and dm_icp.hip).
//
// The matrix lives in LDS as 16x16 blocks of the lower block triangle, block (I,K) at (I(I+1)/2 + K) * 256,
// each block stored TRANSPOSED (T_IK[k][i] = A[I*16+i][K*16+k]) so that every f64-MFMA operand and result access
// is a lane-contiguous, conflict-free ds_read/ds_write_b64.  Per block column J:
//   (a) wave 0: elimination of the 16x16 diagonal block in registers, giving W = L_JJ^-1 (the only serial part),
//   (b) panel:   L_IJ = A_IJ L_JJ^-T        ==  T_IJ <- W T_IJ                    (MFMA)
//   (c) update:  A_IK -= L_IJ L_KJ^T        ==  T_IK <- T_IK - L_KJ L_IJ^T        (MFMA)
// the right-hand side is carried along (forward solve), then L^T x = y runs block-backwards with the stored W_J.
#pragma once
#include "dm_device.h"

#ifndef DBG_ACC
#define DBG_ACC(slot)
#endif
// phase counters of tools/solve_timing.py (a -DDM_SOLVE_TIMING build of dm_fmap.hip defines g_solve_dbg before this header)
#if defined(DM_SOLVE_TIMING) && defined(DM_SOLVE_TIMING_HAVE_COUNTERS)
#define CH_T0() long long _c0 = clock64(); const bool _cd = (blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0);
#define CH_ACC(slot) { const long long _c1 = clock64(); if (_cd) g_solve_dbg[slot] += _c1 - _c0; _c0 = _c1; }
#else
#define CH_T0()
#define CH_ACC(slot)
#endif

// ---- register-resident factorisation of one 16x16 diagonal block (one wave) ------------------------------
// Lane (r = lane & 15, g = lane >> 4) holds columns 4g..4g+3 of row r of the symmetric block S and of the identity
// block E.  Step J eliminates column J in square-root-free form (S' = S - v v^T / piv: only a reciprocal sits on
// the dependency chain), applied to E as row operations; scaling column J by piv^-1/2 turns E into L^-T (= W^T).
// Row J reaches the lanes through DPP row_newbcast, column J through one ds_bpermute per operand.
template <int J>
__device__ __forceinline__ double dpp_row_bcast(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + J, 0xf, 0xf, false);   // v_mov_b32_dpp row_newbcast:J
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + J, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double x, int srclane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), srclane);
    return __hiloint2double(hi, lo);
}
template <int J>
__device__ __forceinline__ void diag_step(double (&s)[4], double (&w)[4], double& pv, int lane, bool& ok) {
    constexpr int gj = J >> 2, ej = J & 3;
    const double piv = readlane_f64(s[ej], J | (gj << 4));                  // S[J][J], wave uniform
    ok = ok && (piv > 0.0);
    double rp = __builtin_amdgcn_rcp(piv);                                   // 1 / piv   (seed + 2 Newton steps)
    rp = fma(fma(-piv, rp, 1.0), rp, rp);
    rp = fma(fma(-piv, rp, 1.0), rp, rp);
    pv = (lane == J) ? piv : pv;                                             // lane J keeps pivot J: the piv^-1/2 column scaling is
                                                                             // done once for all 16 columns after the chain
    const int src = (lane & 15) | (gj << 4);
    const double v = __shfl(s[ej], src);                                     // S[r][J]
    const double z = __shfl(w[ej], src);                                     // E[r][J]
    const double vr = -v * rp, zr = -z * rp;
    // No column mask: for finished columns c < J the broadcast row entry S[J][c] is a rounding-level residue of
    // its own elimination (S[r][c] (1 - piv rp)), so touching them perturbs the result by O(1e-16) only; column J
    // of E is final before this step's update (restored below).
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double t = dpp_row_bcast<J>(s[e]);                             // S[J][4g+e]
        s[e] = fma(vr, t, s[e]);
        w[e] = fma(zr, t, w[e]);
    }
    if ((lane >> 4) == gj) w[ej] = z;                                        // column J of E: E[r][J], scaled by piv^-1/2 at the end
}


// diagonal block J: in-register elimination by wave 0 (see diag_step); leaves Ws[m][k] = W[k][m], W = L_JJ^-1.
// (Tried on MI355X, none faster than this form: a row-per-lane form with the pivot row through DPP row_newbcast 6.7 k
// cycles per block against 5.9 k -- no LDS on the chain, but 700 instead of 400 f64 VALU instructions --, through
// v_readlane 8.4 k; the column broadcast by v_permlane32_swap / v_permlane16_swap instead of ds_bpermute 7.4 k.
// The 16 inverse square roots are taken together after the chain: one rsq + Newton sequence instead of sixteen.)
__device__ __forceinline__ void diag_block(const double* S, double* Ws, double* red, int lane) {
    const int r = lane & 15, g = lane >> 4;
    double ds[4], dw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ds[e] = S[r * 16 + 4 * g + e];
        dw[e] = (r == 4 * g + e) ? 1.0 : 0.0;
    }
    bool ok = true;
    double pv = 1.0;
    diag_step<0>(ds, dw, pv, lane, ok);   diag_step<1>(ds, dw, pv, lane, ok);
    diag_step<2>(ds, dw, pv, lane, ok);   diag_step<3>(ds, dw, pv, lane, ok);
    diag_step<4>(ds, dw, pv, lane, ok);   diag_step<5>(ds, dw, pv, lane, ok);
    diag_step<6>(ds, dw, pv, lane, ok);   diag_step<7>(ds, dw, pv, lane, ok);
    diag_step<8>(ds, dw, pv, lane, ok);   diag_step<9>(ds, dw, pv, lane, ok);
    diag_step<10>(ds, dw, pv, lane, ok);  diag_step<11>(ds, dw, pv, lane, ok);
    diag_step<12>(ds, dw, pv, lane, ok);  diag_step<13>(ds, dw, pv, lane, ok);
    diag_step<14>(ds, dw, pv, lane, ok);  diag_step<15>(ds, dw, pv, lane, ok);
    double inv = __builtin_amdgcn_rsq(pv);                                   // lane c < 16: piv_c^-1/2
    const double hp = 0.5 * pv;
    inv = inv * (1.5 - hp * inv * inv);
    inv = inv * (1.5 - hp * inv * inv);
#pragma unroll
    for (int e = 0; e < 4; ++e) Ws[r * 16 + 4 * g + e] = dw[e] * __shfl(inv, 4 * g + e);
    if (!ok && lane == 0) red[5] = 1.0;
}

struct TriSlots {                             // lower block triangle, block (I, K) at (I (I + 1) / 2 + K) * 256
    __device__ __forceinline__ int operator()(int I, int K) const { return (I * (I + 1) / 2 + K) * 256; }
};

// back substitution  L^T x = y  over block rows NB-1 .. 0:  x_J = W_J^T (y_J - sum_{I>J} L_IJ^T x_I); the W_J are parked in
// the diagonal slots.  One thread per output entry, serial 16-term dot products (no cross-lane reduction chains).
// All 256 threads call it.
template <class Slots>
__device__ __forceinline__ void blocked_back_subst(const double* T, Slots sl, double* rhs, double* xv, int NB, int t) {
    for (int J = NB - 1; J >= 0; --J) {
        const double* Wt = T + sl(J, J);                          // Wt[m*16 + k] = W[k][m]
        if (t < 16) {
            // x_J[k] = sum_m W[m][k] y[m] = sum_m Wt[k*16 + m] y[m]
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int m = 0; m < 16; m += 2) {
                a0 = fma(Wt[t * 16 + m], rhs[J * 16 + m], a0);
                a1 = fma(Wt[t * 16 + m + 1], rhs[J * 16 + m + 1], a1);
            }
            xv[J * 16 + t] = a0 + a1;
        }
        __syncthreads();
        // y_K -= L_JK^T x_J for K < J:  (L_JK^T x)[k] = sum_i T_JK[k*16 + i] x_J[i]
        if (t < J * 16) {
            const int K = t >> 4, k = t & 15;
            const double* Tjk = T + sl(J, K);
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int ii = 0; ii < 16; ii += 2) {
                a0 = fma(Tjk[k * 16 + ii], xv[J * 16 + ii], a0);
                a1 = fma(Tjk[k * 16 + ii + 1], xv[J * 16 + ii + 1], a1);
            }
            rhs[K * 16 + k] -= a0 + a1;
        }
        __syncthreads();
    }
}

// T: blocks; Ws: 256-double scratch (W^T of the current block); rhs: NB*16 (in: right-hand side, becomes y);
// xv: NB*16 (out: solution); red[5]: failure flag (must be 0 on entry); tri_rc: 128 ints, (u -> row << 8 | col) of a
// lower-triangular enumeration with tri_rc[0] = (0, 0).  All 256 threads call it.  Returns false when a pivot was
// not positive.
//
// Look-ahead: the diagonal elimination is a serial chain on one wave (16 dependent steps), so during the trailing
// update of column J wave 0 only refreshes block (J+1, J+1) and immediately eliminates it, while waves 1-3 update the
// rest of the trailing triangle: the chain of column J+1 hides behind the MFMA work of column J.
__device__ __forceinline__ bool blocked_chol_solve(double* T, double* Ws, double* rhs, double* xv, double* red,
                                                   const int* tri_rc, int NB, int t, int lane, int wave) {
    CH_T0()
    if (wave == 0) diag_block(T, Ws, red, lane);
    __syncthreads();
    CH_ACC(1)
    for (int J = 0; J < NB; ++J) {
        if (red[5] != 0.0) break;                // uniform: set before the barrier that closed the previous phase
        double* S = T + (J * (J + 1) / 2 + J) * 256;
        // forward solve of this block of the right-hand side, y_J = W rhs_J (16 lanes of the last wave, beside the panel)
        if (t >= 240) {
            const int k = t - 240;
            double y = 0.0;
#pragma unroll
            for (int m = 0; m < 16; ++m) y = fma(Ws[m * 16 + k], rhs[J * 16 + m], y);    // W[k][m]
            xv[J * 16 + k] = y;              // parked in xv: the other lanes still read rhs_J
        }
        // ---- (b) panel: T_IJ <- W T_IJ for I > J, one block per wave at a time
        for (int I = J + 1 + wave; I < NB; I += 4) {
            double* Tij = T + (I * (I + 1) / 2 + J) * 256;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            double bq[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bq[ks] = Tij[((lane >> 4) + 4 * ks) * 16 + (lane & 15)];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double a = Ws[((lane >> 4) + 4 * ks) * 16 + (lane & 15)];
                acc = mfma_f64_16x16x4(a, bq[ks], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Tij[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
        }
        S[t] = Ws[t];                            // W_J parked in the (now free) diagonal slot: Ws is reused below
        __syncthreads();
        CH_ACC(2)
        if (t < 16) rhs[J * 16 + t] = xv[J * 16 + t];
        // ---- (c) trailing update and right-hand-side update (waves 1-3) | block (J+1, J+1) and its elimination (wave 0)
        {
            const int m = NB - 1 - J;                 // remaining block rows
            const int nupd = m * (m + 1) / 2;
            const int o0 = (lane >> 4) * 16 + (lane & 15);
            if (wave == 0) {
                if (m > 0) {
                    const int I = J + 1;
                    double* Tii = T + (I * (I + 1) / 2 + I) * 256;
                    const double* Tij = T + (I * (I + 1) / 2 + J) * 256;
                    f64x4 acc;
                    double op[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = Tii[o0 + r * 64];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) op[ks] = Tij[o0 + ks * 64];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) acc = mfma_f64_16x16x4(-op[ks], op[ks], acc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) Tii[o0 + r * 64] = acc[r];
                    diag_block(Tii, Ws, red, lane);   // (same wave wrote Tii: the LDS queue of a wave is in order)
                }
            } else {
                // four independent blocks per wave per pass: their LDS reads and MFMA chains overlap
                for (int u0 = 1 + (wave - 1) * 4; u0 < nupd; u0 += 12) {
                    f64x4 acc[4];
                    int off_ik[4], off_ij[4], off_kj[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int u = min(u0 + q, nupd - 1);
                        const int rc = tri_rc[u];
                        const int I = J + 1 + (rc >> 8), K = J + 1 + (rc & 255);             // J < K <= I < NB
                        off_ik[q] = (I * (I + 1) / 2 + K) * 256 + o0;
                        off_ij[q] = (I * (I + 1) / 2 + J) * 256 + o0;
                        off_kj[q] = (K * (K + 1) / 2 + J) * 256 + o0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[q][r] = T[off_ik[q] + r * 64];
                    }
                    double opa[4][4], opb[4][4];       // all LDS operand reads first: their latency overlaps the MFMA chains
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            opa[q][ks] = -T[off_kj[q] + ks * 64];
                            opb[q][ks] = T[off_ij[q] + ks * 64];
                        }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q] = mfma_f64_16x16x4(opa[q][ks], opb[q][ks], acc[q]);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (u0 + q < nupd) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) T[off_ik[q] + r * 64] = acc[q][r];
                        }
                }
                // rhs_I -= L_IJ y_J   (row ii of block I: sum_k T_IJ[k][ii] y_J[k])
                for (int e = t - 64; e < m * 16; e += 192) {
                    const int I = J + 1 + (e >> 4), ii = e & 15;
                    const double* Tij = T + (I * (I + 1) / 2 + J) * 256;
                    double sacc = 0.0;
#pragma unroll
                    for (int k = 0; k < 16; ++k) sacc += Tij[k * 16 + ii] * xv[J * 16 + k];
                    rhs[I * 16 + ii] -= sacc;
                }
            }
        }
        __syncthreads();
        CH_ACC(7)
    }
    __syncthreads();
    if (red[5] != 0.0) return false;
    blocked_back_subst(T, TriSlots{}, rhs, xv, NB, t);
    return true;
}

// block offsets of the two-phase layout (fmap_solve_2phase_kernel): the NA leading block rows as a lower block triangle,
// then the remaining block rows' first NA block columns as a dense panel
struct PanelSlots {
    int NA;
    __device__ __forceinline__ int operator()(int I, int K) const {          // K < NA
        return (I < NA ? I * (I + 1) / 2 + K : NA * (NA + 1) / 2 + (I - NA) * NA + K) * 256;
    }
};

// Columns J = 0 .. NA-1 of the blocked factorisation of an NB x NB block matrix of which only the block columns < NA are
// resident (slots sl(I, K), K < NA <= NB).  Same phases and look-ahead as blocked_chol_solve; the trailing update stops
// at block column NA - 1.  Afterwards L11 / the W_J sit in the leading slots, L21 in the panel slots,
// rhs[0 .. NA) = y1 and rhs[NA .. NB) = b2 - L21 y1.  blk: K-major list of the resident blocks ((I << 8) | K, K < NA,
// K <= I < NB); cstart[K] = position of (K, K) in it, cstart[NA] = its length.  red[5] must be 0 on entry.
__device__ __forceinline__ bool blocked_chol_phase1(double* T, PanelSlots sl, double* Ws, double* rhs, double* xv, double* red,
                                                    const int* blk, const int* cstart, int NB, int t, int lane, int wave) {
    const int NA = sl.NA;
    if (wave == 0) diag_block(T + sl(0, 0), Ws, red, lane);
    __syncthreads();
    for (int J = 0; J < NA; ++J) {
        if (red[5] != 0.0) break;                // uniform: set before the barrier that closed the previous phase
        double* S = T + sl(J, J);
        if (t >= 240) {
            const int k = t - 240;
            double y = 0.0;
#pragma unroll
            for (int m = 0; m < 16; ++m) y = fma(Ws[m * 16 + k], rhs[J * 16 + m], y);    // W[k][m]
            xv[J * 16 + k] = y;
        }
        for (int I = J + 1 + wave; I < NB; I += 4) {
            double* Tij = T + sl(I, J);
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            double bq[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bq[ks] = Tij[((lane >> 4) + 4 * ks) * 16 + (lane & 15)];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double a = Ws[((lane >> 4) + 4 * ks) * 16 + (lane & 15)];
                acc = mfma_f64_16x16x4(a, bq[ks], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Tij[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
        }
        S[t] = Ws[t];                            // W_J parked in the diagonal slot
        __syncthreads();
        if (t < 16) rhs[J * 16 + t] = xv[J * 16 + t];
        {
            const int m = NB - 1 - J;
            const int o0 = (lane >> 4) * 16 + (lane & 15);
            if (wave == 0) {
                if (J + 1 < NA) {
                    const int I = J + 1;
                    double* Tii = T + sl(I, I);
                    const double* Tij = T + sl(I, J);
                    f64x4 acc;
                    double op[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = Tii[o0 + r * 64];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) op[ks] = Tij[o0 + ks * 64];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) acc = mfma_f64_16x16x4(-op[ks], op[ks], acc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) Tii[o0 + r * 64] = acc[r];
                    diag_block(Tii, Ws, red, lane);
                }
            } else {
                const int uend = cstart[NA];
                for (int u0 = cstart[J + 1] + 1 + (wave - 1) * 4; u0 < uend; u0 += 12) {
                    f64x4 acc[4];
                    int off_ik[4], off_ij[4], off_kj[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rc = blk[min(u0 + q, uend - 1)];
                        const int I = rc >> 8, K = rc & 255;                                  // J < K <= I < NB, K < NA
                        off_ik[q] = sl(I, K) + o0;
                        off_ij[q] = sl(I, J) + o0;
                        off_kj[q] = sl(K, J) + o0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[q][r] = T[off_ik[q] + r * 64];
                    }
                    double opa[4][4], opb[4][4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            opa[q][ks] = -T[off_kj[q] + ks * 64];
                            opb[q][ks] = T[off_ij[q] + ks * 64];
                        }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q] = mfma_f64_16x16x4(opa[q][ks], opb[q][ks], acc[q]);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (u0 + q < uend) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) T[off_ik[q] + r * 64] = acc[q][r];
                        }
                }
                for (int e = t - 64; e < m * 16; e += 192) {
                    const int I = J + 1 + (e >> 4), ii = e & 15;
                    const double* Tij = T + sl(I, J);
                    double sacc = 0.0;
#pragma unroll
                    for (int k = 0; k < 16; ++k) sacc += Tij[k * 16 + ii] * xv[J * 16 + k];
                    rhs[I * 16 + ii] -= sacc;
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    return red[5] == 0.0;
}

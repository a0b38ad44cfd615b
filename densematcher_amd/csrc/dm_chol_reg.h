// Register-resident blocked Cholesky solve: ONE WAVE = ONE SPD system, the whole lower block triangle in VGPRs.
//
// Why: the LDS-resident solver (dm_chol.h) is bound by the serial pivot chain of the 16x16 diagonal blocks and LDS holds
// only two 72 KiB systems per CU, so at most two chains are in flight per CU while the matrix cores idle.  A CU has
// 512 KiB of vector registers: a system of order <= 128 (36 blocks of 16 x 16 float64 = 288 VGPRs) fits the register
// budget of one wave (512 per lane at one wave per SIMD), so FOUR independent systems run per CU, one per SIMD, with no
// LDS image, no barriers and no LDS operand traffic at all: every block lives in the accumulator layout of
// v_mfma_f64_16x16x4_f64, and that layout IS the operand layout of the products the factorisation needs:
//
//   block (I, K), stored transposed as in dm_chol.h (T_IK[k][i] = A[16 I + i][16 K + k]), register r of lane (c, g)
//   [c = lane & 15, g = lane >> 4] holds T_IK[g + 4 r][c] = L_IK[c][g + 4 r].  Used as the MFMA "a" operand that register
//   set is the matrix L_IK (a supplies A[i = c][k = g + 4 ks]); used as the "b" operand it is L_IK^T = T_IK.
//     panel      T_IJ <- W T_IJ             a = W (see below), b = T_IJ
//     trailing   T_IK <- T_IK - L_KJ L_IJ^T   a = -T_KJ registers, b = T_IJ registers
//   and the elimination of a diagonal block runs on its accumulator registers as they are (the block is symmetric) and
//   leaves W = L_JJ^-1 as W[c][g + 4 e] -- the "a" operand of the panel product -- without a transpose: the inverse is
//   accumulated by ROW operations (dm_chol.h uses column operations and needs W^T in LDS).
// The forward substitution runs on the matrix cores too (vectors as blocks whose first column is used); the back
// substitution on the vector ALU (row reductions by DPP: its matrix products would need the transposed register layout).
#pragma once
#include "dm_device.h"

// phase counters of tools/ubench_solve_reg.hip (-DDMREG_TIMING: it defines `__device__ long long g_dmreg_t[8]` first);
// nothing in the library build
#ifdef DMREG_TIMING
#define DMREG_VARS long long dmreg_acc[6] = {0, 0, 0, 0, 0, 0}; long long dmreg_last = __builtin_readcyclecounter();
#define DMREG_MARK(slot)                                                                      \
    {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        const long long t_ = __builtin_readcyclecounter();                                    \
        dmreg_acc[slot] += t_ - dmreg_last; dmreg_last = t_;                                  \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    }
#define DMREG_OUT if (blockIdx.x == 0 && threadIdx.x == 0) { for (int q_ = 0; q_ < 6; ++q_) g_dmreg_t[q_] = dmreg_acc[q_]; }
#else
#define DMREG_VARS
#define DMREG_MARK(slot)
#define DMREG_OUT
#endif

namespace dmreg {

template <int J>
__device__ __forceinline__ double row_bcast(double x) {                 // value of lane J of this lane's 16-lane row
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + J, 0xf, 0xf, false);   // row_newbcast:J
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + J, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ double row_ror(double x) {                   // value of lane (c + N) & 15 ... rotate right within the row
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x120 + N, 0xf, 0xf, false);   // row_ror:N
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x120 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// acc += (acc of lane J of this lane's row) * m in ONE instruction: gfx90a+ has a VOP2 v_fmac_f64 and 64-bit DPP with the
// row_newbcast control, so the broadcast of the pivot row is an operand modifier of the update itself instead of two
// v_mov_b32_dpp in front of a v_fma_f64 (bit-identical: fma(bcast * m + acc)).  gfx9 has no hardware interlock for a DPP read
// of a VGPR the previous VALU instruction wrote (2 wait states) and the compiler cannot see into the asm: s_nop 1 in front.
template <int J>
__device__ __forceinline__ void fmac_row_bcast(double& acc, double m) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "n"(J));
}
// sum over the 16 lanes of a row, result in every lane (fixed order: identical in all lanes of the row)
__device__ __forceinline__ double row_allsum(double x) {
    x += row_ror<8>(x);
    x += row_ror<4>(x);
    x += row_ror<2>(x);
    x += row_ror<1>(x);
    return x;
}
__device__ __forceinline__ double readlane_f64(double x, int srclane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), srclane);
    return __hiloint2double(hi, lo);
}

// 1 / piv: v_rcp_f64 is good to 2^-24.4 (tools/ubench_solve_reg.hip); one cubic step r (1 + e + e^2), e = 1 - piv r, leaves
// e^3 = 2^-73: three dependent fused multiply-adds on the pivot chain instead of the four of two Newton steps
__device__ __forceinline__ double rcp_refined(double piv) {
    const double rp = __builtin_amdgcn_rcp(piv);
    const double er = fma(-piv, rp, 1.0);
    return fma(rp, fma(er, er, er), rp);
}

// One step of the square-root-free elimination of a symmetric 16 x 16 block S held as s[e] = S[c][g + 4 e]; the same row
// operations are applied to E (identity at the start): after the 16 steps E = Ltilde^-1 (unit lower triangular) and
// W = diag(piv^-1/2) E = L^-1.  Row J reaches the lanes of a row through DPP row_newbcast, column J (the value S[c][J] of
// this lane's row) lives in group J & 3 and comes across the groups through one ds_bpermute.
// The step is ISSUE-bound, not latency-bound (one wave per SIMD, float64 VALU instructions take 8 cycles each: ~200 cycles per
// pivot): a software-pipelined form that fetched column J + 1 and the next pivot ahead of the row updates (six more lane reads,
// three more multiply-adds) measured the same 26 k cycles per system (tools/ubench_solve_reg.hip, profiles/r03_solver_reg_phases.txt).
template <int J>
__device__ __forceinline__ void elim_step(double (&s)[4], double (&w)[4], double& pv, int c, int lane, bool& ok) {
    constexpr int gj = J & 3, ej = J >> 2;
    const double piv = readlane_f64(s[ej], J | (gj << 4));                   // S[J][J], wave uniform
    ok = ok && (piv > 0.0);
    const double rp = rcp_refined(piv);
    pv = (c == J) ? piv : pv;                                                // the lanes of row J keep pivot J
    const double v = __shfl(s[ej], c | (gj << 4));                           // S[c][J]
    double vr = -v * rp;
    double vrE = (c == J) ? 0.0 : vr;                                        // row J of E is final (pivot row)
    // rows c < J: v is the rounding-level residue of their own elimination (row J itself takes vr = -1 and is cleared), they
    // stay as they are up to O(1e-16).
    // S[c][g + 4 e] -= v S[J][g + 4 e] / piv for the live columns (4 e + 3 > J); E[J][g + 4 e] can be non-zero only for
    // g + 4 e <= J (4 e <= J): always five updates, issued as one group behind a single hazard wait (each reads, through DPP,
    // only the register it accumulates into, written by the previous step)
    constexpr int E0 = J >> 2;                                               // s[E0 .. 3], w[0 .. E0]
    if constexpr (E0 == 0)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %1, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %2, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %3, %3, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %4, %4, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                     : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(w[0]) : "v"(vr), "v"(vrE), "n"(J));
    else if constexpr (E0 == 1)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %1, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %2, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %3, %3, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %4, %4, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                     : "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(w[0]), "+v"(w[1]) : "v"(vr), "v"(vrE), "n"(J));
    else if constexpr (E0 == 2)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %1, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %2, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %3, %3, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %4, %4, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                     : "+v"(s[2]), "+v"(s[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]) : "v"(vr), "v"(vrE), "n"(J));
    else
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %1, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %2, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %3, %3, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %4, %4, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                     : "+v"(s[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "v"(vr), "v"(vrE), "n"(J));
}

// diagonal block (accumulator registers) -> W = L^-1 as W[c][g + 4 e]; ok = every pivot positive
__device__ __forceinline__ void elim_block(const f64x4& S, f64x4& W, int c, int g, int lane, bool& ok) {
    double s[4], w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s[e] = S[e];
        w[e] = (c == g + 4 * e) ? 1.0 : 0.0;
    }
    double pv = 1.0;
    elim_step<0>(s, w, pv, c, lane, ok);   elim_step<1>(s, w, pv, c, lane, ok);
    elim_step<2>(s, w, pv, c, lane, ok);   elim_step<3>(s, w, pv, c, lane, ok);
    elim_step<4>(s, w, pv, c, lane, ok);   elim_step<5>(s, w, pv, c, lane, ok);
    elim_step<6>(s, w, pv, c, lane, ok);   elim_step<7>(s, w, pv, c, lane, ok);
    elim_step<8>(s, w, pv, c, lane, ok);   elim_step<9>(s, w, pv, c, lane, ok);
    elim_step<10>(s, w, pv, c, lane, ok);  elim_step<11>(s, w, pv, c, lane, ok);
    elim_step<12>(s, w, pv, c, lane, ok);  elim_step<13>(s, w, pv, c, lane, ok);
    elim_step<14>(s, w, pv, c, lane, ok);  elim_step<15>(s, w, pv, c, lane, ok);
    double inv = __builtin_amdgcn_rsq(pv);                                   // piv_c^-1/2 of this lane's row
    const double hp = 0.5 * pv;
    inv = inv * (1.5 - hp * inv * inv);
    inv = inv * (1.5 - hp * inv * inv);
#pragma unroll
    for (int e = 0; e < 4; ++e) W[e] = w[e] * inv;
}

// column layout (value[g + 4 r] in register r of every lane of group g) -> row layout (value[c] in every lane with
// lane & 15 == c): pick register c >> 2 in the lanes of group c & 3, fetch it from there
__device__ __forceinline__ double col_to_row(const f64x4& v, int c) {
    const int r = c >> 2;
    const double pick = r == 0 ? v[0] : (r == 1 ? v[1] : (r == 2 ? v[2] : v[3]));
    return __shfl(pick, c | ((c & 3) << 4));
}

constexpr int blk(int I, int K) { return I * (I + 1) / 2 + K; }

// T: the NB (NB + 1) / 2 blocks of the lower block triangle.  rhs(J) returns the right-hand side of block row J in column
// layout restricted to the lanes c == 0 (register r = rhs[16 J + g + 4 r] there, 0 elsewhere); store(J, x) receives the
// solution, x[r] = x[16 J + g + 4 r] in every lane (column layout).  Returns false when a pivot was not positive (nothing
// is stored then).
//
// EV > 0: the finished panel blocks L_IJ (I > J) of the first EV block columns are parked in `ev` (wave-private LDS, 2 KiB per
// block) between their trailing update and the back substitution -- nothing reads them in between.  At NB = 8 the 36 blocks,
// the right-hand sides and the elimination state exceed what the register allocator fits into 256 + 256 registers; without
// this it spilled 89 dwords per lane to scratch (190 MB of HBM writes per 8192 systems).
constexpr int ev_slot(int NB, int J, int I) { return J * (NB - 1) - J * (J - 1) / 2 + (I - J - 1); }
__device__ __forceinline__ void ev_put(double* ev, int slot, const f64x4& v, int lane) {
    double* p = ev + slot * 256 + lane * 2;
    *reinterpret_cast<f64x2*>(p) = f64x2{v[0], v[1]};
    *reinterpret_cast<f64x2*>(p + 128) = f64x2{v[2], v[3]};
}
__device__ __forceinline__ f64x4 ev_get(const double* ev, int slot, int lane) {
    const double* p = ev + slot * 256 + lane * 2;
    const f64x2 a = *reinterpret_cast<const f64x2*>(p), b = *reinterpret_cast<const f64x2*>(p + 128);
    return f64x4{a[0], a[1], b[0], b[1]};
}
template <int NB, int EV = 0, class Rhs, class Store>
__device__ __forceinline__ bool solve(f64x4 (&T)[NB * (NB + 1) / 2], Rhs rhs, Store store, int lane, double* ev = nullptr) {
    const int c = lane & 15, g = lane >> 4;
    bool ok = true;
    f64x4 W;
    DMREG_VARS
    DMREG_MARK(0)
    elim_block(T[blk(0, 0)], W, c, g, lane, ok);
    DMREG_MARK(1)
    // The forward substitution L y = b rides along on the matrix cores: the vectors are 16 x 16 blocks of which only the first
    // column is used (lanes c == 0), which is at once the accumulator layout and the "b" operand layout,
    //   Y_J = W_J (B_J - sum_{K < J} L_JK Y_K)        a = L_JK (its registers as they are), a = W_J
    // right-looking (once Y_J is known every block row below takes its update) and INSIDE the factorisation loop: its
    // dependent accumulations (four matrix instructions per product) hide among the independent ones of the trailing update;
    // on their own at the end they ran at half the issue rate (tools/ubench_solve_reg.hip).
    f64x4 Y[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) Y[J] = rhs(J);
#pragma unroll
    for (int J = 0; J < NB; ++J) {
        // ---- panel: T_IJ <- W T_IJ for the block rows below
#pragma unroll
        for (int I = J + 1; I < NB; ++I) {
            f64x4& Tij = T[blk(I, J)];
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = mfma_f64_16x16x4(W[ks], Tij[ks], acc);
            Tij = acc;
        }
        T[blk(J, J)] = W;                                   // W_J parked in the diagonal slot for the back substitution
        {
            f64x4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) y = mfma_f64_16x16x4(W[ks], Y[J][ks], y);
            Y[J] = y;
        }
        DMREG_MARK(2)
        // ---- block (J+1, J+1) first, and its elimination: the chain of the next column is issued in front of the rest of
        // the trailing update, whose matrix instructions do not depend on it
        if (J + 1 < NB) {
            const f64x4& Tij = T[blk(J + 1, J)];
            f64x4 acc = T[blk(J + 1, J + 1)];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = mfma_f64_16x16x4(-Tij[ks], Tij[ks], acc);
            DMREG_MARK(3)
            elim_block(acc, W, c, g, lane, ok);
            DMREG_MARK(1)
        }
        // ---- trailing update: T_IK <- T_IK - L_KJ L_IJ^T, J < K <= I < NB, and of the right-hand side Y_I -= L_IJ Y_J
#pragma unroll
        for (int I = J + 1; I < NB; ++I) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) Y[I] = mfma_f64_16x16x4(-T[blk(I, J)][ks], Y[J][ks], Y[I]);
#pragma unroll
            for (int K = J + 1; K <= I; ++K) {
                if (I == J + 1 && K == J + 1) continue;
                const f64x4& Tkj = T[blk(K, J)];
                const f64x4& Tij = T[blk(I, J)];
                f64x4 acc = T[blk(I, K)];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = mfma_f64_16x16x4(-Tkj[ks], Tij[ks], acc);
                T[blk(I, K)] = acc;
            }
        }
        if (J < EV) {
#pragma unroll
            for (int I = J + 1; I < NB; ++I) ev_put(ev, ev_slot(NB, J, I), T[blk(I, J)], lane);
        }
        DMREG_MARK(3)
    }
    if (!ok) return false;
    DMREG_MARK(4)
    // ---- back substitution L^T x = y, block rows NB-1 .. 0, on the vector ALU (the matrix products would need the
    // transposed register layout):
    //   u = sum_{I > J} L_IJ^T x_I:  u[g + 4 r] = sum_c T_IJ[g + 4 r][c] x_I[c]   (products in-lane, ONE row reduction per J)
    //   x_J = W_J^T (y_J - u):       x_J[g + 4 e] = sum_c W[c][g + 4 e] z[c]
    double xrow[NB];                                        // x_I[c] (row layout)
#pragma unroll
    for (int J = NB - 1; J >= 0; --J) {
        f64x4 z;
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = row_bcast<0>(Y[J][r]);            // y_J[g + 4 r] from the lanes c == 0
        if (J + 1 < NB) {
            f64x4 u = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int I = J + 1; I < NB; ++I) {
                const f64x4 Lij = (J < EV) ? ev_get(ev, ev_slot(NB, J, I), lane) : T[blk(I, J)];
#pragma unroll
                for (int r = 0; r < 4; ++r) u[r] = fma(Lij[r], xrow[I], u[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] -= row_allsum(u[r]);
        }
        const double zrow = col_to_row(z, c);
        f64x4 x;
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = row_allsum(T[blk(J, J)][e] * zrow);
        store(J, x);
        if (J > 0) xrow[J] = col_to_row(x, c);
    }
    DMREG_MARK(5)
    DMREG_OUT
    return true;
}

}   // namespace dmreg

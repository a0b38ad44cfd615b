// knn21[i] = argmin_j  n1_j - 2 <x_i, y_j>     (x_i = row i of Phi2, y_j = row j of Phi1 C^T; the ZoomOut / ICP map)
// in two passes instead of the float64 G kernel:
//
//  1. fp16 matrix cores.  Both operands are scaled by a power of two (max |.| in [1, 2)) and split in two fp16 pieces,
//     x = xh + xl, y = yh + yl (22 significant bits), and laid out as fp16 "feature" rows
//         target i : [  1,  1,  1, 0 x5 | xh_0, xh_0, xl_0 | xh_1, xh_1, xl_1 | ... | 0 pad ]
//         source j : [ b0, b1, b2, 0 x5 | yh_0, yl_0, yh_0 | yh_1, yl_1, yh_1 | ... | 0 pad ]      b0+b1+b2 = -n1_j sx sy / 2
//     so that one fp16 inner product (exact products, fp32 accumulation) is
//         sx sy (<x_i, y_j> - n1_j / 2) - <xl, yl> - split residuals
//     and the arg-max of it over j is the wanted arg-min up to a bounded error.  The tile kernel, the top-2 bookkeeping
//     and the per-wave top-2 partials are those of dm_simnn_f16 (dm_simnn_core).
//  2. every row whose (best - second) is inside twice the error bound is re-evaluated exactly: float64 products of
//     the original operands, only over the 32-source blocks that the partials cannot rule out (dm_simnn_keep), lowest
//     index on ties.
//     (the split runs on hardware conversions, f64 -> f32 -> f16, dm_split.h: split2_hw -- a direct f64 -> f16 conversion is a
//      ~30-instruction software routine and was most of the row builders' time; the residual stays inside the budget below)
//     Bound, relative to |t_i| max_j |s_j| (>= |x~_i| |y~_j|): fp32 accumulation D (1 + 1/16) 2^-23 (dm_simnn_core)
//     + for the split: dropped <xl, yl> <= 2^-22, two residuals <= 2^-22 each, fp16 subnormal floor <= 2 sqrt(K) 2^-25
//     (computed from the depth K of the search, 25 % slack; about 2^-19 at K = 200), bias pieces 2^-32.
//     If -n1_j sx sy / 2 does not fit fp16 (operand scales more than ~2^15 apart) the pair is flagged as a whole and
//     every row takes the exact path: slower, never wrong.
#include "dm_device.h"
#include "dm_internal.h"
#include "dm_split.h"
#include "dm_exact.h"

static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

constexpr int KS_NCH = DM_NCH; // partial maxima per pair of the target operand
constexpr int KS_BIAS = 8;     // halves reserved in front of the split entries (three used): keeps them 16-byte aligned

// amax[b * KS_NCH + chunk] = max |AT[r][c]| over rows r = chunk (mod KS_NCH), r < K
__global__ __launch_bounds__(256) void ks_absmax_kernel(const double* __restrict__ AT, int K, int N2, int N2pad, int Kpad,
                                                        double* __restrict__ amax) {
    __shared__ double sh[4];
    const int chunk = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const double* M = AT + (long long)b * Kpad * N2pad;
    double m = 0.0;
    for (int r = chunk; r < K; r += KS_NCH)
        for (int c = t; c < N2; c += 256) m = fmax(m, fabs(M[(long long)r * N2pad + c]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((t & 63) == 0) sh[t >> 6] = m;
    __syncthreads();
    if (t == 0) amax[b * KS_NCH + chunk] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

// Per-row terms of a key-set pass: bias[j] = fp32(-nrm_j sx sy / 2) and its per-pair maximum (atomicMax on float bits, zeroed by
// the caller); with `mass`: its fp32 rounding scale32 (key B's per-source factor -- the rounding is part of the key's error
// bound, dm_simnn_core) and the maximum of the ROUNDED values.  One workgroup per 256 rows; run as extra workgroups of the
// launch that builds the source rows (ks_build_kernel), where both operands' norms are known.
struct fs_bias_set {
    const double* nrm; int N, Npad; const double* mass; float* bias; unsigned int* bmax; unsigned int* mmax; float* scale32;
    int rows_out;                                            // entries per pair of bias / scale32 (>= N: padded to whole tiles)
    const double* hint; int32_t* force;                      // (nullable) (B) the pair maxima the target rows were scaled with: a pair whose
                                                             // measured maxima give another scale is re-evaluated exactly as a whole (force[b] = 1)
    double* pair_out;                                        // (nullable) (B) the measured pair maximum: the next call's hint
};
__device__ __forceinline__ void fs_bias_body(const fs_bias_set& s, int b, int chunk, const double* __restrict__ amaxT, int nT,
                                             const double* __restrict__ amaxS, int nS) {
    __shared__ float wb[4], wm[4];
    const int j = chunk * 256 + threadIdx.x;
    if (chunk * 256 >= s.N) return;                          // uniform
    const double sx_ = ks_scale(amaxT + b * nT, nT);
    const double sxy = sx_ * ks_scale(amaxS + b * nS, nS);
    if (chunk == 0 && threadIdx.x == 0) {
        if (s.hint && ks_scale(s.hint + b, 1) != sx_) s.force[b] = 1;
        if (s.pair_out) { double m_ = 0.0; for (int q = 0; q < nT; ++q) m_ = fmax(m_, amaxT[b * nT + q]); s.pair_out[b] = m_; }
    }
    float bb = 0.f, mm = 0.f;
    if (j < s.N) {
        const float v = (float)(-0.5 * s.nrm[(long long)b * s.Npad + j] * sxy);
        s.bias[(long long)b * s.rows_out + j] = v;
        bb = fabsf(v);
        if (s.mass) {
            const float m32 = (float)s.mass[(long long)b * s.N + j];
            s.scale32[(long long)b * s.rows_out + j] = m32;
            mm = fabsf(m32);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { bb = fmaxf(bb, __shfl_xor(bb, off)); mm = fmaxf(mm, __shfl_xor(mm, off)); }
    if ((threadIdx.x & 63) == 0) { wb[threadIdx.x >> 6] = bb; wm[threadIdx.x >> 6] = mm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bb = fmaxf(fmaxf(wb[0], wb[1]), fmaxf(wb[2], wb[3]));
        mm = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        atomicMax(s.bmax + b, __float_as_uint(bb));
        if (s.mass) atomicMax(s.mmax + b, __float_as_uint(mm));
    }
}

// Feature rows, layout [ head: 8 bias slots (3 used) or none | 3 entries per contraction index | zero pad up to fill ],
// row stride ld.  A workgroup builds 64 complete rows in LDS -- the K-major float64 operand is read coalesced over the
// vertices (thread = vertex x group of 16 contraction indices) -- and writes them out as one contiguous run of 16-byte
// chunks.  SRC: source rows (h, l, h) with the bias -n1 sx sy / 2 in the head; else target rows (h, h, l) with ones.
// The target is built once for the largest depth: a search at depth K' < K reads only head + 3 K' (+ pad) entries of it,
// and the source rows are zero beyond head + 3 K'.
template <bool SRC>
__global__ __launch_bounds__(256) void ks_build_kernel(const double* __restrict__ M, const double* __restrict__ n1,
                                                       const double* __restrict__ amaxT, const double* __restrict__ amaxS, int nS,
                                                       int K, int N, int Npad, int Kpad, int ld, int fill, int head,
                                                       _Float16* __restrict__ F, int32_t* __restrict__ overflow, int paired, int nT,
                                                       int rows_out,          // rows per pair of F (>= N: padded outputs)
                                                       int nbuild, fs_bias_set sA, fs_bias_set sB, int nbb) {
    // the first 2 nbb workgroups of a pair (dm_launch_fm_split) write the per-row terms of the pass instead (first: they are
    // short and would otherwise be the tail of the launch)
    if ((int)blockIdx.x < 2 * nbb) {
        const int e = (int)blockIdx.x;
        fs_bias_body(e < nbb ? sA : sB, blockIdx.y, e < nbb ? e : e - nbb, amaxT, nT, amaxS, nS);
        return;
    }
    (void)nbuild;
    extern __shared__ __attribute__((aligned(16))) _Float16 ks_img[];       // 64 rows x (fill + 8) halves
    const int ldl = fill + 8;
    const int b = blockIdx.y, v0 = ((int)blockIdx.x - 2 * nbb) * 64;
    const int vl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int v = v0 + vl;
    const double sx = ks_scale(amaxT + b * nT, nT);
    const double sc = SRC ? ks_scale(amaxS + b * nS, nS) : sx;
    _Float16* row = ks_img + vl * ldl;
    if (v < N && paired) {
        // split rows of the key-set tile kernels (dm_simnn.hip, SPLIT): per 16 contraction indices [16 high | 16 low], targets and
        // sources alike (head = 0, fill = 32 ceil(K / 16)); the kernel forms hx.hy + hx.ly + lx.hy from them
        for (int r0 = rg * 16; r0 < K; r0 += 64) {
            const double* col = M + ((long long)b * Kpad + r0) * Npad + v;
            _Float16* dst = row + 2 * r0;
            f16x8 o[4];
            if (r0 + 16 <= K) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    _Float16 h, l;
                    split2_hw(col[(long long)q * Npad] * sc, h, l);
                    o[q >> 3][q & 7] = h;
                    o[2 + (q >> 3)][q & 7] = l;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    _Float16 h = (_Float16)0.0f, l = (_Float16)0.0f;
                    if (r0 + q < K) split2_hw(col[(long long)q * Npad] * sc, h, l);
                    o[q >> 3][q & 7] = h;
                    o[2 + (q >> 3)][q & 7] = l;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f16x8*>(dst + 8 * q) = o[q];
        }
    } else if (v < N) {
        for (int r0 = rg * 16; r0 < K; r0 += 64) {
            const double* col = M + ((long long)b * Kpad + r0) * Npad + v;
            _Float16* dst = row + head + 3 * r0;
            if (r0 + 16 <= K) {
                f16x8 o[6];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    _Float16 h, l;
                    split2_hw(col[(long long)q * Npad] * sc, h, l);
                    const int e = 3 * q;                                  // target: (h, h, l)   source: (h, l, h)
                    o[e >> 3][e & 7] = h;
                    o[(e + 1) >> 3][(e + 1) & 7] = SRC ? l : h;
                    o[(e + 2) >> 3][(e + 2) & 7] = SRC ? h : l;
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) *reinterpret_cast<f16x8*>(dst + 8 * q) = o[q];
            } else {
                for (int q = 0; r0 + q < K; ++q) {
                    _Float16 h, l;
                    split2_hw(col[(long long)q * Npad] * sc, h, l);
                    dst[3 * q] = h; dst[3 * q + 1] = SRC ? l : h; dst[3 * q + 2] = SRC ? h : l;
                }
            }
        }
        for (int c = head + 3 * K + rg; c < fill; c += 4) row[c] = (_Float16)0.0f;
        if (rg == 0 && head > 0) {
            f16x8 head8 = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
            if (SRC) {
                const double beta = -0.5 * n1[(long long)b * Npad + v] * sx * sc;
                if (!(fabs(beta) < 60000.0)) overflow[b] = 1;
                else {
                    const _Float16 b0 = (_Float16)beta;
                    const double r1 = beta - (double)b0;
                    const _Float16 b1 = (_Float16)r1;
                    head8[0] = b0; head8[1] = b1; head8[2] = (_Float16)(r1 - (double)b1);
                }
            } else {
                head8[0] = head8[1] = head8[2] = (_Float16)1.0f;
            }
            *reinterpret_cast<f16x8*>(row) = head8;
        }
    }
    __syncthreads();
    const int cpr = fill >> 3;                                          // 16-byte chunks per row (fill % 8 == 0)
    const int nrow = min(64, N - v0);
    for (int c = threadIdx.x; c < nrow * cpr; c += 256) {
        const int r = c / cpr, q = c - r * cpr;
        *reinterpret_cast<f16x8*>(F + ((long long)b * rows_out + v0 + r) * ld + 8 * q) = *reinterpret_cast<const f16x8*>(ks_img + r * ldl + 8 * q);
    }
}

// exact re-evaluation: one workgroup per queued row (dm_exact.h: ks_exact_row).  One launch serves up to four reductions of a pass
// (gridDim.y = their number, each with its value kind); the workgroups stride over the CONCATENATION of the queues -- their
// lengths differ by an order of magnitude (config 2: 0.07 - 0.9 % of the rows), and with one slice of the grid per queue the
// longest one ran three rows per workgroup one after the other while the other slices idled.
template <int K0, int K1, int K2 = 0, int K3 = 0, typename TR = double>
__global__ __launch_bounds__(256) void ks_exact_kernel(ks_exact_args a0, ks_exact_args a1, ks_exact_args a2, ks_exact_args a3) {
    extern __shared__ double xrow[];                 // K doubles + 8 x 32 partial sums + 32 scores
    __shared__ unsigned long long cmask[4];
    const int nq = gridDim.y;
    const int c0 = *a0.q.flag_count, c1 = nq > 1 ? *a1.q.flag_count : 0, c2 = nq > 2 ? *a2.q.flag_count : 0, c3 = nq > 3 ? *a3.q.flag_count : 0;
    const int total = c0 + c1 + c2 + c3, nwg = gridDim.x * gridDim.y;
    for (int g = blockIdx.x + gridDim.x * blockIdx.y; g < total; g += nwg) {       // (uniform per workgroup)
        if (g < c0) ks_exact_row<K0, TR, TR>(a0, a0.q.flag_list[g], a0.q.flag_thr[g], xrow, cmask);
        else if (g < c0 + c1) ks_exact_row<K1, TR, TR>(a1, a1.q.flag_list[g - c0], a1.q.flag_thr[g - c0], xrow, cmask);
        else if (g < c0 + c1 + c2) ks_exact_row<K2, TR, TR>(a2, a2.q.flag_list[g - c0 - c1], a2.q.flag_thr[g - c0 - c1], xrow, cmask);
        else ks_exact_row<K3, TR, TR>(a3, a3.q.flag_list[g - c0 - c1 - c2], a3.q.flag_thr[g - c0 - c1 - c2], xrow, cmask);
    }
}

static inline size_t ks_build_lds(int fill) { return (size_t)64 * (fill + 8) * sizeof(_Float16); }
static inline int ks_depth(int K) { return pad_to(KS_BIAS + 3 * K, 32) < 96 ? 96 : pad_to(KS_BIAS + 3 * K, 32); }

size_t dm_knn_split_prep_bytes(int B, int N2, int kf) {
    return dm_align_up((size_t)B * N2 * ks_depth(kf) * 2) + dm_align_up((size_t)B * KS_NCH * 8) + 4096;
}
size_t dm_knn_split_ws_bytes(int B, int N2, int N1, int kf) {
    return dm_align_up((size_t)B * N1 * ks_depth(kf) * 2) + dm_align_up((size_t)B * 4) + dm_simnn_ws_bytes(B, N2, N1) + 8192;
}

int dm_knn_split_prepare(dm_ctx* ctx, int B, int N2, int N2pad, int Kpad, int kf, const double* AT, dm_knn_split_state* st) {
    st->enabled = ctx->opt_knn_split != 0;
    if (!st->enabled) return DM_OK;
    st->kf = kf; st->ldT = ks_depth(kf);
    st->Ft = (_Float16*)dm_ws_take(ctx, (size_t)B * N2 * st->ldT * 2);
    st->amaxT = (double*)dm_ws_take(ctx, (size_t)B * KS_NCH * 8);
    if (!st->Ft || !st->amaxT) return dm_fail(ctx, DM_ENOMEM, "knn_split: workspace not reserved");
    DM_LAUNCH(ctx, "knn_split_absmax", ks_absmax_kernel, dim3(KS_NCH, B), dim3(256), 0, AT, kf, N2, N2pad, Kpad, st->amaxT);
    int rcb = dm_grant_lds(ctx, (const void*)ks_build_kernel<false>, ks_build_lds(st->ldT));
    if (rcb) return rcb;
    DM_LAUNCH(ctx, "knn_split_build", ks_build_kernel<false>, dim3(dm_cdiv(N2, 64), B), dim3(256), ks_build_lds(st->ldT), AT,
              (const double*)nullptr, st->amaxT, (const double*)nullptr, 0, kf, N2, N2pad, Kpad, st->ldT, st->ldT, KS_BIAS, st->Ft,
              (int32_t*)nullptr, 0, KS_NCH, N2, dm_cdiv(N2, 64), fs_bias_set{}, fs_bias_set{}, 0);
    return DM_OK;
}

int dm_launch_knn21(dm_ctx* ctx, const dm_gred_args& a, const dm_knn_split_state& st, const double* amaxS, int nS) {
    if (!st.enabled) return dm_launch_gred(ctx, a);
    if (!a.AT || !a.BT || !a.n1 || !a.knn21 || !amaxS) return dm_fail(ctx, DM_EINVAL, "knn_split: missing operand");
    const int K = a.Ktrue > 0 ? a.Ktrue : a.Kloop;
    if (K > st.kf) return dm_fail(ctx, DM_EINVAL, "knn_split: depth %d beyond the prepared %d", K, st.kf);
    const int D = ks_depth(K);
    _Float16* Fs = (_Float16*)dm_ws_take(ctx, (size_t)a.B * a.N1 * D * 2);
    int32_t* overflow = (int32_t*)dm_ws_take(ctx, (size_t)a.B * 4);
    if (!Fs || !overflow) return dm_fail(ctx, DM_ENOMEM, "knn_split: workspace not reserved");
    DM_CHECK_HIP(ctx, hipMemsetAsync(overflow, 0, (size_t)a.B * 4, ctx->stream));
    int rcb = dm_grant_lds(ctx, (const void*)ks_build_kernel<true>, ks_build_lds(D));
    if (rcb) return rcb;
    DM_LAUNCH(ctx, "knn_split_build", ks_build_kernel<true>, dim3(dm_cdiv(a.N1, 64), a.B), dim3(256), ks_build_lds(D), a.BT,
              a.n1, st.amaxT, amaxS, nS, K, a.N1, a.N1pad, a.Kpad, D, D, KS_BIAS, Fs, overflow, 0, KS_NCH, a.N1, dm_cdiv(a.N1, 64), fs_bias_set{}, fs_bias_set{}, 0);
    dm_simnn_queue q;
    // error of the split on top of the fp32 accumulation, relative to |t_i| max_j |s_j|: the dropped <xl, yl> and the two
    // residuals (3 * 2^-22), the fp16 subnormal floor (2 sqrt(K) 2^-25), 25 % slack; 2^-19 at K = 200
    const float rel_extra = 1.25f * (3.0f * 2.3841858e-7f + 2.0f * sqrtf((float)K) * 2.9802322e-8f);
    int rc = dm_simnn_core(ctx, a.B, a.N2, a.N1, D, st.Ft, st.ldT, Fs, D, rel_extra, overflow, a.knn21, nullptr, nullptr, &q);
    if (rc) return rc;
    const size_t lds = ((size_t)K + 8 * 32 + 32) * sizeof(double);
    ks_exact_args ea{a.AT, a.BT, a.n1, nullptr, nullptr, K, a.N2, a.N2pad, a.N1, a.N1pad, a.Kpad, q, a.knn21};
    DM_LAUNCH(ctx, "knn_split_exact_f64", (ks_exact_kernel<0, 0>), dim3(2048, 1), dim3(256), lds, ea, ea, ea, ea);
    return DM_OK;
}

// ---- all four maps of dm_fm_to_p2p on the fp16 matrix cores ---------------------------------------------------------
// The operands X = Phi2 rows and Y = emb1 rows are scaled by powers of two (sx, sy) and stored as SPLIT rows: per 16
// contraction indices [16 high fp16 halves | 16 low halves]; the tile kernel forms hx.hy + hx.ly + lx.hy from them (three
// k-steps per 32-halves stage), i.e. sx sy <x_i, y_j> up to the split error.  ONE pass, targets X, sources Y, reduces every
// tile in both directions (p2p_split >= 2; p2p_split = 1: two passes of the two-key kernel with the operands swapped):
//   along the sources j:  key A  = s - n1_j sx sy / 2 -> knn21,   key B  = s a1_j -> ind21
//   along the targets i:  key A' = s - n2_i sx sy / 2 -> knn12,   key B' = s      -> ind12   (a1_j > 0 is a per-column factor;
//                                                  columns with a1_j == 0 score 0 everywhere: index 0, like np.argmax)
// then the ambiguous rows of each of the four reductions are re-evaluated exactly (ks_exact_kernel) with the reference's
// own float64 expressions -- emb1 from its K-major float64 copy, Phi2 from the caller's row-major array.
// Any N1, N2 >= 256 (operands padded to whole tiles, edge tiles masked) and K >= 65 (five stages of 16 indices:
// dm_fm_split_ok); otherwise the float64 G kernel.
// (mass, when given: also its fp32 rounding scale32, the per-source factor of the tile kernel's key B -- the rounding is
//  part of the key's error bound, dm_simnn_core -- and the maximum of the ROUNDED values)
// ind12[j] = 0 where the target's mass is zero (the whole indicator column is 0: np.argmax returns the first index)
__global__ __launch_bounds__(256) void fs_zero_mass_kernel(const double* __restrict__ mass, long long n, int32_t* __restrict__ ind12) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o < n && mass[o] == 0.0) ind12[o] = 0;
}

// X feature rows straight from the basis as it is in memory (row-major, fp32 or fp64: no transpose), in the split-row layout
// of the key-set tile kernels ([16 high | 16 low] per 16 contraction indices): thread (vertex, group of 8 indices) reads
// 32 / 64 contiguous bytes and writes two 16-byte chunks; same values as ks_build_kernel (paired) on the float64 copy of
// the same numbers
template <typename TR>
__global__ __launch_bounds__(256) void fs_build_rows_kernel(const TR* __restrict__ Phi, int N, int K, int ld, const double* __restrict__ amaxT,
                                                            int nT, int D, _Float16* __restrict__ F, int rows_out) {
    const int b = blockIdx.y;
    const int ngrp = D / 16;                                    // groups of 8 indices (D = 32 ceil(K / 16))
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= (long long)N * ngrp) return;
    const int v = (int)(o / ngrp), q = (int)(o - (long long)v * ngrp);
    // (fp32 basis: x sx, |.| < 2, is exact in fp32 -- sx is a power of two --, and so is x sx - h: the pieces equal those of
    //  the float64 split; fp64 basis: the split itself runs in float64, split2)
    const double sx = ks_scale(amaxT + b * nT, nT);
    const TR* src = Phi + ((long long)b * N + v) * ld + 8 * q;
    _Float16* dst = F + ((long long)b * rows_out + v) * D + 32 * (q >> 1) + 8 * (q & 1);
    TR xin[8];
    constexpr int amask = sizeof(TR) == 4 ? 3 : 1;
    if (8 * q + 8 <= K && ((ld & amask) == 0) && ((((uintptr_t)Phi) & 15) == 0)) {   // 16-byte loads
        if constexpr (sizeof(TR) == 4) {
            typedef __attribute__((address_space(1))) const f32x4 gf32x4;
            const f32x4 v0 = ((gf32x4*)src)[0], v1 = ((gf32x4*)src)[1];
#pragma unroll
            for (int u = 0; u < 4; ++u) { xin[u] = v0[u]; xin[4 + u] = v1[u]; }
        } else {
            typedef __attribute__((address_space(1))) const f64x2 gf64x2;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const f64x2 w = ((gf64x2*)src)[u]; xin[2 * u] = w[0]; xin[2 * u + 1] = w[1]; }
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) xin[u] = (8 * q + u < K) ? src[u] : (TR)0;
    }
    f16x8 hv, lv;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        _Float16 h, l;
        fs_split_entry<TR>(xin[u], sx, h, l);                         // exact scaling (sx is a power of two); 0 beyond K
        hv[u] = h; lv[u] = l;
    }
    *reinterpret_cast<f16x8*>(dst) = hv;
    *reinterpret_cast<f16x8*>(dst + 16) = lv;
}

// split rows of a row-major basis for the key-set tile kernels: D halves per row (>= 32 ceil(K / 16), zero beyond K), rows_out rows
// per pair in F (the caller zeroes rows >= N)
template <typename TR>
int dm_fm_split_build_rows(dm_ctx* ctx, int B, int N, int K, const TR* Phi, int ld, const double* amaxT, int nT, int D, _Float16* F, int rows_out) {
    const long long n = (long long)N * (D / 16);
    DM_LAUNCH(ctx, "fm_split_build_rows", fs_build_rows_kernel<TR>, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, Phi, N, K, ld,
              amaxT, nT, D, F, rows_out);
    return DM_OK;
}
template int dm_fm_split_build_rows<float>(dm_ctx*, int, int, int, const float*, int, const double*, int, int, _Float16*, int);
template int dm_fm_split_build_rows<double>(dm_ctx*, int, int, int, const double*, int, const double*, int, int, _Float16*, int);

static inline int fs_depth(int K) { return 32 * ((K + 15) / 16); }    // halves per split row
size_t dm_fm_split_zero_bytes(int B) { return 4 * dm_align_up((size_t)B * 4) + dm_align_up(dm_simnn_ctl_bytes(B)); }   // max |bias A|, max mass, max |bias B|, the pairs' force flags, the tile pass's control block (its own memset saved)
bool dm_fm_split_ok(const dm_ctx* ctx, int N2, int N1, int K) {
    return ctx->opt_p2p_split != 0 && dm_simnn_dual_ok(ctx, N2, N1, fs_depth(K), true) && N1 >= 256 && N2 >= 256;
}
size_t dm_fm_split_ws_bytes(int B, int N2, int N1, int K) {
    const size_t D = fs_depth(K), R2 = pad_to(N2, 256), R1 = pad_to(N1, 256);
    return dm_align_up((size_t)B * R2 * D * 2) + dm_align_up((size_t)B * R1 * D * 2) +
           2 * dm_align_up((size_t)B * R1 * 4) + dm_align_up((size_t)B * R2 * 4) +
           dm_simnn_ws_bytes(B, N2, N1, 3) + dm_simnn_ws_bytes(B, N1, N2, 1) + 16384;
}
// a: AT, BT (K-major f64), n1, n2, mass1 and all four outputs; amaxS: per-256-column maxima of |BT| (colnorm_kernel);
// zeroed: dm_fm_split_zero_bytes block, zeroed before dm_launch_phiT(Phi2) filled its first part
// the target rows' buffer: the first piece dm_launch_fm_split takes from the arena (the caller takes it early when the second
// embedding is to write the rows); padding rows are zeroed here
_Float16* dm_fm_split_take_fx(dm_ctx* ctx, int B, int N2, int K, int* D_out, int* rows_out) {
    const int D = fs_depth(K), R2 = pad_to(N2, 256);
    _Float16* Fx = (_Float16*)dm_ws_take(ctx, (size_t)B * R2 * D * 2);
    if (!Fx) return nullptr;
    if (R2 != N2 && hipMemsetAsync(Fx, 0, (size_t)B * R2 * D * 2, ctx->stream) != hipSuccess) return nullptr;
    *D_out = D; *rows_out = R2;
    return Fx;
}

template <typename TR>
int dm_launch_fm_split(dm_ctx* ctx, const dm_gred_args& a, const double* amaxS, int nS, const double* amaxT, int nT, void* zeroed,
                       const TR* Phi2, int ld2, const dm_fm_split_pre* pre) {
    const int B = a.B, N1 = a.N1, N2 = a.N2;
    const int K = a.Ktrue > 0 ? a.Ktrue : a.Kloop;
    if (!a.BT || !a.n1 || !a.n2 || !a.mass1 || !a.knn21 || !a.knn12 || !a.ind21 || !a.ind12 || !amaxS || !amaxT || !zeroed || !Phi2)
        return dm_fail(ctx, DM_EINVAL, "fm_split: missing operand");
    int D = fs_depth(K);
    // Any N2, N1: the split rows and the per-row terms are padded to whole 256-tiles (zero rows; the tiles that reach into the
    // padding mask it in their reductions), so a mesh with 2000 vertices takes the same path as one with 2048.
    int R2 = pad_to(N2, 256);
    const int R1 = pad_to(N1, 256);
    _Float16* Fx = pre ? pre->Fx : dm_fm_split_take_fx(ctx, B, N2, K, &D, &R2);
    _Float16* Fy = (_Float16*)dm_ws_take(ctx, (size_t)B * R1 * D * 2);
    float* biasA = (float*)dm_ws_take(ctx, (size_t)B * R1 * 4);
    float* biasB = (float*)dm_ws_take(ctx, (size_t)B * R2 * 4);
    float* scale32 = (float*)dm_ws_take(ctx, (size_t)B * R1 * 4);       // fp32 rounding of mass1: key B of the tile kernel
    if (!Fx || !Fy || !biasA || !biasB || !scale32) return dm_fail(ctx, DM_ENOMEM, "fm_split: workspace not reserved");
    if (R1 != N1) DM_CHECK_HIP(ctx, hipMemsetAsync(Fy, 0, (size_t)B * R1 * D * 2, ctx->stream));
    if (R1 != N1) {                                           // (padded terms reach LDS and masked lanes: keep them finite)
        DM_CHECK_HIP(ctx, hipMemsetAsync(biasA, 0, (size_t)B * R1 * 4, ctx->stream));
        DM_CHECK_HIP(ctx, hipMemsetAsync(scale32, 0, (size_t)B * R1 * 4, ctx->stream));
    }
    if (R2 != N2) DM_CHECK_HIP(ctx, hipMemsetAsync(biasB, 0, (size_t)B * R2 * 4, ctx->stream));
    const size_t mstride = dm_align_up((size_t)B * 4) / 4;
    unsigned int* bmaxA = reinterpret_cast<unsigned int*>(zeroed);
    unsigned int* mmax = bmaxA + mstride; unsigned int* bmaxB = bmaxA + 2 * mstride;
    // pairs whose target rows were written with a hinted scale that is not their data's: every row takes the exact path
    int32_t* force = (pre && pre->built) ? reinterpret_cast<int32_t*>(bmaxA + 3 * mstride) : nullptr;
    if (!(pre && pre->built)) {
        const long long n = (long long)N2 * (D / 16);
        DM_LAUNCH(ctx, "fm_split_build_rows", fs_build_rows_kernel<TR>, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, Phi2, N2, K, ld2,
                  amaxT, nT, D, Fx, R2);
    }
    int rcb = dm_grant_lds(ctx, (const void*)ks_build_kernel<true>, ks_build_lds(D));
    if (rcb) return rcb;
    {
        // (the same launch writes the per-row terms of the pass: 2 x ceil(N / 256) extra workgroups per pair)
        const fs_bias_set sA{a.n1, N1, a.N1pad, a.mass1, biasA, bmaxA, mmax, scale32, R1, force ? pre->hint : nullptr, force, pre ? pre->pair_out : nullptr};
        const fs_bias_set sB{a.n2, N2, a.N2pad, nullptr, biasB, bmaxB, nullptr, nullptr, R2, nullptr, nullptr, nullptr};
        const int nbuild = dm_cdiv(N1, 64), nbb = dm_cdiv(N1 > N2 ? N1 : N2, 256);
        DM_LAUNCH(ctx, "knn_split_build", ks_build_kernel<true>, dim3(nbuild + 2 * nbb, B), dim3(256), ks_build_lds(D), a.BT,
                  a.n1, amaxT, amaxS, nS, K, N1, a.N1pad, a.Kpad, D, D, 0, Fy, (int32_t*)nullptr, 1, nT, R1, nbuild, sA, sB, nbb);
    }
    const float rel_extra = 1.25f * (3.0f * 2.3841858e-7f + 2.0f * sqrtf((float)K) * 2.9802322e-8f);
    const size_t lds = ((size_t)K + 8 * 32 + 32) * sizeof(double);
    if (ctx->opt_p2p_split >= 2) {
        // one pass, both directions: every tile reduces along its source rows (knn21, ind21) and, transposed through LDS,
        // along its target rows (knn12, ind12)
        dm_simnn_queue qa, qb, qc, qd;
        // (ind12[j] = 0 where mass1[j] = 0: the whole indicator column is 0 and np.argmax returns the first index)
        dm_simnn_cols cols{biasB, reinterpret_cast<const float*>(bmaxB), a.knn12, a.ind12, &qc, &qd, a.mass1};
        dm_simnn_dual dual{biasA, scale32, reinterpret_cast<const float*>(bmaxA), reinterpret_cast<const float*>(mmax), a.ind21, &qb, &cols, true};
        dm_simnn_ext ext;                                   // (the pass's per-pair maxima and queue counters live in the caller's zeroed block)
        ext.ctl = reinterpret_cast<char*>(zeroed) + 4 * dm_align_up((size_t)B * 4);
        int rc = dm_simnn_core(ctx, B, N2, N1, D, Fx, D, Fy, D, rel_extra, force, a.knn21, nullptr, nullptr, &qa, &dual, &ext);
        if (rc) return rc;
        // (Phi2 is read where the caller keeps it: targets of e0 / e1, candidates of f0 / f1)
        ks_exact_args e0{nullptr, a.BT, a.n1, nullptr, nullptr, K, N2, a.N2pad, N1, a.N1pad, a.Kpad, qa, a.knn21, Phi2, nullptr, ld2};
        ks_exact_args e1 = e0;
        e1.n1 = nullptr; e1.massS = a.mass1; e1.q = qb; e1.nn = a.ind21;
        ks_exact_args f0{a.BT, nullptr, a.n2, nullptr, nullptr, K, N1, a.N1pad, N2, a.N2pad, a.Kpad, qc, a.knn12, nullptr, Phi2, ld2};
        ks_exact_args f1 = f0;
        f1.n1 = nullptr; f1.massT = a.mass1; f1.q = qd; f1.nn = a.ind12;
        DM_LAUNCH(ctx, "fm_split_exact_f64", (ks_exact_kernel<0, 1, 0, 2, TR>), dim3(512, 4), dim3(256), lds, e0, e1, f0, f1);
        return DM_OK;
    }
    // pass A: targets = Phi2 rows, candidates = emb1 rows
    {
        dm_simnn_queue qa, qb;
        dm_simnn_dual dual{biasA, scale32, reinterpret_cast<const float*>(bmaxA), reinterpret_cast<const float*>(mmax), a.ind21, &qb, nullptr, true};
        int rc = dm_simnn_core(ctx, B, N2, N1, D, Fx, D, Fy, D, rel_extra, force, a.knn21, nullptr, nullptr, &qa, &dual);
        if (rc) return rc;
        ks_exact_args e0{nullptr, a.BT, a.n1, nullptr, nullptr, K, N2, a.N2pad, N1, a.N1pad, a.Kpad, qa, a.knn21, Phi2, nullptr, ld2};
        ks_exact_args e1 = e0;
        e1.n1 = nullptr; e1.massS = a.mass1; e1.q = qb; e1.nn = a.ind21;
        DM_LAUNCH(ctx, "fm_split_exact_f64", (ks_exact_kernel<0, 1, 0, 0, TR>), dim3(1024, 2), dim3(256), lds, e0, e1, e0, e1);
    }
    // pass B: targets = emb1 rows, candidates = Phi2 rows
    {
        dm_simnn_queue qa, qb;
        dm_simnn_dual dual{biasB, nullptr, reinterpret_cast<const float*>(bmaxB), nullptr, a.ind12, &qb, nullptr, true};
        int rc = dm_simnn_core(ctx, B, N1, N2, D, Fy, D, Fx, D, rel_extra, force, a.knn12, nullptr, nullptr, &qa, &dual);
        if (rc) return rc;
        ks_exact_args e0{a.BT, nullptr, a.n2, nullptr, nullptr, K, N1, a.N1pad, N2, a.N2pad, a.Kpad, qa, a.knn12, nullptr, Phi2, ld2};
        ks_exact_args e1 = e0;
        e1.n1 = nullptr; e1.massT = a.mass1; e1.q = qb; e1.nn = a.ind12;
        DM_LAUNCH(ctx, "fm_split_exact_f64", (ks_exact_kernel<0, 2, 0, 0, TR>), dim3(1024, 2), dim3(256), lds, e0, e1, e0, e1);
        const long long n = (long long)B * N1;
        DM_LAUNCH(ctx, "fm_split_zero_mass", fs_zero_mass_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, a.mass1, n, a.ind12);
    }
    return DM_OK;
}
template int dm_launch_fm_split<float>(dm_ctx*, const dm_gred_args&, const double*, int, const double*, int, void*, const float*, int, const dm_fm_split_pre*);
template int dm_launch_fm_split<double>(dm_ctx*, const dm_gred_args&, const double*, int, const double*, int, void*, const double*, int, const dm_fm_split_pre*);

// Exact float64 re-evaluation of one queued row of a split nearest-neighbour / indicator pass (dm_knnsplit.hip,
// dm_zoomfuse.hip).
#pragma once
#include "dm_device.h"
#include "dm_internal.h"

// exact re-evaluation: one workgroup (256 threads) per queued row.  Thread (part = t >> 5, c = t & 31) accumulates the contraction
// rows r = part, part + 8, ... of candidate j = 32 block + c (each wave instruction reads two 256-byte runs of BT);
// the eight partial sums are added in a fixed order, so duplicated columns score identically and the lowest index wins
// -- and g(i, j) is the same number whichever operand plays the target (the four maps of dm_fm_to_p2p agree on it).
// KIND (the value the reference compares, targets = columns of AT, candidates = columns of BT):
//   0  arg-min_j  w[j] - 2 g        w = |candidate|^2                 knn21 (and knn12 with the operands swapped)
//   1  arg-max_j  g massS[j]        indicator row,    convert.py:144  ind21
//   2  arg-max_j  g massT[i]        indicator column (the target's own mass)  ind12, operands swapped
struct ks_exact_args {
    const double* AT; const double* BT; const double* n1; const double* massS; const double* massT;
    int K, N2, N2pad, N1, N1pad, Kpad;
    dm_simnn_queue q;                                  // flagged rows + the partials that prune their candidates
    int32_t* nn;
    // one operand may be read where the caller keeps it, row-major (B, N, ldrow) of the kernel's TR, instead of from a K-major
    // float64 copy: Trow replaces AT (targets), Crow replaces BT (candidates; row stride ldcrow, 0 = ldrow).  Same values, same
    // summation schedule.
    const void* Trow = nullptr; const void* Crow = nullptr; int ldrow = 0; int ldcrow = 0;
};

// one queued row: entry o = b N2 + i of the list, threshold thr.  xrow: K doubles + 8 x 32 partial sums of dynamic LDS, cmask:
// four words of LDS; all 256 threads of the workgroup call it (it synchronises).  TR: type of Trow, TC: type of Crow.
template <int KIND, typename TR, typename TC>
__device__ __forceinline__ void ks_exact_row(const ks_exact_args& a, int o, float thr, double* xrow, unsigned long long* cmask) {
    const double* __restrict__ AT = a.AT; const double* __restrict__ BT = a.BT; const double* __restrict__ n1 = a.n1;
    const double* __restrict__ massS = a.massS; const double* __restrict__ massT = a.massT;
    const int K = a.K, N2 = a.N2, N2pad = a.N2pad, N1 = a.N1, N1pad = a.N1pad, Kpad = a.Kpad;
    const float* __restrict__ qpb = a.q.pb; const int32_t* __restrict__ qpj = a.q.pj; const float* __restrict__ qps = a.q.ps;
    const int nparts = a.q.nparts, pw = a.q.pw, Npad_s = a.q.Npad, nsub = nparts * (pw / 32);
    int32_t* __restrict__ nn = a.nn;
    double* part_s = xrow + K;
    const TR* __restrict__ Trow = reinterpret_cast<const TR*>(a.Trow);
    const TC* __restrict__ Crow = reinterpret_cast<const TC*>(a.Crow);
    const int ldrow = a.ldrow, ldcrow = a.ldcrow ? a.ldcrow : a.ldrow;
    // thread = (candidate c of a block of 32, part of the contraction).  Row-major candidates: the eight parts of a candidate
    // sit in neighbouring lanes, so one load instruction reads eight 64-byte runs instead of 64 scattered elements.
    const int t = threadIdx.x;
    const int c = Crow ? (t >> 3) : (t & 31), part = Crow ? (t & 7) : (t >> 5);
    {
        const int b = o / N2, i = o - b * N2;
        const double* A = AT ? AT + (long long)b * Kpad * N2pad + i : nullptr;
        const double* Bm = BT ? BT + (long long)b * Kpad * N1pad : nullptr;
        // the block filter of the first 256 blocks (dm_simnn_keep with its three loads side by side: short-circuit evaluation made
        // them three dependent round trips), requested before the target row: one round trip for both
        auto keep_of = [&](int sbt) -> bool {
            const int q = min((sbt * 32) / pw, nparts - 1);
            const long long oq = ((long long)b * nparts + q) * Npad_s + i;
            const float vs = qps[oq], vb = qpb[oq];
            const int vj = qpj[oq];
            return sbt < nsub && (vs >= thr || (vb >= thr && (vj >> 5) == sbt));
        };
        bool keep = keep_of(t);
        __syncthreads();
        if (Trow) { for (int r = t; r < K; r += 256) xrow[r] = (double)Trow[((long long)b * N2 + i) * ldrow + r]; }
        else { for (int r = t; r < K; r += 256) xrow[r] = A[(long long)r * N2pad]; }
        __syncthreads();
        double bv = KIND == 0 ? DM_INF_F64 : -DM_INF_F64;
        int bj = DM_IDX_NONE;
        const double mt = KIND == 2 ? massT[(long long)b * N2 + i] : 0.0;
        // candidate blocks: one gather of the row's partials (256 blocks at a time), then only the blocks that can still
        // hold the optimum are visited, in ascending order (dm_simnn_keep)
        for (int sb0 = 0; sb0 < nsub; sb0 += 256) {
            if (sb0) keep = keep_of(sb0 + t);
            const unsigned long long km = __ballot(keep);
            if ((t & 63) == 0) cmask[t >> 6] = km;
            __syncthreads();
            for (int w = 0; w < 4; ++w) {
                unsigned long long mm = cmask[w];                 // uniform
                while (mm) {
                    const int sb = sb0 + w * 64 + __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    const int j = sb * 32 + c;
                    double sacc = 0.0;
                    if (j < N1 && Crow) {
                        const TC* Cr = Crow + ((long long)b * N1 + j) * ldcrow;
                        int r = part;
                        for (; r + 56 < K; r += 64) {
                            TC y[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) y[u] = Cr[r + 8 * u];
#pragma unroll
                            for (int u = 0; u < 8; ++u) sacc = fma(xrow[r + 8 * u], (double)y[u], sacc);
                        }
                        for (; r < K; r += 8) sacc = fma(xrow[r], (double)Cr[r], sacc);
                    } else if (j < N1) {
                        // loads in batches of eight ahead of their (ordered) fma chain: a plain loop would take one L2 round
                        // trip per term
                        int r = part;
                        for (; r + 56 < K; r += 64) {
                            double y[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) y[u] = Bm[(long long)(r + 8 * u) * N1pad + j];
#pragma unroll
                            for (int u = 0; u < 8; ++u) sacc = fma(xrow[r + 8 * u], y[u], sacc);
                        }
                        for (; r < K; r += 8) sacc = fma(xrow[r], Bm[(long long)r * N1pad + j], sacc);
                    }
                    part_s[part * 32 + c] = sacc;
                    __syncthreads();
                    if (t < 32 && sb * 32 + t < N1) {
                        const int j = sb * 32 + t;                      // (thread t < 32 finishes candidate t of the block)
                        double g = part_s[t];
#pragma unroll
                        for (int q = 1; q < 8; ++q) g += part_s[q * 32 + t];
                        // blocks ascend: strict comparisons keep the lowest index
                        if (KIND == 0) {
                            const double v = n1[(long long)b * N1pad + j] - 2.0 * g;          // |y|^2 - 2 <x, y>
                            if (v < bv) { bv = v; bj = j; }
                        } else {
                            const double v = g * (KIND == 1 ? massS[(long long)b * N1 + j] : mt);
                            if (v > bv) { bv = v; bj = j; }
                        }
                    }
                    __syncthreads();
                }
            }
            __syncthreads();                                      // (cmask is rewritten by the next chunk of blocks)
        }
        if (t < 32) {
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const double ov = __shfl_xor(bv, off);
                const int oj = __shfl_xor(bj, off);
                if (KIND == 0) argmin_merge(bv, bj, ov, oj); else argmax_merge(bv, bj, ov, oj);
            }
            if (t == 0 && bj != DM_IDX_NONE) nn[o] = bj;
        }
    }
}

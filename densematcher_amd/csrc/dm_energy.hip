// General functional-map energy and gradient on the device (dm_fmap_energy_grad, dm_fmap_descr_ops):
// what scipy's L-BFGS-B evaluates in FunctionalMapping.fit when terms beyond w_descr / w_lap are switched on.
//
// Reference arithmetic reproduced (oracle/dm_oracle.py: energy_grad_general; pyFM/optimize/base_functions.py:480-763):
//   E(C) = w_descr 1/2 |C A - B|^2 + w_lap 1/2 sum C^2 ev                                    :31-121
//        + w_dcomm sum_d 1/2 |C L_d - R_d C|^2,  L_d = Phi1^T A1 diag(f_d) Phi1, R_d likewise   :124-226, 546-565
//        + terms in the mapped indicator M = Phi2 C Phi1^T A1 (N2 x N1):                       :296-428
//            w_p2p sum (M^2 - M)^2;  w_stochastic [sum_j (sum_i M^2 - n2/n1)^2 + sum_i (sum_j M^2 - 1)^2];
//            w_ent sum -c log(c + 1e-10), c = clamp(M, 0, 1);  w_range01 sum relu(-M)^2 + relu(M - 1)^2;
//            w_sumto1 [sum_j (cs_j - mean cs)^2 + sum_i (rs_i - mean rs)^2]   (v = None branch)
//   gradient: the analytic derivative of the above (what torch autograd returns there), dE/dC = Phi2^T (dE/dM) (A1 Phi1)
//   for the M terms; column 0 is zeroed (:759).
//
// Everything is float64 on the f64 matrix cores.  The mapped indicator is NOT materialised: its tiles are produced twice
// (row / column statistics, then the element-wise derivative contracted with Phi1 on the spot), workspace O(N k).  All
// reductions use fixed orders: the same input gives the same bits.
#include "dm_gemm_f64.h"
#include "dm_internal.h"
#include "dm_energy_dev.h"

static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

enum { W_DESCR = 0, W_LAP, W_DCOMM, W_P2P, W_STOCH, W_ENT, W_RANGE01, W_SUMTO1, W_AREA, W_CONFORMAL, W_COUNT };
struct mterm_weights { double p2p, stoch, ent, range01, sumto1; };

// ---- functors -----------------------------------------------------------------------------------------------------
struct OutNTScaledCols {           // M = (product) * mass1[j]
    double* p; long long stride_b; int ld; const float* mass1; int N1;
    __device__ __forceinline__ void store(int b, int i, int j, double v) const {
        p[b * stride_b + (long long)i * ld + j] = v * (double)mass1[(long long)b * N1 + j];
    }
};
struct OutNTSub {                  // T <- T - product   (in place, one thread per element)
    double* p; long long stride_b; int ld;
    __device__ __forceinline__ void store(int b, int i, int j, double v) const { p[b * stride_b + (long long)i * ld + j] -= v; }
};
// rows of a (B, nrows, ld) float64 matrix shared by `div` consecutive batch slots z (z / div selects the pair);
// `trans` reads element (row, k) at p[k * ld + row]
struct KRowsF64Bcast {
    const double* p; long long stride_b; int ld; int nrows; int ncols; int trans; int div;
    __device__ __forceinline__ void load8(int z, int row, int k0, double (&v)[8]) const {
        const double* base = p + (long long)(z / div) * stride_b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            double x = 0.0;
            if (row < nrows && k < ncols) x = trans ? base[(long long)k * ld + row] : base[(long long)row * ld + k];
            v[e] = x;
        }
    }
};
// row i of the (k2 x nops*k1) matrix [T_0 | T_1 | ...] / row j of [L_0 | L_1 | ...]: element (row, d*kin + k) = X[b][d][row][k]
struct KRowsF64Blocks {
    const double* p; int nops; int nrows; int kin;     // p (B, nops, nrows, kin)
    __device__ __forceinline__ void load8(int b, int row, int k0, double (&v)[8]) const {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = k0 + e;
            double x = 0.0;
            if (row < nrows && kk < nops * kin) {
                const int d = kk / kin, k = kk - d * kin;
                x = p[(((long long)b * nops + d) * nrows + row) * kin + k];
            }
            v[e] = x;
        }
    }
};
struct RowsF64TN {                 // K-major float64 operand for gemm_tn_f64
    const double* p; long long stride_b; int ld; int ncols;
    __device__ __forceinline__ void load4(int b, int n, int col0, double (&v)[4]) const {
        const double* row = p + b * stride_b + (long long)n * ld;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (col0 + e < ncols) ? row[col0 + e] : 0.0;
    }
};
// rows of Phi (fp32) scaled by mass[n] * F[n][d]; batch slot z = b * D + d
struct RowsPhiTimesDescr {
    const float* Phi; long long stride_b; int ld; int ncols;
    const float* mass; int N; const void* F; int f16; int D;
    __device__ __forceinline__ void load4(int z, int n, int col0, double (&v)[4]) const {
        const int b = z / D, d = z - b * D;
        const float* row = Phi + b * stride_b + (long long)n * ld;
        const long long fo = ((long long)b * N + n) * D + d;
        const double f = f16 ? (double)reinterpret_cast<const _Float16*>(F)[fo] : (double)reinterpret_cast<const float*>(F)[fo];
        const double s = (double)mass[(long long)b * N + n] * f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (col0 + e < ncols) ? s * (double)row[col0 + e] : 0.0;
    }
};
struct RowsPhiBatchDiv {           // plain rows of Phi (fp32) for batch slot z = b * D + d
    const float* Phi; long long stride_b; int ld; int ncols; int D;
    __device__ __forceinline__ void load4(int z, int n, int col0, double (&v)[4]) const {
        const float* row = Phi + (long long)(z / D) * stride_b + (long long)n * ld;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (col0 + e < ncols) ? (double)row[col0 + e] : 0.0;
    }
};
struct OutTNPartial {              // split-K partials (nsplit, Z, m, c), or the result itself when nsplit == 1
    double* p; long long Z; int M; int N;
    __device__ __forceinline__ void store(int z, int split, int m, int c, double v) const {
        p[(((long long)split * Z + z) * M + m) * N + c] = v;
    }
};

// out[i] = sum_q partial[q][i]   (fixed order)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ partial, int nsplit, long long n,
                                                              double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int q = 0; q < nsplit; ++q) s += partial[(long long)q * n + i];
    out[i] = s;
}

__global__ __launch_bounds__(256) void quad_terms_kernel(quad_args qa, int k1, int k2, double* __restrict__ grad, double* __restrict__ e_quad) {
    __shared__ double sh[4];
    const double tot = quad_pair(qa, blockIdx.x, threadIdx.x, k1, k2, grad, sh);
    if (threadIdx.x == 0) e_quad[blockIdx.x] = tot;
}

// ---- statistics of the mapped indicator ------------------------------------------------------------------------------
// cs / csq from the chunk partials, and the totals of rs and cs.  Workgroup g of a pair owns columns / rows 256 g .. 256 g + 255 and
// leaves its share of the two totals in stat[(b * G + g) * 2 + {0: rows, 1: columns}]; the consumers add the G shares in order
// (em_means).  (One workgroup per pair took 87 us for a single pair: 2 x 32 x 2048 dependent loads.)
// (rs / rsq arrive as ncs partial sets, one per column split, set q at offset q * B * N2: summed into set 0 here, fixed order)
__global__ __launch_bounds__(256) void m_finish_stats_kernel(const double* __restrict__ pcs, const double* __restrict__ pcsq, int nchunk,
                                                             int N2, int N1, double* __restrict__ rs, double* __restrict__ rsq, int ncs,
                                                             long long split_stride, double* __restrict__ cs,
                                                             double* __restrict__ csq, double* __restrict__ stat) {
    __shared__ double sh[4];
    const int b = blockIdx.y, g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
    double sc = 0.0;
    {
        const int j = g * 256 + t;
        if (j < N1) {
            double s = 0.0, q = 0.0;
            int ch = 0;
            for (; ch + 8 <= nchunk; ch += 8) {          // (eight loads of each array in flight, added in chunk order)
                double a[8], c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { a[u] = pcs[((long long)b * nchunk + ch + u) * N1 + j]; c[u] = pcsq[((long long)b * nchunk + ch + u) * N1 + j]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { s += a[u]; q += c[u]; }
            }
            for (; ch < nchunk; ++ch) { s += pcs[((long long)b * nchunk + ch) * N1 + j]; q += pcsq[((long long)b * nchunk + ch) * N1 + j]; }
            cs[(long long)b * N1 + j] = s; csq[(long long)b * N1 + j] = q;
            sc = s;
        }
    }
    const double tot_c = block_sum_256(sc, sh);
    double sr = 0.0;
    {
        const int i = g * 256 + t;
        if (i < N2) {
            double s = rs[(long long)b * N2 + i], q = rsq[(long long)b * N2 + i];
            for (int c = 1; c < ncs; ++c) { s += rs[c * split_stride + (long long)b * N2 + i]; q += rsq[c * split_stride + (long long)b * N2 + i]; }
            rs[(long long)b * N2 + i] = s; rsq[(long long)b * N2 + i] = q;
            sr = s;
        }
    }
    const double tot_r = block_sum_256(sr, sh);
    if (t == 0) { stat[((long long)b * G + g) * 2] = tot_r; stat[((long long)b * G + g) * 2 + 1] = tot_c; }
}
// Row and column SUMS of the mapped indicator are linear in it: with M_ij = (E2_i . Phi1_j) a1_j and E2 = Phi2 C,
//   rs_i = E2_i . p,  p = sum_j a1_j Phi1_j (k1)        cs_j = a1_j Phi1_j . q,  q = sum_i E2_i = C^T s2,  s2 = sum_i Phi2_i (k2)
// -- O(N k) instead of a pass over the N2 x N1 tiles.  When only w_sumto1 asks for statistics (the notebook's terms; the sums
// of squares of w_stochastic still take the tile pass) two small kernels replace em_stats_kernel + m_finish_stats_kernel:
//   em_basis_sums_kernel    p and s2: they do not depend on C (kept across an optimiser's evaluations, option "energy_keep_gram")
//   em_stats_linear_kernel  a workgroup per 256 rows: q from s2 and C, then rs, cs and the block's shares of their totals
// (fixed summation orders; rsq / csq are zeroed: their terms carry a zero weight).  A single workgroup per pair doing all
// of it took 32 us -- nothing hides its loads -- against 22 + 12 us for the tile pass; this form takes 7.
__global__ __launch_bounds__(1024) void em_basis_sums_kernel(const float* __restrict__ Phi1, int ld1, const float* __restrict__ mass1, int N1, int k1,
                                                             const float* __restrict__ Phi2, int ld2, int N2, int k2, int cw,
                                                             double* __restrict__ p_out, double* __restrict__ s2_out) {
    __shared__ double part[1024];
    const int b = blockIdx.x, t = threadIdx.x, c = t & (cw - 1), rl = t / cw, nrl = 1024 / cw;
    for (int which = 0; which < 2; ++which) {
        const float* P = which ? Phi2 + (long long)b * N2 * ld2 : Phi1 + (long long)b * N1 * ld1;
        const int N = which ? N2 : N1, ld = which ? ld2 : ld1, k = which ? k2 : k1;
        const float* a1 = which ? nullptr : mass1 + (long long)b * N1;
        double s = 0.0;
        if (c < k)
            for (int j = rl; j < N; j += 8 * nrl) {          // (eight loads in flight, added in order)
                float x[8], m8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int jj = j + u * nrl; x[u] = jj < N ? P[(long long)jj * ld + c] : 0.f; m8[u] = (jj < N && a1) ? a1[jj] : 1.f; }
#pragma unroll
                for (int u = 0; u < 8; ++u) s = fma((double)m8[u], (double)x[u], s);
            }
        part[t] = s;
        __syncthreads();
        if (t < cw && t < k) {
            double a = 0.0;
            for (int r = 0; r < nrl; ++r) a += part[r * cw + t];
            (which ? s2_out + (long long)b * k2 : p_out + (long long)b * k1)[t] = a;
        }
        __syncthreads();
    }
}
// Phi2 != null (maps up to 32 x 32): the block's rows of E2 = Phi2 C are produced here (C in the LDS, a row per thread) and
// written for the derivative pass, instead of by a product launch of their own.
__global__ __launch_bounds__(256) void em_stats_linear_kernel(double* __restrict__ E2, const float* __restrict__ Phi1, int ld1,
                                                              const float* __restrict__ mass1, int N1, int N2, int k1, int k2,
                                                              const double* __restrict__ C, const double* __restrict__ p_in,
                                                              const double* __restrict__ s2_in, double* __restrict__ rs, double* __restrict__ rsq,
                                                              double* __restrict__ cs, double* __restrict__ csq, double* __restrict__ stat,
                                                              const float* __restrict__ Phi2, int ld2) {
    __shared__ double pv[256], qv[256];
    __shared__ double sh[4];
    __shared__ double Cs[32 * 32];
    const int b = blockIdx.y, g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
    double* E = E2 + (long long)b * N2 * k1;
    const float* P = Phi1 + (long long)b * N1 * ld1;
    const float* a1 = mass1 + (long long)b * N1;
    if (t < k1) {
        pv[t] = p_in[(long long)b * k1 + t];
        const double* Cb = C + (long long)b * k2 * k1;
        double a = 0.0;
        for (int k = 0; k < k2; k += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = k + u < k2 ? Cb[(long long)(k + u) * k1 + t] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) a = fma(k + u < k2 ? s2_in[(long long)b * k2 + k + u] : 0.0, x[u], a);
        }
        qv[t] = a;
    }
    if (Phi2) for (int e = t; e < k2 * k1; e += 256) Cs[e] = C[(long long)b * k2 * k1 + e];      // (uniform)
    __syncthreads();
    const int i = g * 256 + t;
    double r = 0.0, cc = 0.0;
    if (i < N2 && Phi2) {
        const float* row = Phi2 + ((long long)b * N2 + i) * ld2;
        float x[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) x[k] = k < k2 ? row[k] : 0.f;
        for (int c = 0; c < k1; ++c) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 32; ++k) a = fma((double)x[k], k < k2 ? Cs[k * k1 + c] : 0.0, a);
            E[(long long)i * k1 + c] = a;
            r = fma(a, pv[c], r);
        }
        rs[(long long)b * N2 + i] = r; rsq[(long long)b * N2 + i] = 0.0;
    } else if (i < N2) {
        for (int k = 0; k < k1; k += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = k + u < k1 ? E[(long long)i * k1 + k + u] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) r = fma(x[u], k + u < k1 ? pv[k + u] : 0.0, r);
        }
        rs[(long long)b * N2 + i] = r; rsq[(long long)b * N2 + i] = 0.0;
    }
    if (i < N1) {
        for (int k = 0; k < k1; k += 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = k + u < k1 ? P[(long long)i * ld1 + k + u] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) cc = fma((double)x[u], k + u < k1 ? qv[k + u] : 0.0, cc);
        }
        cc *= (double)a1[i];
        cs[(long long)b * N1 + i] = cc; csq[(long long)b * N1 + i] = 0.0;
    }
    const double tr = block_sum_256(r, sh);
    const double tc = block_sum_256(cc, sh);
    if (t == 0) { stat[((long long)b * G + g) * 2 + 0] = tr; stat[((long long)b * G + g) * 2 + 1] = tc; }
}

// mean of the row sums (which = 0, over N2 rows) / column sums (which = 1, over N1 columns) of pair b from the G shares
__device__ __forceinline__ double em_mean(const double* __restrict__ stat, int b, int G, int which, int n) {
    double s = 0.0;
    for (int g = 0; g < G; ++g) s += stat[((long long)b * G + g) * 2 + which];
    return s / (double)n;
}

// ---- tile-fused passes over the mapped indicator (M is never stored) ------------------------------------------------------
// M_ij = (E2_i . Phi1_j) a1_j is produced 64 x 64 tile by tile on the f64 matrix cores from E2 = Phi2 C (N2 x k1) and the
// rows of Phi1, consumed in registers and dropped:
//   pass 1 (only when w_stochastic or w_sumto1 > 0): row sums / row sums of squares (complete inside a workgroup, which
//          owns EM_RG rows and sweeps every column tile) and column sums per row group (N2 / EM_RG partials per column);
//   pass 2: the tile again, the element-wise derivative dE/dM * a1_j, the energy, and at once the product of that tile
//          with the same Phi1 rows, Y_i += sum_j D'_ij Phi1_j (64 x k1 accumulators per workgroup) -- the gradient is then
//          Phi2^T Y.  Workspace O(N k) instead of the N2 x N1 matrix (32 MiB per pair at N = 2048, 537 MiB at N = 8192).
// Every sum has a fixed order (lane trees, then waves, then tiles ascending): the same input gives the same bits.
constexpr int EM_T = 64;       // tile edge
constexpr int EM_BK = 32;      // contraction chunk of the first product
constexpr int EM_LDA = 34;     // LDS row stride (doubles) of the staged E2 chunk
constexpr int EM_LDD = 66;     // LDS row stride of the derivative tile
constexpr int EM_RG = 512;     // rows per workgroup of pass 1

struct em_params {
    const double* E2; const float* Phi1; int ld1; const float* mass1;
    int N1, N2, k1, ldp;                       // ldp: LDS row stride (floats) of the Phi1 tile, >= k1 rounded up to 32, = 4 (mod 32)
    double* rs; double* rsq;                   // (ncs, B, N2): one set per column split (summed by m_finish_stats_kernel)
    double* pcs; double* pcsq; int ngroups;    // (B, ngroups, N1)
    int rg;                                    // rows per workgroup of pass 1 (multiple of 64, <= EM_RG)
    int ncs, cchunk;                           // column splits of both passes, columns per split (multiple of 64)
    int Bn;                                    // pairs in the batch (stride of the split partials)
    const double* cs; const double* csq; const double* stat; int nstat;     // stat: (B, nstat, 2) shares of the totals (em_mean)
    mterm_weights w;
    double* Y;                                 // (ncs, B, N2, k1): one partial per column split
    double* pe;                                // (B, ncs * N2 / 64) energy shares
};

// Phi1 rows j0 .. j0+63 (all k1 columns, zero padded) -> Ps[64][ldp], kept in fp32 (its memory type; widened at the
// fragment reads) so that the tile fits beside the other buffers up to k1 = 256
__device__ __forceinline__ void em_load_phi_tile(const em_params& p, int b, int j0, float* Ps, int t) {
    const float* P = p.Phi1 + (long long)b * p.N1 * p.ld1;
    const int kq = p.ldp;                      // fill the padding too (the second product reads whole 16-column tiles)
    for (int e = t; e < EM_T * kq; e += 256) {
        const int r = e / kq, c = e - r * kq;
        const int j = j0 + r;
        Ps[e] = (j < p.N1 && c < p.k1) ? P[(long long)j * p.ld1 + c] : 0.0f;
    }
}
// acc[mt][nt] = E2[r0 + wm*32 + mt*16 + ., :] . Phi1[j0 + wn*32 + nt*16 + ., :]   (64 x 64 tile, 4 waves 2 x 2)
__device__ __forceinline__ void em_tile_product(const em_params& p, int b, int r0, const float* Ps, double* As, f64x4 (&acc)[2][2],
                                                int t, int lane, int wm, int wn) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = f64x4{0.0, 0.0, 0.0, 0.0};
    const double* E = p.E2 + (long long)b * p.N2 * p.k1;
    const int lrow = t >> 2, lk = (t & 3) * 8;
    for (int kc = 0; kc < p.k1; kc += EM_BK) {
        double ra[8];
        const int i = r0 + lrow;
#pragma unroll
        for (int e = 0; e < 8; ++e) ra[e] = (i < p.N2 && kc + lk + e < p.k1) ? E[(long long)i * p.k1 + kc + lk + e] : 0.0;
        __syncthreads();                       // (the previous chunk's fragment reads are done)
#pragma unroll
        for (int e = 0; e < 8; ++e) As[lrow * EM_LDA + lk + e] = ra[e];
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < EM_BK / 4; ++ks) {
            const int kk = ks * 4 + (lane >> 4);
            double a[2], bb[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) a[mt] = As[(wm * 32 + mt * 16 + (lane & 15)) * EM_LDA + kk];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bb[nt] = (double)Ps[(wn * 32 + nt * 16 + (lane & 15)) * p.ldp + kc + kk];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_f64_16x16x4(a[mt], bb[nt], acc[mt][nt]);
        }
    }
}
__device__ __forceinline__ double em_row16_allsum(double x) {       // sum over the 16 lanes of a DPP row, fixed order
    x += __shfl_xor(x, 8); x += __shfl_xor(x, 4); x += __shfl_xor(x, 2); x += __shfl_xor(x, 1);
    return x;
}

// pass 1: grid (ngroups, ncs, B): workgroup = p.rg rows x one column split.  (A batch fills the chip with whole-row workgroups;
// a single pair -- the reference's one-pair-per-call use -- gets small row groups and column splits instead of 4 workgroups.)
__global__ __launch_bounds__(256) void em_stats_kernel(em_params p) {
    extern __shared__ __attribute__((aligned(16))) double em_sm[];
    double* As = em_sm;                                   // [64][EM_LDA]
    double* rsacc = As + EM_T * EM_LDA;                   // [EM_RG] row sums, [EM_RG] row sums of squares
    double* rqacc = rsacc + EM_RG;
    double* rowpart = rqacc + EM_RG;                      // [2 wn][2][64]
    double* colpart = rowpart + 2 * 2 * 64;               // [2 wm][2][64]
    float* Ps = reinterpret_cast<float*>(colpart + 2 * 2 * 64);   // [64][ldp]
    const int b = blockIdx.z, cs_ = blockIdx.y, g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int g0 = g * p.rg, nrb = min(p.rg, p.N2 - g0 + EM_T - 1) / EM_T;
    for (int r = t; r < EM_RG; r += 256) { rsacc[r] = 0.0; rqacc[r] = 0.0; }
    const float* a1 = p.mass1 + (long long)b * p.N1;
    const int jbeg = cs_ * p.cchunk, jend = min(p.N1, jbeg + p.cchunk);
    for (int j0 = jbeg; j0 < jend; j0 += EM_T) {
        __syncthreads();
        em_load_phi_tile(p, b, j0, Ps, t);
        double cacc = 0.0, cqacc = 0.0;                   // threads 64..127: column j0 + t - 64, over this group's rows
        double a1c[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int j = j0 + wn * 32 + nt * 16 + (lane & 15);
            a1c[nt] = j < p.N1 ? (double)a1[j] : 0.0;
        }
        for (int rb = 0; rb < nrb; ++rb) {
            f64x4 acc[2][2];
            em_tile_product(p, b, g0 + rb * EM_T, Ps, As, acc, t, lane, wm, wn);
            double cs_[2] = {0.0, 0.0}, cq_[2] = {0.0, 0.0};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int il = wm * 32 + mt * 16 + (lane >> 4) + 4 * r;
                    const bool rv = g0 + rb * EM_T + il < p.N2;
                    double s = 0.0, q = 0.0;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const double m = rv ? acc[mt][nt][r] * a1c[nt] : 0.0;
                        s += m; q += m * m;
                        cs_[nt] += m; cq_[nt] += m * m;
                    }
                    s = em_row16_allsum(s); q = em_row16_allsum(q);
                    if ((lane & 15) == 0) { rowpart[(wn * 2 + 0) * 64 + il] = s; rowpart[(wn * 2 + 1) * 64 + il] = q; }
                }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                double s = cs_[nt], q = cq_[nt];
                s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
                q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
                if (lane < 16) { colpart[(wm * 2 + 0) * 64 + wn * 32 + nt * 16 + lane] = s; colpart[(wm * 2 + 1) * 64 + wn * 32 + nt * 16 + lane] = q; }
            }
            __syncthreads();
            if (t < 64) {
                rsacc[rb * EM_T + t] += rowpart[(0 * 2 + 0) * 64 + t] + rowpart[(1 * 2 + 0) * 64 + t];
                rqacc[rb * EM_T + t] += rowpart[(0 * 2 + 1) * 64 + t] + rowpart[(1 * 2 + 1) * 64 + t];
            } else if (t < 128) {
                const int c = t - 64;
                cacc += colpart[(0 * 2 + 0) * 64 + c] + colpart[(1 * 2 + 0) * 64 + c];
                cqacc += colpart[(0 * 2 + 1) * 64 + c] + colpart[(1 * 2 + 1) * 64 + c];
            }
            // (rowpart / colpart are rewritten only after the next tile's staging barriers)
        }
        if (t >= 64 && t < 128 && j0 + t - 64 < p.N1) {
            const long long o = ((long long)b * p.ngroups + g) * p.N1 + j0 + t - 64;
            p.pcs[o] = cacc; p.pcsq[o] = cqacc;
        }
    }
    __syncthreads();
    for (int r = t; r < p.rg; r += 256)
        if (g0 + r < p.N2) {
            const long long o = ((long long)cs_ * p.Bn + b) * p.N2 + g0 + r;
            p.rs[o] = rsacc[r]; p.rsq[o] = rqacc[r];
        }
}

// pass 2: grid (N2 / 64, ncs, B); NT2 = 16-column tiles of Y per wave (k1 <= 64 NT2)
template <int NT2>
__global__ __launch_bounds__(256) void em_deriv_kernel(em_params p) {
    extern __shared__ __attribute__((aligned(16))) double em_sm[];
    __shared__ double sh[4];
    double* As = em_sm;                                   // [64][EM_LDA]
    double* Ds = As + EM_T * EM_LDA;                      // [64][EM_LDD]
    float* Ps = reinterpret_cast<float*>(Ds + EM_T * EM_LDD);     // [64][ldp]
    const int b = blockIdx.z, cs_ = blockIdx.y, rb = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int r0 = rb * EM_T;
    const int jbeg = cs_ * p.cchunk, jend = min(p.N1, jbeg + p.cchunk);
    const float* a1 = p.mass1 + (long long)b * p.N1;
    const mterm_weights w = p.w;
    const bool stats = w.stoch > 0.0 || w.sumto1 > 0.0;
    const double mean_r = stats ? em_mean(p.stat, b, p.nstat, 0, p.N2) : 0.0, mean_c = stats ? em_mean(p.stat, b, p.nstat, 1, p.N1) : 0.0;
    const double n2_over_n1 = (double)p.N2 / (double)p.N1;
    // this lane's rows: statistics terms
    double dr_s[2][4], dr_q[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = r0 + wm * 32 + mt * 16 + (lane >> 4) + 4 * r;
            const bool rv = stats && i < p.N2;
            dr_s[mt][r] = (rv && w.sumto1 > 0.0) ? p.rs[(long long)b * p.N2 + i] - mean_r : 0.0;
            dr_q[mt][r] = (rv && w.stoch > 0.0) ? p.rsq[(long long)b * p.N2 + i] - 1.0 : 0.0;
        }
    f64x4 Yacc[4][NT2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < NT2; ++q) Yacc[mt][q] = f64x4{0.0, 0.0, 0.0, 0.0};
    double eacc = 0.0;
    for (int j0 = jbeg; j0 < jend; j0 += EM_T) {
        __syncthreads();                                   // the previous tile's second product has read Ps / Ds
        em_load_phi_tile(p, b, j0, Ps, t);
        f64x4 acc[2][2];
        em_tile_product(p, b, r0, Ps, As, acc, t, lane, wm, wn);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int jl = wn * 32 + nt * 16 + (lane & 15), j = j0 + jl;
            const bool cv = j < p.N1;
            const double a1j = cv ? (double)a1[j] : 0.0;
            const double dc_q = (cv && w.stoch > 0.0) ? p.csq[(long long)b * p.N1 + j] - n2_over_n1 : 0.0;
            const double dc_s = (cv && w.sumto1 > 0.0) ? p.cs[(long long)b * p.N1 + j] - mean_c : 0.0;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int il = wm * 32 + mt * 16 + (lane >> 4) + 4 * r;
                    double d = 0.0;
                    if (cv && r0 + il < p.N2) {
                        const double m = acc[mt][nt][r] * a1j;
                        if (w.p2p > 0.0) { const double q = m * m - m; eacc += w.p2p * q * q; d += w.p2p * 2.0 * q * (2.0 * m - 1.0); }
                        if (w.stoch > 0.0) d += w.stoch * (2.0 * dc_q + 2.0 * dr_q[mt][r]) * 2.0 * m;
                        if (w.ent > 0.0) {
                            const double c = fmin(fmax(m, 0.0), 1.0);
                            const double lg = log(c + 1e-10);
                            eacc += w.ent * (-c * lg);
                            if (m >= 0.0 && m <= 1.0) d += w.ent * (-lg - c / (c + 1e-10));   // torch.clamp passes the gradient on [0, 1]
                        }
                        if (w.range01 > 0.0) {
                            const double lo = fmax(-m, 0.0), hi = fmax(m - 1.0, 0.0);
                            eacc += w.range01 * (lo * lo + hi * hi);
                            d += w.range01 * (-2.0 * lo + 2.0 * hi);
                        }
                        if (w.sumto1 > 0.0) d += w.sumto1 * (2.0 * dc_s + 2.0 * dr_s[mt][r]);
                        d *= a1j;
                    }
                    Ds[il * EM_LDD + jl] = d;
                }
        }
        __syncthreads();
        // Y[64 x k1] += D'[64 x 64] Phi1tile[64 x k1]: wave `wave` owns the 16-column tiles wave, wave + 4, ...
#pragma unroll 4
        for (int ks = 0; ks < EM_T / 4; ++ks) {
            const int kk = ks * 4 + (lane >> 4);
            double a[4], bb[NT2];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a[mt] = Ds[(mt * 16 + (lane & 15)) * EM_LDD + kk];
#pragma unroll
            for (int q = 0; q < NT2; ++q) {
                const int n = (wave + 4 * q) * 16 + (lane & 15);
                bb[q] = n < p.ldp ? (double)Ps[kk * p.ldp + n] : 0.0;
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int q = 0; q < NT2; ++q) Yacc[mt][q] = mfma_f64_16x16x4(a[mt], bb[q], Yacc[mt][q]);
        }
    }
    // the rows' shares of the row-statistics terms (once per row), then the workgroup's energy share
    if (stats && cs_ == 0 && t < 64 && r0 + t < p.N2) {
        const double ds = (w.sumto1 > 0.0) ? p.rs[(long long)b * p.N2 + r0 + t] - mean_r : 0.0;
        const double dq = (w.stoch > 0.0) ? p.rsq[(long long)b * p.N2 + r0 + t] - 1.0 : 0.0;
        eacc += w.stoch * dq * dq + w.sumto1 * ds * ds;
    }
    const double tot = block_sum_256(eacc, sh);
    if (t == 0) p.pe[((long long)b * gridDim.y + cs_) * gridDim.x + rb] = tot;
    double* Yb = p.Y + ((long long)cs_ * p.Bn + b) * p.N2 * p.k1;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < NT2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = r0 + mt * 16 + (lane >> 4) + 4 * r, n = (wave + 4 * q) * 16 + (lane & 15);
                if (i < p.N2 && n < p.k1) Yb[(long long)i * p.k1 + n] = Yacc[mt][q][r];
            }
}

// Y (set 0) += the other column splits' partials (fixed order)
__global__ __launch_bounds__(256) void em_sum_splits_kernel(double* __restrict__ Y, long long n, int ncs) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = Y[i];
    for (int c = 1; c < ncs; ++c) s += Y[c * n + i];
    Y[i] = s;
}

// e_dc[b] = 1/2 sum T^2 over the pair's nops * k2 * k1 entries   (one workgroup per pair)
__global__ __launch_bounds__(256) void half_sumsq_kernel(const double* __restrict__ T, long long n, double* __restrict__ out) {
    __shared__ double sh[4];
    const int b = blockIdx.x, t = threadIdx.x;
    double acc = 0.0;
    for (long long e = t; e < n; e += 256) { const double x = T[(long long)b * n + e]; acc += x * x; }
    const double tot = block_sum_256(acc, sh);
    if (t == 0) out[b] = 0.5 * tot;
}

// Area and conformal shape-difference terms (base_functions.py:228-294):
//   E_area = 1/2 |C^T C - I|^2,                grad = 2 C (C^T C - I)
//   E_conf = 1/2 |C^T D2 C - D1|^2,            grad = 2 D2 C (C^T D2 C - D1),    D = diag(lam / max(lam1.max, lam2.max))
// One workgroup per pair: the k1 x k1 difference matrices (already times their weights) go through a scratch block in global
// memory, then the k2 x k1 gradient is formed from them.  O(k^3) on the vector ALU: these terms belong to small maps (the
// iterative fit of the notebook runs 15 x 15); gs = w_area grad_area + w_conf grad_conf, es = the weighted energy.
__global__ __launch_bounds__(256) void shape_terms_kernel(const double* __restrict__ C, const double* __restrict__ lam1, const double* __restrict__ lam2,
                                                          int k1, int k2, double w_area, double w_conf, double* __restrict__ Msc,
                                                          double* __restrict__ gs, double* __restrict__ es) {
    __shared__ double sh[4];
    __shared__ double sscale;
    const int b = blockIdx.x, t = threadIdx.x;
    const double* Cb = C + (long long)b * k2 * k1;
    const double* l1 = lam1 + (long long)b * k1;
    const double* l2 = lam2 + (long long)b * k2;
    double m = -DM_INF_F64;
    for (int e = t; e < k1; e += 256) m = fmax(m, l1[e]);
    for (int e = t; e < k2; e += 256) m = fmax(m, l2[e]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((t & 63) == 0) sh[t >> 6] = m;
    __syncthreads();
    if (t == 0) sscale = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
    __syncthreads();
    const double scale = sscale;
    double* Ma = Msc + (long long)b * 2 * k1 * k1;
    double* Mc = Ma + (long long)k1 * k1;
    double acc = 0.0;
    for (int e = t; e < k1 * k1; e += 256) {
        const int a = e / k1, c = e - a * k1;
        double ta = 0.0, tc = 0.0;
        for (int i = 0; i < k2; ++i) {
            const double p = Cb[(long long)i * k1 + a] * Cb[(long long)i * k1 + c];
            ta += p;
            tc = fma(l2[i] / scale, p, tc);
        }
        const double da = ta - (a == c ? 1.0 : 0.0), dc = tc - (a == c ? l1[a] / scale : 0.0);
        acc += 0.5 * (w_area * da * da + w_conf * dc * dc);
        Ma[e] = w_area * da;
        Mc[e] = w_conf * dc;
    }
    const double tot = block_sum_256(acc, sh);
    if (t == 0) es[b] = tot;
    __syncthreads();                                            // (the workgroup's own global writes are visible to it behind the barrier)
    for (int e = t; e < k2 * k1; e += 256) {
        const int i = e / k1, c = e - i * k1;
        const double d2 = l2[i] / scale;
        double g = 0.0;
        for (int a = 0; a < k1; ++a) g = fma(Cb[(long long)i * k1 + a], 2.0 * (Ma[a * k1 + c] + d2 * Mc[a * k1 + c]), g);
        gs[(long long)b * k2 * k1 + e] = g;
    }
}

// grad += gm + w_dc (g1 - g2), column 0 zeroed; energy = e_quad + e_m + w_dc e_dc.  gm arrives as the split-K partials of
// Phi2^T Y (added here in split order, as reduce_partials_kernel would); e_m = the tile kernels' workgroup shares + the
// column-statistics terms (what m_finish_energy_kernel computes), summed here by the pair's workgroup.
struct combine_mterms {
    const double* gm_part; int gm_nsplit;
    const double* pe; int nblk, N1, N2; const double* cs; const double* csq; const double* stat; int nstat; mterm_weights mw;
};
__global__ __launch_bounds__(256) void combine_kernel(double* __restrict__ grad, combine_mterms cm, long long n_all, const double* __restrict__ g1,
                                                      const double* __restrict__ g2, double w_dc, int k1, int k2,
                                                      const double* __restrict__ e_quad, const double* __restrict__ e_dc, double* __restrict__ energy,
                                                      quad_args qa, const double* __restrict__ gs, const double* __restrict__ es) {
    __shared__ double sh[4];
    const int b = blockIdx.x, t = threadIdx.x;
    // small maps: the quadratic terms are evaluated here instead of by a launch of their own (qa.C != null)
    double eq = 0.0;
    if (qa.C) { eq = quad_pair(qa, b, t, k1, k2, grad, sh); __syncthreads(); }
    for (int e = t; e < k2 * k1; e += 256) {
        const long long o = (long long)b * k2 * k1 + e;
        double g = grad[o];
        if (cm.gm_part) {
            double s = 0.0;
            for (int q = 0; q < cm.gm_nsplit; ++q) s += cm.gm_part[(long long)q * n_all + o];
            g += s;
        }
        if (g1) g += w_dc * (g1[o] - g2[o]);
        if (gs) g += gs[o];                                             // area / conformal terms (shape_terms_kernel)
        grad[o] = (e % k1 == 0) ? 0.0 : g;                              // base_functions.py:759
    }
    double e_m = 0.0;
    if (cm.pe) {                                                        // (uniform)
        double acc = 0.0;
        for (int q = t; q < cm.nblk; q += 256) acc += cm.pe[(long long)b * cm.nblk + q];
        const double mean_c = em_mean(cm.stat, b, cm.nstat, 1, cm.N1), n2_over_n1 = (double)cm.N2 / (double)cm.N1;
        if (cm.mw.stoch > 0.0 || cm.mw.sumto1 > 0.0)
            for (int j = t; j < cm.N1; j += 256) {
                const double dq = cm.csq[(long long)b * cm.N1 + j] - n2_over_n1, ds = cm.cs[(long long)b * cm.N1 + j] - mean_c;
                acc += cm.mw.stoch * dq * dq + cm.mw.sumto1 * ds * ds;
            }
        e_m = block_sum_256(acc, sh);
    }
    if (t == 0) energy[b] = (qa.C ? eq : e_quad[b]) + (cm.pe ? e_m : 0.0) + (e_dc ? w_dc * e_dc[b] : 0.0) + (es ? es[b] : 0.0);
}

extern "C" int dm_fmap_energy_grad(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, int D, const float* Phi1, int ld1,
                                   const float* Phi2, int ld2, const float* mass1, const float* A, const float* Bm,
                                   const double* lam1, const double* lam2, const double* ops1, const double* ops2, int n_ops,
                                   const double* weights, const double* C, double* energy, double* grad) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N1 > 0 && N2 > 0 && k1 > 0 && k2 > 0 && D > 0, "sizes must be positive");
    DM_REQUIRE(ctx, A && Bm && lam1 && lam2 && weights && C && energy && grad, "null pointer");
    double w[W_COUNT];
    for (int q = 0; q < W_COUNT; ++q) {
        w[q] = weights[q];
        DM_REQUIRE(ctx, w[q] >= 0.0, "weights must be >= 0");
    }
    const mterm_weights mw{w[W_P2P], w[W_STOCH], w[W_ENT], w[W_RANGE01], w[W_SUMTO1]};
    const bool m_terms = mw.p2p > 0 || mw.stoch > 0 || mw.ent > 0 || mw.range01 > 0 || mw.sumto1 > 0;
    const bool dcomm = w[W_DCOMM] > 0.0 && n_ops > 0;
    const bool shape = w[W_AREA] > 0.0 || w[W_CONFORMAL] > 0.0;
    DM_REQUIRE(ctx, !m_terms || (Phi1 && Phi2 && mass1 && ld1 >= k1 && ld2 >= k2), "the indicator terms need Phi1, Phi2, mass1");
    DM_REQUIRE(ctx, !(w[W_DCOMM] > 0.0) || (ops1 && ops2 && n_ops > 0), "w_dcomm > 0 needs the descriptor operators (dm_fmap_descr_ops)");
    DM_REQUIRE(ctx, !(w[W_DCOMM] > 0.0) || (long long)B * n_ops <= 65535, "too many (pair, descriptor) slots for one launch: split the batch");
    DM_REQUIRE(ctx, B <= 65535, "batch too large for one launch");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));

    const size_t bKK = (size_t)B * k2 * k1 * 8;
    DM_REQUIRE(ctx, !m_terms || k1 <= 256, "the indicator terms need k1 <= 256");
    // Work decomposition of the two indicator passes: a workgroup sweeps `rg` rows x one column split.  A batch fills the chip
    // with whole-row sweeps (rg = 512, no splits); a single pair (the reference's use: one pair per call) would leave 252 of the
    // 256 CUs idle that way, so the row groups shrink to one 64-row block and the columns are split until ~1024 workgroups exist
    // (two per CU hide the other's log / divide chains; measured on the notebook fit, workgroups x column tiles each: 256 x 4
    // 56.7 + 30.6 us for the two passes, 512 x 2 46.0 + 21.6, 1024 x 1 50.6 + 21.1 with twice the partials to add up).
    int rg = EM_RG, ncs = 1;
    {
        const int ncu2 = 4 * (ctx->n_cu > 0 ? ctx->n_cu : 256);
        while (rg > EM_T && (long long)B * dm_cdiv(N2, rg) < ncu2) rg >>= 1;
        const int max_split = dm_cdiv(N1, 2 * EM_T) > 0 ? dm_cdiv(N1, 2 * EM_T) : 1;        // at least two column tiles per workgroup
        while (ncs < max_split && (long long)B * dm_cdiv(N2, EM_T) * ncs < ncu2) ncs <<= 1;
    }
    const int cchunk = pad_to(dm_cdiv(N1, ncs), EM_T);
    const int ngroups = dm_cdiv(N2, rg);
    // split-K chunk of Gm = Phi2^T Y: by the map's size only (a pair's sums must not depend on the batch it is in); a map of one
    // 64 x 64 tile has nothing else to fill the chip with, so its chunks are short (one pair, N2 = 2048: 4 -> 16 workgroups)
    const int kc_m = (k1 <= TN_T && k2 <= TN_T) ? 128 : 512;
    const int nsplit_m = dm_cdiv(N2, kc_m);
    const int nsplit_d = dcomm ? dm_cdiv(n_ops * k2, 512) : 0;
    size_t need = dm_align_up((size_t)B * (k1 + k2) * k1 * 8) + 4 * dm_align_up(bKK) + 4 * dm_align_up((size_t)B * 8) + 65536;
    if (m_terms)      // O(N k): the mapped indicator is never stored (em_stats_kernel / em_deriv_kernel)
        need += dm_align_up((size_t)B * N2 * k1 * 8) + dm_align_up((size_t)ncs * B * N2 * k1 * 8) + 2 * dm_align_up((size_t)ncs * B * N2 * 8) +
                2 * dm_align_up((size_t)B * N1 * 8) + 2 * dm_align_up((size_t)B * ngroups * N1 * 8) + dm_align_up((size_t)B * dm_cdiv(N1 > N2 ? N1 : N2, 256) * 2 * 8) +
                dm_align_up((size_t)B * ncs * dm_cdiv(N2, EM_T) * 8) + dm_align_up((size_t)nsplit_m * bKK) + 2 * dm_align_up((size_t)B * (k1 + k2) * 8);
    if (dcomm) need += dm_align_up((size_t)B * n_ops * k2 * k1 * 8) + dm_align_up((size_t)nsplit_d * bKK);
    if (shape) need += dm_align_up((size_t)B * 2 * k1 * k1 * 8) + dm_align_up(bKK) + dm_align_up((size_t)B * 8) + 4096;
    int rc = dm_ws_reserve(ctx, need);
    if (rc) return rc;
    double* PQ = (double*)dm_ws_take(ctx, (size_t)B * (k1 + k2) * k1 * 8);
    // P, Q do not depend on C: an L-BFGS driver asks to keep them across its evaluations (option "energy_keep_gram")
    const bool keep_gram = ctx->opt_energy_keep_gram != 0;
    bool sums_valid = false;                               // (p, s2 of the indicator statistics ride along with P, Q)
    if (keep_gram) {
        const size_t gb = (size_t)B * (k1 + k2) * k1 * 8 + (size_t)B * (k1 + k2) * 8;      // P, Q | p, s2 (em_basis_sums_kernel)
        const bool same = ctx->gram_valid && ctx->gram_key_ptr[2] == (const void*)Phi1 && ctx->gram_key_ptr[3] == (const void*)Phi2 &&
                          ctx->gram_key_ptr[4] == (const void*)mass1 && ctx->gram_key_dim[4] == N1 && ctx->gram_key_dim[5] == N2 && ctx->gram_key_ptr[0] == (const void*)A && ctx->gram_key_ptr[1] == (const void*)Bm &&
                          ctx->gram_key_dim[0] == B && ctx->gram_key_dim[1] == k1 && ctx->gram_key_dim[2] == k2 && ctx->gram_key_dim[3] == D;
        if (!same) {
            ctx->gram_valid = false;
            if (gb > ctx->gram_keep_bytes) {
                DM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (ctx->gram_keep) DM_CHECK_HIP(ctx, hipFree(ctx->gram_keep));
                ctx->gram_keep = nullptr; ctx->gram_keep_bytes = 0;
                DM_CHECK_HIP(ctx, hipMalloc((void**)&ctx->gram_keep, gb));
                ctx->gram_keep_bytes = gb;
            }
            ctx->gram_key_ptr[0] = A; ctx->gram_key_ptr[1] = Bm; ctx->gram_key_ptr[2] = Phi1; ctx->gram_key_ptr[3] = Phi2; ctx->gram_key_ptr[4] = mass1;
            ctx->gram_key_dim[4] = N1; ctx->gram_key_dim[5] = N2;
            ctx->gram_key_dim[0] = B; ctx->gram_key_dim[1] = k1; ctx->gram_key_dim[2] = k2; ctx->gram_key_dim[3] = D;
        }
        PQ = ctx->gram_keep;
        sums_valid = ctx->gram_valid && ctx->gram_sums_valid;
        if (!ctx->gram_valid) ctx->gram_sums_valid = false;
    }
    double* CP = (double*)dm_ws_take(ctx, bKK);
    double* Gm = (double*)dm_ws_take(ctx, bKK);
    double* G1 = (double*)dm_ws_take(ctx, bKK);
    double* G2 = (double*)dm_ws_take(ctx, bKK);
    double* e_quad = (double*)dm_ws_take(ctx, (size_t)B * 8);
    double* e_m = (double*)dm_ws_take(ctx, (size_t)B * 8);
    double* e_dc = (double*)dm_ws_take(ctx, (size_t)B * 8);
    if (!PQ || !CP || !Gm || !G1 || !G2 || !e_quad || !e_m || !e_dc) return dm_fail(ctx, DM_ENOMEM, "energy: workspace not reserved");

    // ---- quadratic terms: P = A A^T, Q = Bm A^T (unscaled), C P
    quad_args qz;
    {
        KRowsStackedF32 opa{A, Bm, k1, k2, D};
        KRowsF32 opb{A, (long long)k1 * D, D, k1, D};
        OutNT out{PQ, (long long)(k1 + k2) * k1, k1};
        if (!keep_gram || !ctx->gram_valid)
            DM_LAUNCH(ctx, "energy_gram_nt_f64", (gemm_nt_f64<KRowsStackedF32, KRowsF32, OutNT>),
                      dim3(dm_cdiv(k1 + k2, NT_T) * dm_cdiv(k1, NT_T), 1, B), dim3(256), 0, opa, opb, out, k1 + k2, k1, D);
        if (keep_gram) ctx->gram_valid = true;
        KRowsF64 ca{C, (long long)k2 * k1, k1, k2, k1, 0};
        KRowsF64 pb{PQ, (long long)(k1 + k2) * k1, k1, k1, k1, 1};          // (C P)_ij = sum_k C_ik P_kj
        OutNT ocp{CP, (long long)k2 * k1, k1};
        const bool cp_inline = k1 <= 32 && k2 <= 32;      // (by the map's size only)
        if (!cp_inline)
            DM_LAUNCH(ctx, "energy_cp_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutNT>), dim3(dm_cdiv(k2, NT_T) * dm_cdiv(k1, NT_T), 1, B),
                      dim3(256), 0, ca, pb, ocp, k2, k1, k1);
        qz = quad_args{C, cp_inline ? (const double*)nullptr : (const double*)CP, PQ, Bm, lam1, lam2, D, w[W_DESCR], w[W_LAP]};
        if (!cp_inline) {
            DM_LAUNCH(ctx, "energy_quad", quad_terms_kernel, dim3(B), dim3(256), 0, qz, k1, k2, grad, e_quad);
            qz.C = nullptr;                                // (done: combine_kernel reads grad / e_quad)
        }
    }

    // ---- terms in the mapped indicator
    combine_mterms cm;
    memset(&cm, 0, sizeof(cm));
    if (m_terms) {
        const int nrb = dm_cdiv(N2, EM_T);
        const bool stats = mw.stoch > 0 || mw.sumto1 > 0;
        double* E2 = (double*)dm_ws_take(ctx, (size_t)B * N2 * k1 * 8);            // Phi2 C
        double* Yv = (double*)dm_ws_take(ctx, (size_t)ncs * B * N2 * k1 * 8);      // Y = D' Phi1, one partial per column split
        double* rs = (double*)dm_ws_take(ctx, (size_t)ncs * B * N2 * 8);
        double* rsq = (double*)dm_ws_take(ctx, (size_t)ncs * B * N2 * 8);
        double* cs = (double*)dm_ws_take(ctx, (size_t)B * N1 * 8);
        double* csq = (double*)dm_ws_take(ctx, (size_t)B * N1 * 8);
        double* pcs = (double*)dm_ws_take(ctx, (size_t)B * ngroups * N1 * 8);
        double* pcsq = (double*)dm_ws_take(ctx, (size_t)B * ngroups * N1 * 8);
        const int nstat = dm_cdiv(N1 > N2 ? N1 : N2, 256);
        double* stat = (double*)dm_ws_take(ctx, (size_t)B * nstat * 2 * 8);
        double* pe = (double*)dm_ws_take(ctx, (size_t)B * ncs * nrb * 8);
        double* part = (double*)dm_ws_take(ctx, (size_t)nsplit_m * bKK);
        if (!E2 || !Yv || !rs || !rsq || !cs || !csq || !pcs || !pcsq || !stat || !pe || !part)
            return dm_fail(ctx, DM_ENOMEM, "energy: workspace not reserved");
        // (maps up to 32 x 32 whose statistics come from em_stats_linear_kernel: that kernel writes E2 as well)
        const bool e2_inline = stats && !(mw.stoch > 0) && k1 <= 32 && k2 <= 32;
        if (!e2_inline) {   // E2 = Phi2 C   (the left factor of the mapped indicator, convert.py:144)
            KRowsF32 opa{Phi2, (long long)N2 * ld2, ld2, N2, k2};
            KRowsF64 opb{C, (long long)k2 * k1, k1, k1, k2, 1};
            OutNT out{E2, (long long)N2 * k1, k1};
            DM_LAUNCH(ctx, "emb2_nt_f64", (gemm_nt_f64<KRowsF32, KRowsF64, OutNT>), dim3(dm_cdiv(N2, NT_T) * dm_cdiv(k1, NT_T), 1, B),
                      dim3(256), 0, opa, opb, out, N2, k1, k2);
        }
        em_params ep;
        memset(&ep, 0, sizeof(ep));
        ep.E2 = E2; ep.Phi1 = Phi1; ep.ld1 = ld1; ep.mass1 = mass1; ep.N1 = N1; ep.N2 = N2; ep.k1 = k1;
        ep.ldp = pad_to(k1, 32) + 4;
        ep.rg = rg; ep.ncs = ncs; ep.cchunk = cchunk; ep.Bn = B;
        ep.rs = rs; ep.rsq = rsq; ep.pcs = pcs; ep.pcsq = pcsq; ep.ngroups = ngroups; ep.cs = cs; ep.csq = csq; ep.stat = stat; ep.nstat = nstat;
        ep.w = mw; ep.Y = Yv; ep.pe = pe;
        if (stats && !(mw.stoch > 0) && k2 <= 256) {
            double* pv = keep_gram ? ctx->gram_keep + (size_t)B * (k1 + k2) * k1 : (double*)dm_ws_take(ctx, (size_t)B * k1 * 8);
            double* s2 = keep_gram ? pv + (size_t)B * k1 : (double*)dm_ws_take(ctx, (size_t)B * k2 * 8);
            if (!pv || !s2) return dm_fail(ctx, DM_ENOMEM, "energy: workspace not reserved");
            if (!keep_gram || !sums_valid) {
                int cw = 16;
                while (cw < k1 || cw < k2) cw <<= 1;           // (k1 <= 256; a wider k2 takes the tile pass below)
                DM_LAUNCH(ctx, "energy_basis_sums", em_basis_sums_kernel, dim3(B), dim3(1024), 0, Phi1, ld1, mass1, N1, k1, Phi2, ld2, N2, k2, cw, pv, s2);
                if (keep_gram) ctx->gram_sums_valid = true;
            }
            DM_LAUNCH(ctx, "energy_stats_linear", em_stats_linear_kernel, dim3(nstat, B), dim3(256), 0, E2, Phi1, ld1, mass1, N1, N2, k1, k2,
                      C, (const double*)pv, (const double*)s2, rs, rsq, cs, csq, stat, e2_inline ? Phi2 : (const float*)nullptr, ld2);
        } else if (stats) {
            const size_t lds1 = ((size_t)EM_T * EM_LDA + 2 * EM_RG + 2 * 2 * 2 * 64) * 8 + (size_t)EM_T * ep.ldp * 4;
            rc = dm_grant_lds(ctx, (const void*)em_stats_kernel, lds1);
            if (rc) return rc;
            DM_LAUNCH(ctx, "energy_stats_tiles", em_stats_kernel, dim3(ngroups, ncs, B), dim3(256), lds1, ep);
            DM_LAUNCH(ctx, "energy_finish_stats", m_finish_stats_kernel, dim3(nstat, B), dim3(256), 0, pcs, pcsq, ngroups, N2, N1, rs, rsq, ncs,
                      (long long)B * N2, cs, csq, stat);
        }
        {
            const size_t lds2 = ((size_t)EM_T * EM_LDA + EM_T * EM_LDD) * 8 + (size_t)EM_T * ep.ldp * 4;
            const int nt2 = dm_cdiv(dm_cdiv(k1, 16), 4);
#define EM_DERIV(NT2_)                                                                                                  \
            {                                                                                                          \
                rc = dm_grant_lds(ctx, (const void*)em_deriv_kernel<NT2_>, lds2);                                      \
                if (rc) return rc;                                                                                     \
                DM_LAUNCH(ctx, "energy_deriv_tiles", em_deriv_kernel<NT2_>, dim3(nrb, ncs, B), dim3(256), lds2, ep);   \
            }
            if (nt2 <= 1) EM_DERIV(1) else if (nt2 <= 2) EM_DERIV(2) else EM_DERIV(4)
#undef EM_DERIV
        }
        if (ncs > 1) {
            // (kept as its own launch: folded into the product's operand reads it made the four workgroups of a single pair's
            //  product walk ncs dependent loads per entry, 28 -> 122 us)
            const long long ny = (long long)B * N2 * k1;
            DM_LAUNCH(ctx, "energy_sum_splits", em_sum_splits_kernel, dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, Yv, ny, ncs);
        }
        {   // Gm = Phi2^T Y; its split-K partials are added by combine_kernel, which also finishes the terms' energy (two launches
            // fewer per evaluation, same additions)
            RowsF32Scaled opx{Phi2, (long long)N2 * ld2, ld2, k2, nullptr, 0};
            RowsF64TN opy{Yv, (long long)N2 * k1, k1, k1};
            OutTNPartial op{part, B, k2, k1};
            DM_LAUNCH(ctx, "energy_back_tn_f64", (gemm_tn_f64<RowsF32Scaled, RowsF64TN, OutTNPartial>),
                      dim3(dm_cdiv(k2, TN_T) * dm_cdiv(k1, TN_T), nsplit_m, B), dim3(256), 0, opx, opy, op, k2, k1, N2, kc_m);
        }
        cm.gm_part = part; cm.gm_nsplit = nsplit_m;
        cm.pe = pe; cm.nblk = ncs * nrb; cm.N1 = N1; cm.N2 = N2; cm.cs = cs; cm.csq = csq; cm.stat = stat; cm.nstat = nstat; cm.mw = mw;
    }

    // ---- descriptor commutativity: T_d = C L_d - R_d C;  G1 = sum_d T_d L_d^T;  G2 = sum_d R_d^T T_d
    if (dcomm) {
        double* T = (double*)dm_ws_take(ctx, (size_t)B * n_ops * k2 * k1 * 8);
        double* part = (double*)dm_ws_take(ctx, (size_t)nsplit_d * bKK);
        if (!T || !part) return dm_fail(ctx, DM_ENOMEM, "energy: workspace not reserved");
        const int Z = B * n_ops;
        const dim3 gz(dm_cdiv(k2, NT_T) * dm_cdiv(k1, NT_T), 1, Z);
        {
            KRowsF64Bcast ca{C, (long long)k2 * k1, k1, k2, k1, 0, n_ops};
            KRowsF64 lb{ops1, (long long)k1 * k1, k1, k1, k1, 1};          // (C L)_ij = sum_k C_ik L_kj
            OutNT ot{T, (long long)k2 * k1, k1};
            DM_LAUNCH(ctx, "dcomm_cl_nt_f64", (gemm_nt_f64<KRowsF64Bcast, KRowsF64, OutNT>), gz, dim3(256), 0, ca, lb, ot, k2, k1, k1);
            KRowsF64 ra{ops2, (long long)k2 * k2, k2, k2, k2, 0};
            KRowsF64Bcast cb{C, (long long)k2 * k1, k1, k1, k2, 1, n_ops};  // (R C)_ij = sum_k R_ik C_kj
            OutNTSub os{T, (long long)k2 * k1, k1};
            DM_LAUNCH(ctx, "dcomm_rc_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64Bcast, OutNTSub>), gz, dim3(256), 0, ra, cb, os, k2, k1, k2);
        }
        DM_LAUNCH(ctx, "dcomm_energy", half_sumsq_kernel, dim3(B), dim3(256), 0, T, (long long)n_ops * k2 * k1, e_dc);
        {
            KRowsF64Blocks ta{T, n_ops, k2, k1};
            KRowsF64Blocks lb{ops1, n_ops, k1, k1};
            OutNT og{G1, (long long)k2 * k1, k1};
            DM_LAUNCH(ctx, "dcomm_tl_nt_f64", (gemm_nt_f64<KRowsF64Blocks, KRowsF64Blocks, OutNT>),
                      dim3(dm_cdiv(k2, NT_T) * dm_cdiv(k1, NT_T), 1, B), dim3(256), 0, ta, lb, og, k2, k1, n_ops * k1);
            RowsF64TN rx{ops2, (long long)n_ops * k2 * k2, k2, k2};
            RowsF64TN ty{T, (long long)n_ops * k2 * k1, k1, k1};
            OutTNPartial op{nsplit_d > 1 ? part : G2, B, k2, k1};
            DM_LAUNCH(ctx, "dcomm_rt_tn_f64", (gemm_tn_f64<RowsF64TN, RowsF64TN, OutTNPartial>),
                      dim3(dm_cdiv(k2, TN_T) * dm_cdiv(k1, TN_T), nsplit_d, B), dim3(256), 0, rx, ty, op, k2, k1, n_ops * k2, 512);
            if (nsplit_d > 1) {
                const long long n = (long long)B * k2 * k1;
                DM_LAUNCH(ctx, "splitk_reduce", reduce_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, part, nsplit_d, n, G2);
            }
        }
    }
    double* gs = nullptr; double* es = nullptr;
    if (shape) {
        double* Msc = (double*)dm_ws_take(ctx, (size_t)B * 2 * k1 * k1 * 8);
        gs = (double*)dm_ws_take(ctx, bKK);
        es = (double*)dm_ws_take(ctx, (size_t)B * 8);
        if (!Msc || !gs || !es) return dm_fail(ctx, DM_ENOMEM, "energy_grad: workspace not reserved");
        DM_LAUNCH(ctx, "energy_shape_terms", shape_terms_kernel, dim3(B), dim3(256), 0, C, lam1, lam2, k1, k2, w[W_AREA], w[W_CONFORMAL], Msc, gs, es);
    }
    DM_LAUNCH(ctx, "energy_combine", combine_kernel, dim3(B), dim3(256), 0, grad, cm, (long long)B * k2 * k1,
              dcomm ? G1 : (const double*)nullptr, dcomm ? G2 : (const double*)nullptr, w[W_DCOMM], k1, k2, e_quad,
              dcomm ? e_dc : (const double*)nullptr, energy, qz, (const double*)gs, (const double*)es);
    return DM_OK;
}

// ops[b][d] = Phi^T diag(mass * F[:, d]) Phi   (k x k float64) for every descriptor d
extern "C" int dm_fmap_descr_ops(dm_ctx* ctx, int B, int N, int D, int k, const float* Phi, int ld, const float* mass, const void* F,
                                 int f_dtype, double* ops) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N > 0 && D > 0 && k > 0, "sizes must be positive");
    DM_REQUIRE(ctx, Phi && mass && F && ops, "null pointer");
    DM_REQUIRE(ctx, ld >= k, "eigenvector row stride smaller than k");
    DM_REQUIRE(ctx, f_dtype == DM_F16 || f_dtype == DM_F32, "f_dtype must be DM_F16 or DM_F32");
    DM_REQUIRE(ctx, (long long)B * D <= 65535, "too many (pair, descriptor) slots for one launch: split the batch");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    RowsPhiTimesDescr opx{Phi, (long long)N * ld, ld, k, mass, N, F, f_dtype == DM_F16, D};
    RowsPhiBatchDiv opy{Phi, (long long)N * ld, ld, k, D};
    OutTNPartial out{ops, (long long)B * D, k, k};
    DM_LAUNCH(ctx, "descr_ops_tn_f64", (gemm_tn_f64<RowsPhiTimesDescr, RowsPhiBatchDiv, OutTNPartial>),
              dim3(dm_cdiv(k, TN_T) * dm_cdiv(k, TN_T), 1, B * D), dim3(256), 0, opx, opy, out, k, k, N, N);
    return DM_OK;
}

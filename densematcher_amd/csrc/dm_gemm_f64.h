// Small batched float64 GEMM tiles on the f64 matrix cores (v_mfma_f64_16x16x4_f64).
// Two operand forms, both 64x64 output tiles, 256 threads = 4 waves (2x2), each wave
// a 32x32 sub-tile = 2x2 MFMA tiles (16 accumulator f64 per lane):
//
//   gemm_nt_f64<OpA, OpB, Out>:  O[i][j] = sum_k A[i][k] B[j][k]   (operand rows are K-contiguous)
//   gemm_tn_f64<OpX, OpY, Out>:  O[m][c] = sum_n X[n][m] Y[n][c]   (operands are K-major), split-K
//
// Operands are read through functors so the same tile code serves conversions
// (f16/f32 -> f64), row scaling (mass) and row gathers (p2p maps).
#pragma once
#include "dm_device.h"

// ----------------------------------------------------------------------------------------------
// NT form.  OpA/OpB: void load8(int b, int row, int k0, double (&v)[8]) const  -- 8 consecutive k,
// zero outside the operand.  Out: void store(int b, int i, int j, double v) const.
// grid = (tiles_i * tiles_j, 1, B)
// ----------------------------------------------------------------------------------------------
constexpr int NT_T = 64;     // tile
constexpr int NT_BK = 32;    // k per stage
constexpr int NT_LD = 34;    // padded LDS row stride (f64): 2*34 = 68 = 4 (mod 64) dwords -> conflict-free b64 reads

template <class OpA, class OpB, class Out>
__global__ __launch_bounds__(256) void gemm_nt_f64(OpA opa, OpB opb, Out out, int M, int N, int K) {
    __shared__ double As[NT_T * NT_LD];
    __shared__ double Bs[NT_T * NT_LD];
    const int tiles_j = (N + NT_T - 1) / NT_T;
    const int ti = blockIdx.x / tiles_j, tj = blockIdx.x % tiles_j;
    const int b = blockIdx.z;
    const int i0 = ti * NT_T, j0 = tj * NT_T;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = t >> 2, lk = (t & 3) * 8;

    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = f64x4{0.0, 0.0, 0.0, 0.0};

    double ra[8], rb[8];
    const int ns = (K + NT_BK - 1) / NT_BK;
    opa.load8(b, i0 + lrow, lk, ra);
    opb.load8(b, j0 + lrow, lk, rb);
    for (int s = 0; s < ns; ++s) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            As[lrow * NT_LD + lk + e] = ra[e];
            Bs[lrow * NT_LD + lk + e] = rb[e];
        }
        __syncthreads();
        if (s + 1 < ns) {
            opa.load8(b, i0 + lrow, (s + 1) * NT_BK + lk, ra);
            opb.load8(b, j0 + lrow, (s + 1) * NT_BK + lk, rb);
        }
#pragma unroll
        for (int ks = 0; ks < NT_BK / 4; ++ks) {
            const int kk = ks * 4 + (lane >> 4);
            double a[2], bb[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) a[mt] = As[(wm * 32 + mt * 16 + (lane & 15)) * NT_LD + kk];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bb[nt] = Bs[(wn * 32 + nt * 16 + (lane & 15)) * NT_LD + kk];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_f64_16x16x4(a[mt], bb[nt], acc[mt][nt]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wm * 32 + mt * 16 + (lane >> 4) + 4 * r;
                const int j = j0 + wn * 32 + nt * 16 + (lane & 15);
                if (i < M && j < N) out.store(b, i, j, acc[mt][nt][r]);
            }
}

// ----------------------------------------------------------------------------------------------
// TN form.  OpX/OpY: void load4(int b, int n, int col0, double (&v)[4]) const -- 4 consecutive
// columns of K-row n, zero outside.  Out: void store(int b, int split, int m, int c, double v).
// grid = (tiles_m * tiles_c, nsplit, B); split s handles K rows [s*kchunk, min((s+1)*kchunk, K)).
// ----------------------------------------------------------------------------------------------
constexpr int TN_T = 64;
constexpr int TN_BK = 16;
constexpr int TN_LD = 80;    // padded LDS row stride (f64): 160 = 32 (mod 64) dwords -> the two k rows of a
                             // half-wave land on disjoint bank halves

template <class OpX, class OpY, class Out>
__global__ __launch_bounds__(256) void gemm_tn_f64(OpX opx, OpY opy, Out out, int M, int N, int K, int kchunk) {
    __shared__ double Xs[2][TN_BK * TN_LD];
    __shared__ double Ys[2][TN_BK * TN_LD];
    const int tiles_c = (N + TN_T - 1) / TN_T;
    const int tm = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
    const int split = blockIdx.y, b = blockIdx.z;
    const int m0 = tm * TN_T, c0 = tc * TN_T;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = t >> 4, lc = (t & 15) * 4;
    const int kbeg = split * kchunk;
    const int kend = min(K, kbeg + kchunk);

    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = f64x4{0.0, 0.0, 0.0, 0.0};

    const int ns = (kend - kbeg + TN_BK - 1) / TN_BK;
    double rx[4], ry[4];
    // (macros, not lambdas: by-reference lambda captures of the staging arrays end up in scratch)
#define TN_FETCH(s_)                                                            \
    {                                                                           \
        const int n_ = kbeg + (s_) * TN_BK + lrow;                              \
        if (n_ < kend) {                                                        \
            opx.load4(b, n_, m0 + lc, rx);                                      \
            opy.load4(b, n_, c0 + lc, ry);                                      \
        } else {                                                                \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) rx[e] = ry[e] = 0.0;  \
        }                                                                       \
    }
#define TN_STASH(buf_)                                                          \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                             \
        Xs[buf_][lrow * TN_LD + lc + e] = rx[e];                                \
        Ys[buf_][lrow * TN_LD + lc + e] = ry[e];                                \
    }
    if (ns > 0) {
        TN_FETCH(0)
        TN_STASH(0)
    }
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        const int buf = s & 1;
        if (s + 1 < ns) TN_FETCH(s + 1)
#pragma unroll
        for (int ks = 0; ks < TN_BK / 4; ++ks) {
            const int kk = ks * 4 + (lane >> 4);
            double a[2], bb[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) a[mt] = Xs[buf][kk * TN_LD + wm * 32 + mt * 16 + (lane & 15)];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bb[nt] = Ys[buf][kk * TN_LD + wn * 32 + nt * 16 + (lane & 15)];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_f64_16x16x4(a[mt], bb[nt], acc[mt][nt]);
        }
        if (s + 1 < ns) { TN_STASH(buf ^ 1) }
        __syncthreads();
    }
#undef TN_FETCH
#undef TN_STASH
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + mt * 16 + (lane >> 4) + 4 * r;
                const int c = c0 + wn * 32 + nt * 16 + (lane & 15);
                if (m < M && c < N) out.store(b, split, m, c, acc[mt][nt][r]);
            }
}

// ----------------------------------------------------------------------------------------------
// common operand functors
// ----------------------------------------------------------------------------------------------
// rows of a float32 matrix (B, rows, ld), optionally scaled per row (mass), K-major use (load4)
struct RowsF32Scaled {
    const float* p; long long stride_b; int ld; int ncols;
    const float* scale; long long scale_stride_b;   // nullable
    __device__ __forceinline__ void load4(int b, int n, int col0, double (&v)[4]) const {
        const float* row = p + b * stride_b + (long long)n * ld;
        const double s = scale ? (double)scale[b * scale_stride_b + n] : 1.0;
        if (col0 + 3 < ncols && ((ld & 3) == 0) && ((((uintptr_t)p) & 15) == 0) && ((stride_b & 3) == 0)) {
            const float4 q = *reinterpret_cast<const float4*>(row + col0);
            v[0] = s * (double)q.x; v[1] = s * (double)q.y; v[2] = s * (double)q.z; v[3] = s * (double)q.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (col0 + e < ncols) ? s * (double)row[col0 + e] : 0.0;
        }
    }
};

// gathered, scaled rows of a float32 matrix: row n of the operand is scale[b][n] * p[b][idx[b][n]][:]
struct RowsF32GatherScaled {
    const float* p; long long stride_b; int ld; int ncols;
    const int32_t* idx; long long idx_stride_b; int nrows_src;
    const float* scale; long long scale_stride_b;
    __device__ __forceinline__ void load4(int b, int n, int col0, double (&v)[4]) const {
        int r = idx[b * idx_stride_b + n];
        r = min(max(r, 0), nrows_src - 1);
        const float* row = p + b * stride_b + (long long)r * ld;
        const double s = (double)scale[b * scale_stride_b + n];
        if (col0 + 3 < ncols && ((ld & 3) == 0) && ((((uintptr_t)p) & 15) == 0) && ((stride_b & 3) == 0)) {
            const float4 q = *reinterpret_cast<const float4*>(row + col0);
            v[0] = s * (double)q.x; v[1] = s * (double)q.y; v[2] = s * (double)q.z; v[3] = s * (double)q.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (col0 + e < ncols) ? s * (double)row[col0 + e] : 0.0;
        }
    }
};

// rows of an fp16 matrix (B, rows, ld), optionally scaled per row, K-major use
struct RowsF16Scaled {
    const _Float16* p; long long stride_b; int ld; int ncols;
    const float* scale; long long scale_stride_b;   // nullable
    __device__ __forceinline__ void load4(int b, int n, int col0, double (&v)[4]) const {
        const _Float16* row = p + b * stride_b + (long long)n * ld;
        const double s = scale ? (double)scale[b * scale_stride_b + n] : 1.0;
        if (col0 + 3 < ncols && ((ld & 3) == 0) && ((((uintptr_t)p) & 7) == 0) && ((stride_b & 3) == 0)) {
            const f16x4 q = *reinterpret_cast<const f16x4*>(row + col0);
            v[0] = s * (double)q[0]; v[1] = s * (double)q[1]; v[2] = s * (double)q[2]; v[3] = s * (double)q[3];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (col0 + e < ncols) ? s * (double)row[col0 + e] : 0.0;
        }
    }
};

// K-contiguous rows (NT form) of a float32 matrix, rows [0, nrows), columns [0, ncols)
struct KRowsF32 {
    const float* p; long long stride_b; int ld; int nrows; int ncols;
    __device__ __forceinline__ void load8(int b, int row, int k0, double (&v)[8]) const {
        if (row >= nrows) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.0;
            return;
        }
        const float* r = p + b * stride_b + (long long)row * ld;
        if (k0 + 7 < ncols && ((ld & 3) == 0) && ((((uintptr_t)p) & 15) == 0) && ((stride_b & 3) == 0)) {
            const float4 q0 = *reinterpret_cast<const float4*>(r + k0);
            const float4 q1 = *reinterpret_cast<const float4*>(r + k0 + 4);
            v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
            v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (k0 + e < ncols) ? (double)r[k0 + e] : 0.0;
        }
    }
};

// K-contiguous rows of a float64 matrix (B, nrows, ld); `trans` reads element (row,k) at p[k*ld + row]
struct KRowsF64 {
    const double* p; long long stride_b; int ld; int nrows; int ncols; int trans;
    __device__ __forceinline__ void load8(int b, int row, int k0, double (&v)[8]) const {
        const double* base = p + b * stride_b;
        if (!trans && row < nrows && k0 + 7 < ncols && ((ld & 1) == 0) && ((stride_b & 1) == 0) && ((((uintptr_t)p) & 15) == 0)) {
            const f64x2* q = reinterpret_cast<const f64x2*>(base + (long long)row * ld + k0);   // four 16-byte loads
#pragma unroll
            for (int e = 0; e < 4; ++e) { const f64x2 x = q[e]; v[2 * e] = x[0]; v[2 * e + 1] = x[1]; }
            return;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            double x = 0.0;
            if (row < nrows && k < ncols) x = trans ? base[(long long)k * ld + row] : base[(long long)row * ld + k];
            v[e] = x;
        }
    }
};

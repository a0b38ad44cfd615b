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
#include <type_traits>

#include "dm_device.h"

// Operand functors stage their data in the type it has in memory (`elem_t`, default double): the conversion to float64
// happens when the staged registers are written to LDS one stage later.  Converting inside load8 would make every fetch
// wait for its own data (s_waitcnt right behind the load) instead of overlapping the matrix instructions of the stage.
// Optional FAST form of an NT operand functor: `bool fast_ok(int K, int rows_used) const` (uniform: every 8-wide piece of the
// product's K range is a whole, aligned vector inside the operand, and the product does not read rows past the operand as zeros),
// `load8_fast` (no branch, no select: rows past the operand repeat the last one -- their result entries are not stored) and
// `cvt(const elem_t&, int row)` (what depends on the row is applied when the staged element is written to LDS, not behind the
// load).  Why: load8's "vector or element-wise" decision depends on the thread's k0, so the compiler emits both paths under exec
// masks and waits vmcnt(0) where they join -- every prefetch was waited for on the spot (r04, found in the Gram kernel's ISA).
template <class T, class = void> struct dm_has_fast { static constexpr bool value = false; };
template <class T> struct dm_has_fast<T, std::void_t<decltype(&T::fast_ok)>> { static constexpr bool value = true; };
// Out::skip(b) (optional): a wave-uniform "this batch entry has nothing to compute" -- gemm_nt_f64 then hands every output of the tile to
// Out::store_skipped(b, i, j) instead of running the product (the ICP's polar iteration: pairs that have converged), gemm_tn_f64 writes nothing
template <class T, class = void> struct dm_has_skip { static constexpr bool value = false; };
template <class T> struct dm_has_skip<T, std::void_t<decltype(&T::skip)>> { static constexpr bool value = true; };
template <class T, class = void> struct dm_elem { typedef double type; };
template <class T> struct dm_elem<T, std::void_t<typename T::elem_t>> { typedef typename T::elem_t type; };
typedef __attribute__((address_space(1))) const f32x4 dm_gf32x4;      // global address space: global_load, not flat_load
// TN form: a functor may define raw_t + load4raw / cvt / zero (same idea); otherwise load4 delivers doubles directly.
template <class T, class = void> struct dm_tn_raw {
    struct type { double v[4]; };
    static __device__ __forceinline__ int pre(const T&, int, int) { return 0; }
    static __device__ __forceinline__ void load(const T& op, int b, int n, int col0, int, type& r) { op.load4(b, n, col0, r.v); }
    static __device__ __forceinline__ void zero(type& r) { r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0.0; }
    static __device__ __forceinline__ void cvt(const type& r, double (&v)[4]) { v[0] = r.v[0]; v[1] = r.v[1]; v[2] = r.v[2]; v[3] = r.v[3]; }
};
template <class T> struct dm_tn_raw<T, std::void_t<typename T::raw_t>> {
    typedef typename T::raw_t type;
    // pre(): a value the row's load depends on (a gather index), fetched one stage earlier than the row itself
    static __device__ __forceinline__ int pre(const T& op, int b, int n) { return op.pre(b, n); }
    static __device__ __forceinline__ void load(const T& op, int b, int n, int col0, int pv, type& r) { op.load4raw(b, n, col0, pv, r); }
    static __device__ __forceinline__ void zero(type& r) { T::zero(r); }
    static __device__ __forceinline__ void cvt(const type& r, double (&v)[4]) { T::cvt(r, v); }
};
// four consecutive entries of a row (kept in their memory type) and the row's scale
template <typename TR, typename TS> struct dm_x4_scaled { TR q[4]; TS s; };
typedef __attribute__((address_space(1))) const f64x2 dm_gf64x2;
// four consecutive entries of a row: one 16-byte load (fp32) or two (fp64) when aligned, zero beyond ncols
template <typename TR>
__device__ __forceinline__ void dm_load_row4(const TR* row, int col0, int ncols, bool aligned, TR (&q)[4]) {
    if (col0 + 3 < ncols && aligned) {
        if constexpr (sizeof(TR) == 4) {
            const f32x4 x = *(dm_gf32x4*)(row + col0);
            q[0] = x[0]; q[1] = x[1]; q[2] = x[2]; q[3] = x[3];
        } else {
            const f64x2 x0 = ((dm_gf64x2*)(row + col0))[0], x1 = ((dm_gf64x2*)(row + col0))[1];
            q[0] = x0[0]; q[1] = x0[1]; q[2] = x1[0]; q[3] = x1[1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = (col0 + e < ncols) ? row[col0 + e] : (TR)0;
    }
}
// can rows of a (B, rows, ld) matrix of TR be read with 16-byte loads at column offsets that are multiples of 4?
template <typename TR>
__device__ __forceinline__ bool dm_rows_aligned(const TR* p, long long stride_b, int ld) {
    constexpr int m = sizeof(TR) == 4 ? 3 : 1;
    return ((ld & m) == 0) && ((((uintptr_t)p) & 15) == 0) && ((stride_b & m) == 0);
}

// ----------------------------------------------------------------------------------------------
// NT form.  OpA/OpB: void load8(int b, int row, int k0, elem_t (&v)[8]) const  -- 8 consecutive k,
// zero outside the operand.  Out: void store(int b, int i, int j, double v) const.
// grid = (tiles_i * tiles_j, 1, B)
// ----------------------------------------------------------------------------------------------
constexpr int NT_T = 64;     // tile
constexpr int NT_BK = 32;    // k per stage
constexpr int NT_LD = 34;    // padded LDS row stride (f64): 2*34 = 68 = 4 (mod 64) dwords -> conflict-free b64 reads

// NPRE: stages of operand loads in flight (register sets).  1 = the next stage is requested while the current one is multiplied:
// a stage is 32 matrix instructions per wave (~1 us), a global round trip ~2 us, so short products (K = 128: four stages) and
// staged conversions wait for their operands every stage; NPRE = 3 - 4 keeps that many stages in flight (the loads past the last
// stage re-read it: unconditional, so the compiler counts them instead of waiting for all).  Same sums in the same order.
template <bool FAST, class Op>
__device__ __forceinline__ void nt_load8(const Op& op, int b, int row, int k0, typename dm_elem<Op>::type (&v)[8]) {
    if constexpr (FAST) op.load8_fast(b, row, k0, v); else op.load8(b, row, k0, v);
}
template <bool FAST, class Op>
__device__ __forceinline__ double nt_cvt(const Op& op, const typename dm_elem<Op>::type& x, int row) {
    if constexpr (FAST) return op.cvt(x, row); else return (double)x;
}
template <bool FAST, int NPRE, class OpA, class OpB>
__device__ __forceinline__ void gemm_nt_body(const OpA& opa, const OpB& opb, int b, int i0, int j0, int K, double* As, double* Bs,
                                             f64x4 (&acc)[2][2]) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = t >> 2, lk = (t & 3) * 8;
    typename dm_elem<OpA>::type ra[NPRE][8];
    typename dm_elem<OpB>::type rb[NPRE][8];
    const int ns = (K + NT_BK - 1) / NT_BK;
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
        const int sl = (NPRE == 1 || u < ns) ? u : ns - 1;
        nt_load8<FAST>(opa, b, i0 + lrow, sl * NT_BK + lk, ra[u]);
        nt_load8<FAST>(opb, b, j0 + lrow, sl * NT_BK + lk, rb[u]);
    }
    for (int s0 = 0; s0 < ns; s0 += NPRE) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int s = s0 + u;
            if (s < ns) {                                          // (uniform)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    As[lrow * NT_LD + lk + e] = nt_cvt<FAST>(opa, ra[u][e], i0 + lrow);
                    Bs[lrow * NT_LD + lk + e] = nt_cvt<FAST>(opb, rb[u][e], j0 + lrow);
                }
                __syncthreads();
                if (NPRE > 1 || s + 1 < ns) {
                    const int sl = s + NPRE < ns ? s + NPRE : ns - 1;
                    nt_load8<FAST>(opa, b, i0 + lrow, sl * NT_BK + lk, ra[u]);
                    nt_load8<FAST>(opb, b, j0 + lrow, sl * NT_BK + lk, rb[u]);
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
        }
    }
}

// NPRE: stages of operand loads in flight (register sets).  1 = the next stage is requested while the current one is multiplied:
// a stage is 32 matrix instructions per wave (~1 us), a global round trip ~2 us, so short products (K = 128: four stages) wait
// for their operands every stage; NPRE = 2 - 4 keeps that many stages in flight (the loads past the last stage re-read it:
// unconditional, so the compiler counts them instead of waiting for all).  Same sums in the same order.
// XCD: a one-dimensional grid of (pairs x tiles) workgroups in an XCD-aware order -- the tiles of a pair run on ONE XCD, so the operand
// panels they share are fetched into one L2 instead of eight (the headline's Gram product: 8 tiles per pair, 351 MB fetched for 100 MB
// of operands when its tiles were dealt round-robin over the XCDs)
template <class OpA, class OpB, class Out, int NPRE = 1, bool XCD = false>
__global__ __launch_bounds__(256) void gemm_nt_f64(OpA opa, OpB opb, Out out, int M, int N, int K) {
    __shared__ double As[NT_T * NT_LD];
    __shared__ double Bs[NT_T * NT_LD];
    const int tiles_j = (N + NT_T - 1) / NT_T;
    const int tiles_all = ((M + NT_T - 1) / NT_T) * tiles_j;
    const int vid = XCD ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tile_id = XCD ? vid % tiles_all : vid;
    const int ti = tile_id / tiles_j, tj = tile_id % tiles_j;
    const int b = XCD ? vid / tiles_all : (int)blockIdx.z;
    const int i0 = ti * NT_T, j0 = tj * NT_T;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    if constexpr (dm_has_skip<Out>::value) {
        if (out.skip(b)) {                                    // (uniform)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = i0 + wm * 32 + mt * 16 + (lane >> 4) + 4 * r;
                        const int j = j0 + wn * 32 + nt * 16 + (lane & 15);
                        if (i < M && j < N) out.store_skipped(b, i, j);
                    }
            return;
        }
    }

    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = f64x4{0.0, 0.0, 0.0, 0.0};

    if constexpr (dm_has_fast<OpA>::value && dm_has_fast<OpB>::value) {
        if (opa.fast_ok(K, M) && opb.fast_ok(K, N)) gemm_nt_body<true, NPRE>(opa, opb, b, i0, j0, K, As, Bs, acc);   // (uniform)
        else gemm_nt_body<false, NPRE>(opa, opb, b, i0, j0, K, As, Bs, acc);
    } else {
        gemm_nt_body<false, NPRE>(opa, opb, b, i0, j0, K, As, Bs, acc);
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
    if constexpr (dm_has_skip<Out>::value) { if (out.skip(b)) return; }            // (uniform)
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
    // Operands are fetched TWO stages ahead into two register sets (a stage is only 16 MFMAs per wave, ~0.4 us: shorter
    // than a gathered row's round trip), gather indices one stage before their rows.
    typename dm_tn_raw<OpX>::type rxa, rxb;
    typename dm_tn_raw<OpY>::type rya, ryb;
    // (macros, not lambdas: by-reference lambda captures of the staging arrays end up in scratch)
    int px = 0, py = 0;
#define TN_PRE(s_)                                                              \
    {                                                                           \
        const int n_ = kbeg + (s_) * TN_BK + lrow;                              \
        if (n_ < kend) { px = dm_tn_raw<OpX>::pre(opx, b, n_); py = dm_tn_raw<OpY>::pre(opy, b, n_); } \
    }
#define TN_FETCH(s_, rx_, ry_)                                                  \
    {                                                                           \
        const int n_ = kbeg + (s_) * TN_BK + lrow;                              \
        if (n_ < kend) {                                                        \
            dm_tn_raw<OpX>::load(opx, b, n_, m0 + lc, px, rx_);                 \
            dm_tn_raw<OpY>::load(opy, b, n_, c0 + lc, py, ry_);                 \
        } else {                                                                \
            dm_tn_raw<OpX>::zero(rx_);                                          \
            dm_tn_raw<OpY>::zero(ry_);                                          \
        }                                                                       \
    }
#define TN_STASH(buf_, rx_, ry_)                                                \
    {                                                                           \
        double vx_[4], vy_[4];                                                  \
        dm_tn_raw<OpX>::cvt(rx_, vx_);                                          \
        dm_tn_raw<OpY>::cvt(ry_, vy_);                                          \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                         \
            Xs[buf_][lrow * TN_LD + lc + e] = vx_[e];                           \
            Ys[buf_][lrow * TN_LD + lc + e] = vy_[e];                           \
        }                                                                       \
    }
#define TN_COMPUTE(buf_)                                                        \
    _Pragma("unroll") for (int ks = 0; ks < TN_BK / 4; ++ks) {                  \
        const int kk = ks * 4 + (lane >> 4);                                    \
        double a[2], bb[2];                                                     \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) a[mt] = Xs[buf_][kk * TN_LD + wm * 32 + mt * 16 + (lane & 15)];  \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) bb[nt] = Ys[buf_][kk * TN_LD + wn * 32 + nt * 16 + (lane & 15)]; \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                        \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                    \
                if (vm[mt] && vn[nt]) acc[mt][nt] = mfma_f64_16x16x4(a[mt], bb[nt], acc[mt][nt]);  \
    }
    // 16 x 16 blocks that lie completely outside the M x N result are skipped (wave-uniform): the result is rarely a
    // multiple of the 64 x 64 tile (ZoomOut sweeps k = 51 .. 200) and the f64 matrix pipe is what bounds this kernel
    const bool vm[2] = {m0 + wm * 32 < M, m0 + wm * 32 + 16 < M};
    const bool vn[2] = {c0 + wn * 32 < N, c0 + wn * 32 + 16 < N};
    // stage s lives in LDS buffer s & 1 and was staged in register set s & 1 (a: even, b: odd)
    if (ns > 0) {
        TN_PRE(0)
        TN_FETCH(0, rxa, rya)
        if (ns > 1) {
            TN_PRE(1)
            TN_FETCH(1, rxb, ryb)
            if (ns > 2) TN_PRE(2)
        }
        TN_STASH(0, rxa, rya)
    }
    __syncthreads();
    for (int s = 0; s < ns; s += 2) {
        // even stage: rows of s + 2 into set a (its stage s is in LDS), compute, stash s + 1 from set b
        if (s + 2 < ns) {
            TN_FETCH(s + 2, rxa, rya)
            if (s + 3 < ns) TN_PRE(s + 3)
        }
        TN_COMPUTE(0)
        if (s + 1 < ns) TN_STASH(1, rxb, ryb)
        __syncthreads();
        if (s + 1 >= ns) break;
        // odd stage
        if (s + 3 < ns) {
            TN_FETCH(s + 3, rxb, ryb)
            if (s + 4 < ns) TN_PRE(s + 4)
        }
        TN_COMPUTE(1)
        if (s + 2 < ns) TN_STASH(0, rxa, rya)
        __syncthreads();
    }
#undef TN_COMPUTE
#undef TN_PRE
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
// rows of a real matrix (B, rows, ld) of TR = float | double, optionally scaled per row (mass, TS = float | double),
// K-major use (load4).  The product scale * entry is formed in float64 (one rounding: what the reference's A @ Phi does).
template <typename TR, typename TS>
struct RowsScaled {
    typedef dm_x4_scaled<TR, TS> raw_t;
    const TR* p; long long stride_b; int ld; int ncols;
    const TS* scale; long long scale_stride_b;   // nullable
    __device__ __forceinline__ int pre(int, int) const { return 0; }
    __device__ __forceinline__ void load4raw(int b, int n, int col0, int, raw_t& r) const {
        const TR* row = p + b * stride_b + (long long)n * ld;
        r.s = scale ? scale[b * scale_stride_b + n] : (TS)1;
        dm_load_row4<TR>(row, col0, ncols, dm_rows_aligned<TR>(p, stride_b, ld), r.q);
    }
    static __device__ __forceinline__ void zero(raw_t& r) { r.q[0] = r.q[1] = r.q[2] = r.q[3] = (TR)0; r.s = (TS)0; }
    static __device__ __forceinline__ void cvt(const raw_t& r, double (&v)[4]) {
        const double s = (double)r.s;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = s * (double)r.q[e];
    }
};
typedef RowsScaled<float, float> RowsF32Scaled;

// gathered, scaled rows: row n of the operand is scale[b][n] * p[b][idx[b][n]][:]
template <typename TR, typename TS>
struct RowsGatherScaled {
    typedef dm_x4_scaled<TR, TS> raw_t;
    const TR* p; long long stride_b; int ld; int ncols;
    const int32_t* idx; long long idx_stride_b; int nrows_src;
    const TS* scale; long long scale_stride_b;
    __device__ __forceinline__ int pre(int b, int n) const { return idx[b * idx_stride_b + n]; }
    __device__ __forceinline__ void load4raw(int b, int n, int col0, int pv, raw_t& r) const {
        const int ri = min(max(pv, 0), nrows_src - 1);
        const TR* row = p + b * stride_b + (long long)ri * ld;
        r.s = scale[b * scale_stride_b + n];
        dm_load_row4<TR>(row, col0, ncols, dm_rows_aligned<TR>(p, stride_b, ld), r.q);
    }
    static __device__ __forceinline__ void zero(raw_t& r) { r.q[0] = r.q[1] = r.q[2] = r.q[3] = (TR)0; r.s = (TS)0; }
    static __device__ __forceinline__ void cvt(const raw_t& r, double (&v)[4]) {
        const double s = (double)r.s;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = s * (double)r.q[e];
    }
};

// rows of an fp16 matrix (B, rows, ld), optionally scaled per row, K-major use
template <typename TS = float>
struct RowsF16Scaled {
    const _Float16* p; long long stride_b; int ld; int ncols;
    const TS* scale; long long scale_stride_b;   // nullable
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

// K-contiguous rows (NT form) of a real matrix (TR = float | double), rows [0, nrows), columns [0, ncols)
template <typename TR>
struct KRows {
    typedef TR elem_t;
    const TR* p; long long stride_b; int ld; int nrows; int ncols;
    __device__ __forceinline__ void load8(int b, int row, int k0, TR (&v)[8]) const {
        // rows beyond the operand only feed outputs the caller masks or rows of zeros: clamp instead of branching
        const TR* r = p + b * stride_b + (long long)min(row, nrows - 1) * ld;
        const bool in = row < nrows;
        if (k0 + 7 < ncols && dm_rows_aligned<TR>(p, stride_b, ld)) {
            TR q0[4], q1[4];
            dm_load_row4<TR>(r, k0, ncols, true, q0);
            dm_load_row4<TR>(r, k0 + 4, ncols, true, q1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = in ? q0[e] : (TR)0; v[4 + e] = in ? q1[e] : (TR)0; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (in && k0 + e < ncols) ? r[k0 + e] : (TR)0;
        }
    }
};
typedef KRows<float> KRowsF32;

// two fp32 matrices stacked: rows 0 .. k1-1 = A (B, k1, D), rows k1 .. k1+k2-1 = Bm (B, k2, D); K = D contiguous
struct KRowsStackedF32 {
    typedef float elem_t;
    const float* A; const float* Bm; int k1, k2, D;
    __device__ __forceinline__ void load8(int b, int row, int k0, float (&v)[8]) const {
        const bool in = row < k1 + k2;
        const int rc = in ? row : 0;
        const float* r = (rc < k1) ? A + ((long long)b * k1 + rc) * D : Bm + ((long long)b * k2 + (rc - k1)) * D;
        if (k0 + 7 < D && ((D & 3) == 0) && (((((uintptr_t)A) | ((uintptr_t)Bm)) & 15) == 0)) {
            const f32x4 q0 = *(dm_gf32x4*)(r + k0), q1 = *(dm_gf32x4*)(r + k0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = in ? q0[e] : 0.0f; v[4 + e] = in ? q1[e] : 0.0f; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (in && k0 + e < D) ? r[k0 + e] : 0.0f;
        }
    }
};

// The same two operands read from the split-K partials of the projections (dm_project.hip), BEFORE their reduction: the value
// of an entry is float(double(p0) + double(p1)) -- exactly what the reduce kernel would have stored -- formed when the staged
// pair is written to LDS (staging keeps the two floats as loaded).  Saves the reduce launches and their round trip through
// HBM when a caller does not need the projected descriptors themselves (dm_fmap_fit).  One or two chunks per operand: the
// stride to the second chunk is 0 when there is only one (its entries are then the stored values themselves).
struct PartPair {
    float p0, p1;
    // (the fp32 sum IS float(double(p0) + double(p1)): the double sum of two floats is exact or rounds innocuously, 53 >= 2 * 24 + 2;
    //  one fp32 add + one conversion instead of four float64-rate operations per staged element -- vector ALU work that float64
    //  matrix instructions do not overlap with)
    __device__ __forceinline__ operator double() const { return (double)(p0 + p1); }
    __device__ __forceinline__ double sum(bool two) const { return (double)(two ? p0 + p1 : p0); }
};
struct KRowsStackedPart {
    typedef PartPair elem_t;
    const float* A; const float* Bm; long long nA, nB;     // second chunk of A at A + nA (= B k1 D; 0: none), of Bm at Bm + nB
    int k1, k2, D;
    __device__ __forceinline__ bool fast_ok(int K, int rows_used) const {
        return ((K + NT_BK - 1) / NT_BK) * NT_BK <= D && (D & 3) == 0 && (((((uintptr_t)A) | ((uintptr_t)Bm)) & 15) == 0) && ((nA | nB) & 3) == 0 &&
               rows_used <= k1 + k2;
    }
    __device__ __forceinline__ void load8_fast(int b, int row, int k0, PartPair (&v)[8]) const {
        const int rc = min(row, k1 + k2 - 1);
        const bool isA = rc < k1;
        const float* r = isA ? A + ((long long)b * k1 + rc) * D : Bm + ((long long)b * k2 + (rc - k1)) * D;
        const long long nq = isA ? nA : nB;                // (0: the second load re-reads the first chunk; cvt ignores it)
        const f32x4 a0 = *(dm_gf32x4*)(r + k0), a1 = *(dm_gf32x4*)(r + k0 + 4);
        const f32x4 b0 = *(dm_gf32x4*)(r + nq + k0), b1 = *(dm_gf32x4*)(r + nq + k0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = PartPair{a0[e], b0[e]}; v[4 + e] = PartPair{a1[e], b1[e]}; }
    }
    __device__ __forceinline__ double cvt(const PartPair& x, int row) const { return x.sum((min(row, k1 + k2 - 1) < k1 ? nA : nB) != 0); }
    __device__ __forceinline__ void load8(int b, int row, int k0, PartPair (&v)[8]) const {
        const bool in = row < k1 + k2;
        const int rc = in ? row : 0;
        const bool isA = rc < k1;
        const float* r = isA ? A + ((long long)b * k1 + rc) * D : Bm + ((long long)b * k2 + (rc - k1)) * D;
        const long long nq = isA ? nA : nB;
        const bool two = nq != 0;
        if (k0 + 7 < D && ((D & 3) == 0) && (((((uintptr_t)A) | ((uintptr_t)Bm)) & 15) == 0) && ((nA | nB) & 3) == 0) {
            const f32x4 a0 = *(dm_gf32x4*)(r + k0), a1 = *(dm_gf32x4*)(r + k0 + 4);
            const f32x4 b0 = *(dm_gf32x4*)(r + nq + k0), b1 = *(dm_gf32x4*)(r + nq + k0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = PartPair{in ? a0[e] : 0.0f, (in && two) ? b0[e] : 0.0f};
                v[4 + e] = PartPair{in ? a1[e] : 0.0f, (in && two) ? b1[e] : 0.0f};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = in && k0 + e < D;
                v[e] = PartPair{ok ? r[k0 + e] : 0.0f, (ok && two) ? r[nq + k0 + e] : 0.0f};
            }
        }
    }
};
struct KRowsPart {                                         // the rows of A alone (the second operand of the Gram product)
    typedef PartPair elem_t;
    const float* A; long long nA; int k1, D;               // nA: stride to the second chunk, 0: none
    __device__ __forceinline__ bool fast_ok(int K, int rows_used) const {
        return ((K + NT_BK - 1) / NT_BK) * NT_BK <= D && (D & 3) == 0 && ((((uintptr_t)A) & 15) == 0) && (nA & 3) == 0 && rows_used <= k1;
    }
    __device__ __forceinline__ void load8_fast(int b, int row, int k0, PartPair (&v)[8]) const {
        const float* r = A + ((long long)b * k1 + min(row, k1 - 1)) * D;
        const f32x4 a0 = *(dm_gf32x4*)(r + k0), a1 = *(dm_gf32x4*)(r + k0 + 4);
        const f32x4 b0 = *(dm_gf32x4*)(r + nA + k0), b1 = *(dm_gf32x4*)(r + nA + k0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = PartPair{a0[e], b0[e]}; v[4 + e] = PartPair{a1[e], b1[e]}; }
    }
    __device__ __forceinline__ double cvt(const PartPair& x, int) const { return x.sum(nA != 0); }
    __device__ __forceinline__ void load8(int b, int row, int k0, PartPair (&v)[8]) const {
        const bool in = row < k1;
        const float* r = A + ((long long)b * k1 + min(row, k1 - 1)) * D;
        const long long nq = nA;
        const bool two = nq != 0;
        if (k0 + 7 < D && ((D & 3) == 0) && ((((uintptr_t)A) & 15) == 0) && (nA & 3) == 0) {
            const f32x4 a0 = *(dm_gf32x4*)(r + k0), a1 = *(dm_gf32x4*)(r + k0 + 4);
            const f32x4 b0 = *(dm_gf32x4*)(r + nq + k0), b1 = *(dm_gf32x4*)(r + nq + k0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = PartPair{in ? a0[e] : 0.0f, (in && two) ? b0[e] : 0.0f};
                v[4 + e] = PartPair{in ? a1[e] : 0.0f, (in && two) ? b1[e] : 0.0f};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = in && k0 + e < D;
                v[e] = PartPair{ok ? r[k0 + e] : 0.0f, (ok && two) ? r[nq + k0 + e] : 0.0f};
            }
        }
    }
};

// K-contiguous rows of a float64 matrix (B, nrows, ld); `trans` reads element (row,k) at p[k*ld + row]
struct KRowsF64 {
    const double* p; long long stride_b; int ld; int nrows; int ncols; int trans;
    __device__ __forceinline__ bool fast_ok(int K, int rows_used) const {
        return !trans && ((K + NT_BK - 1) / NT_BK) * NT_BK <= ncols && ((ld & 1) == 0) && ((stride_b & 1) == 0) && ((((uintptr_t)p) & 15) == 0) &&
               rows_used <= nrows;
    }
    __device__ __forceinline__ void load8_fast(int b, int row, int k0, double (&v)[8]) const {
        const f64x2* q = reinterpret_cast<const f64x2*>(p + b * stride_b + (long long)min(row, nrows - 1) * ld + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const f64x2 x = q[e]; v[2 * e] = x[0]; v[2 * e + 1] = x[1]; }
    }
    __device__ __forceinline__ double cvt(const double& x, int) const { return x; }
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

// Functional map -> vertex maps (dm_fm_to_p2p) and its building blocks.
//
//   G = Phi2[:, :k2] C Phi1[:, :k1]^T  is produced 128x128 tile by tile on the float64
//   matrix cores (v_mfma_f64_16x16x4_f64) and consumed in registers by four fused
//   arg-reductions; it never reaches memory.  Operands are staged K-major in float64
//   (AT = Phi2^T, BT = (Phi1 C^T)^T) by two small pre-kernels so that every LDS fragment
//   read is a conflict-free ds_read_b64.
//
// Reference arithmetic reproduced (oracle/dm_oracle.py: fm_to_p2p, indicator_argmax):
//   knn21[i] = argmin_j  n1_j - 2 G_ij      n1_j = |C Phi1_j|^2      pyFM/spectral/convert.py:138-140
//   knn12[j] = argmin_i  n2_i - 2 G_ij      n2_i = |Phi2_i C|^2      pyFM/spectral/convert.py:134-136
//   ind21[i] = argmax_j  G_ij a1_j                                   convert.py:144, functional_map.py:49
//   ind12[j] = argmax_i  G_ij a1_j                                   convert.py:144, functional_map.py:50
// lowest index on ties (NumPy argmax/argmin).
#include <stdlib.h>

#include "dm_gemm_f64.h"
#include "dm_internal.h"
#include "dm_split.h"
#include "dm_indicator_dev.h"

static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

// =================================================================================================
// K-major float64 copy of Phi:  out[b][c][i] = Phi[b][i][c]
// =================================================================================================
// amax (nullable, zeroed by the caller): DM_NCH partial maxima of |Phi[:, :k]| per pair, kept as the bit patterns of
// non-negative doubles (which order like unsigned integers) so that workgroups can combine them with atomicMax
template <typename Tin>
__global__ __launch_bounds__(256) void phiT_kernel(const Tin* __restrict__ Phi, int N, int k, int ld,
                                                   double* __restrict__ out, int kpad, int Npad, double* __restrict__ amax) {
    __shared__ Tin tile[64][65];
    const int b = blockIdx.z;
    const int i0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const Tin* P = Phi + (long long)b * N * ld;
    double* O = out + (long long)b * kpad * Npad;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    double m = 0.0;
    for (int r = ty; r < 64; r += 4) {
        const int i = i0 + r, c = c0 + tx;
        const Tin v = (i < N && c < k) ? P[(long long)i * ld + c] : (Tin)0;
        tile[r][tx] = v;
        m = fmax(m, fabs((double)v));
    }
    if (amax) {                                           // uniform
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
        if (tx == 0 && m > 0.0)
            atomicMax(reinterpret_cast<unsigned long long*>(amax) + (long long)b * DM_NCH + (blockIdx.x + blockIdx.y * gridDim.x) % DM_NCH,
                      (unsigned long long)__double_as_longlong(m));
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, i = i0 + tx;
        if (c < kpad && i < Npad) O[(long long)c * Npad + i] = (double)tile[tx][r];
    }
}

template <typename TR>
int dm_launch_phiT(dm_ctx* ctx, int B, int N, int k, const TR* Phi, int ld, double* out, int kpad, int Npad, double* amax) {
    dim3 grid(dm_cdiv(Npad, 64), dm_cdiv(kpad, 64), B);
    DM_LAUNCH(ctx, "phiT", phiT_kernel<TR>, grid, dim3(256), 0, Phi, N, k, ld, out, kpad, Npad, amax);
    return DM_OK;
}
template int dm_launch_phiT<float>(dm_ctx*, int, int, int, const float*, int, double*, int, int, double*);
template int dm_launch_phiT<double>(dm_ctx*, int, int, int, const double*, int, double*, int, int, double*);

// fp32 lumped masses -> float64 (exact), once per call: every kernel of the library reads masses as float64
__global__ __launch_bounds__(256) void widen_mass_kernel(const float* __restrict__ m, long long n, double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (double)m[i];
}
int dm_widen_mass(dm_ctx* ctx, long long n, const float* mass, double* buf, const double** out) {
    *out = nullptr;
    if (!mass) return DM_OK;
    if (!buf) return dm_fail(ctx, DM_ENOMEM, "widen_mass: workspace not reserved");
    DM_LAUNCH(ctx, "widen_mass", widen_mass_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mass, n, buf);
    *out = buf;
    return DM_OK;
}
int dm_widen_mass(dm_ctx*, long long, const double* mass, double*, const double** out) { *out = mass; return DM_OK; }

static int launch_transpose_f64(dm_ctx* ctx, int B, int N, int k, const double* X, int ld, double* out, int kpad, int Npad) {
    return dm_launch_phiT<double>(ctx, B, N, k, X, ld, out, kpad, Npad, nullptr);
}

// =================================================================================================
// embT[b][r][j] = sum_m Cm[b][r][m] Phi[b][j][m]; column norms
// =================================================================================================
struct OutKMajor {
    double* p; long long stride_b; int ld;
    __device__ __forceinline__ void store(int b, int i, int j, double v) const {
        p[b * stride_b + (long long)i * ld + j] = v;
    }
};

// nrm[b][j] = sum_r embT[b][r][j]^2 over r < kr (fixed ascending order: deterministic);
// amax_part[b * gridDim.x + blockIdx.x] = max |embT| over the workgroup's 256 columns (nullable; dm_knnsplit.hip scales by it)
__global__ __launch_bounds__(256) void colnorm_kernel(const double* __restrict__ embT, int kr, int krpad, int Npad,
                                                      double* __restrict__ nrm, double* __restrict__ amax_part) {
    __shared__ double wmax[4];
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const double* E = embT + (long long)b * krpad * Npad;
    double s = 0.0, m = 0.0;
    if (j < Npad) {
        int r = 0;
        for (; r + 16 <= kr; r += 16) {              // 16 loads in flight, summed in ascending order
            double x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = E[(long long)(r + u) * Npad + j];
#pragma unroll
            for (int u = 0; u < 16; ++u) { s += x[u] * x[u]; m = fmax(m, fabs(x[u])); }
        }
        for (; r < kr; ++r) {
            const double x = E[(long long)r * Npad + j];
            s += x * x;
            m = fmax(m, fabs(x));
        }
        nrm[(long long)b * Npad + j] = s;
    }
    if (amax_part) {                                 // uniform
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) amax_part[b * gridDim.x + blockIdx.x] = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
    }
}

// embT (K-major, optional) + column norms + per-64-column maxima in one kernel.  A workgroup tile is ALL rows of the
// product x 64 vertices (walked in groups of 64 RT rows), so the squared sums never leave the workgroup and are added in
// a fixed order (identical columns get identical norms).  Each wave owns 16 RT rows x 64 columns = RT x 4 accumulator tiles of
// v_mfma_f64_16x16x4_f64.  The contraction runs in stages of 16 through two LDS buffers (one barrier per stage), operands
// staged through registers; workgroups are persistent and fetch the first stage of their NEXT tile during the last stage
// of the current one, so no load latency is exposed between tiles (the contraction is only 4-13 stages deep).
constexpr int EB_BK = 16;     // contraction per stage
constexpr int EB_LD = 18;     // LDS row stride (f64): 36 dwords -> the 16 rows of a fragment read start on distinct 4-bank groups
static inline size_t embed_lds(int RT) { return ((size_t)2 * (64 * RT + 64) * EB_LD + 3 * 64 + 8) * sizeof(double); }

// FX: the basis rows that stream by also leave the kernel as split fp16 rows of the tile kernels (dm_embed_fx: what fs_build_rows_kernel
// writes in a pass of its own), scaled with the power of two the HINTED maxima give; the thread that stages entries k .. k + 3 of a
// vertex's row for the matrix cores splits them and stores 8 + 8 bytes (four threads = one vertex's 64 bytes of a 16-index group)
template <int RT, bool STORE, typename TR, bool FX = false>
__global__ __launch_bounds__(256, 2) void embed_tile_kernel(const double* __restrict__ Cm, long long strideC, int ldc, int transC,
                                                         const TR* __restrict__ Phi, long long stridePhi, int ld,
                                                         double* __restrict__ embT, int krpad, int Npad, int kr, int N, int K,
                                                         double* __restrict__ nrm, double* __restrict__ amax_part, int ntile_j,
                                                         int total, int nrg, double* __restrict__ amax_in_part, dm_embed_fx fx) {
    extern __shared__ __attribute__((aligned(16))) double eb_sm[];
    constexpr int RA = 64 * RT;
    double* Abuf = eb_sm;                                   // [2][RA][EB_LD]
    double* Bbuf = eb_sm + 2 * RA * EB_LD;                  // [2][64][EB_LD]
    double* xs = Bbuf + 2 * 64 * EB_LD;                     // [3][64] column sums of waves 1-3
    double* wmax = xs + 3 * 64;                             // [4] output maxima | [4] input maxima
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int srow = t >> 2, sk = (t & 3) * 4;             // staging: row srow (+ 64 q), contraction entries sk .. sk+3
    const int ns = (K + EB_BK - 1) / EB_BK;
    const bool fastA = !transC && ((ldc & 1) == 0) && ((strideC & 1) == 0) && ((((uintptr_t)Cm) & 15) == 0);
    constexpr int amask = sizeof(TR) == 4 ? 3 : 1;       // 16-byte loads of the basis rows
    const bool fastB = ((ld & amask) == 0) && ((stridePhi & amask) == 0) && ((((uintptr_t)Phi) & 15) == 0);

    // staged operands: kept exactly as loaded (the fp32 -> fp64 conversion happens when they are written to LDS one stage
    // later: converting at load time would make every fetch wait for its own data); pointers carry the global address
    // space so that the loads are global_load, not flat_load (a flat load also counts on lgkmcnt and would be waited for
    // by the fragment reads' lgkmcnt(0))
    typedef __attribute__((address_space(1))) const double gdouble;
    typedef __attribute__((address_space(1))) const TR gfloat;
    typedef __attribute__((address_space(1))) const f64x2 gf64x2;
    typedef __attribute__((address_space(1))) const f32x4 gf32x4;
    double ra[RT][4];
    TR rb[4];
    // The operand fetch is a stream of stages that runs one stage ahead of the compute loop, across tile boundaries.  Its
    // position is kept as running pointers (one 64-bit add per operand row and stage: f64 MFMA shares the vector ALU, every
    // address instruction in the loop is paid in full).  Rows >= kr and vertices >= N only feed accumulator entries that
    // the epilogue masks, so their loads are clamped to a valid address instead of being predicated; a stage that lies
    // completely inside the contraction (wave-uniform test) is then branch-free.  The generic path handles a ragged last
    // stage and unaligned operands.
    int f_tile = blockIdx.x, f_s = 0, f_rg = 0;           // fetch stream position: tile, row group of the tile, stage
    double fx_hint_next = 0.0;
    gdouble* fa[RT];
    gfloat* fb = nullptr;
    const long long a_step = transC ? (long long)EB_BK * ldc : EB_BK;
    // (macros, not lambdas: by-reference lambda captures of the staging arrays end up in scratch)
#define EB_SET_TILE()                                                                                                  \
    {                                                                                                                  \
        const int b_ = f_tile / ntile_j, j0_ = (f_tile - b_ * ntile_j) * 64;                                           \
        gdouble* cb_ = (gdouble*)Cm + (long long)b_ * strideC;                                                         \
        _Pragma("unroll") for (int q = 0; q < RT; ++q) {                                                               \
            const int row_ = min(f_rg * RA + srow + 64 * q, kr - 1);                                                   \
            fa[q] = cb_ + (transC ? (long long)sk * ldc + row_ : (long long)row_ * ldc + sk);                          \
        }                                                                                                              \
        fb = (gfloat*)Phi + (long long)b_ * stridePhi + (long long)min(j0_ + srow, N - 1) * ld + sk;                   \
        f_s = 0;                                                                                                       \
        if constexpr (FX) fx_hint_next = fx.hint[b_];      /* the pair's hinted maximum, a tile ahead of its use */     \
    }
#define EB_FETCH()                                                                                                     \
    {                                                                                                                  \
        if ((f_s + 1) * EB_BK <= K && fastB && (fastA || transC)) {                                                    \
            _Pragma("unroll") for (int q = 0; q < RT; ++q) {                                                           \
                if (!transC) {                                                                                         \
                    const f64x2 x0_ = ((gf64x2*)fa[q])[0], x1_ = ((gf64x2*)fa[q])[1];                                  \
                    ra[q][0] = x0_[0]; ra[q][1] = x0_[1]; ra[q][2] = x1_[0]; ra[q][3] = x1_[1];                        \
                } else {                                                                                               \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) ra[q][e] = fa[q][(long long)e * ldc];                \
                }                                                                                                      \
            }                                                                                                          \
            if constexpr (sizeof(TR) == 4) {                                                                           \
                const f32x4 x_ = *(gf32x4*)fb;                                                                         \
                rb[0] = x_[0]; rb[1] = x_[1]; rb[2] = x_[2]; rb[3] = x_[3];                                            \
            } else {                                                                                                   \
                const f64x2 y0_ = ((gf64x2*)fb)[0], y1_ = ((gf64x2*)fb)[1];                                            \
                rb[0] = y0_[0]; rb[1] = y0_[1]; rb[2] = y1_[0]; rb[3] = y1_[1];                                        \
            }                                                                                                          \
        } else {                                                                                                       \
            const int k0_ = f_s * EB_BK + sk;                                                                          \
            _Pragma("unroll") for (int q = 0; q < RT; ++q)                                                             \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                          \
                    ra[q][e] = (k0_ + e < K) ? (transC ? fa[q][(long long)e * ldc] : fa[q][e]) : 0.0;                  \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) rb[e] = (k0_ + e < K) ? fb[e] : (TR)0;                       \
        }                                                                                                              \
        _Pragma("unroll") for (int q = 0; q < RT; ++q) fa[q] += a_step;                                                \
        fb += EB_BK;                                                                                                   \
        if (++f_s == ns) {                                                                                             \
            if (++f_rg == nrg) { f_rg = 0; f_tile += (int)gridDim.x; }                                                 \
            if (f_tile < total) EB_SET_TILE()                                                                          \
        }                                                                                                              \
    }

    int tile = blockIdx.x;
    if (tile >= total) return;
    EB_SET_TILE()
    EB_FETCH()
    int g = 0;                                              // stage counter across tiles: LDS buffer = g & 1
    while (tile < total) {
        const int b = tile / ntile_j, j0 = (tile - b * ntile_j) * 64;
        double* E = STORE ? embT + (long long)b * krpad * Npad : nullptr;
        double csum[4] = {0.0, 0.0, 0.0, 0.0}, amax = 0.0, inmax = 0.0;   // inmax: max |Phi| over the tile's vertex slab
        double fx_sx = 1.0;
        _Float16* fx_row = nullptr;
        if constexpr (FX) {
            // (the fetch stream's last EB_SET_TILE before this point was the one for THIS tile -- it ran during the previous tile's last
            //  stage --; the next one comes during this tile's last stage, after the value has been used)
            fx_sx = ks_scale(&fx_hint_next, 1);
            if (j0 + srow < N) fx_row = fx.F + ((long long)b * fx.rows_out + j0 + srow) * fx.D + sk;
        }
        // more than 64 RT rows: the tile is walked in row groups (the vertex slab is streamed again from L2 for each), the
        // column sums run on across the groups
        for (int rg = 0; rg < nrg; ++rg) {
            f64x4 acc[RT][4];
#pragma unroll
            for (int a_ = 0; a_ < RT; ++a_)
#pragma unroll
                for (int c_ = 0; c_ < 4; ++c_) acc[a_][c_] = f64x4{0.0, 0.0, 0.0, 0.0};
            for (int s = 0; s < ns; ++s, ++g) {
                double* As = Abuf + (g & 1) * RA * EB_LD;
                double* Bs = Bbuf + (g & 1) * 64 * EB_LD;
#pragma unroll
                for (int q = 0; q < RT; ++q) {
                    *reinterpret_cast<f64x2*>(As + (srow + 64 * q) * EB_LD + sk) = f64x2{ra[q][0], ra[q][1]};
                    *reinterpret_cast<f64x2*>(As + (srow + 64 * q) * EB_LD + sk + 2) = f64x2{ra[q][2], ra[q][3]};
                }
                *reinterpret_cast<f64x2*>(Bs + srow * EB_LD + sk) = f64x2{(double)rb[0], (double)rb[1]};
                *reinterpret_cast<f64x2*>(Bs + srow * EB_LD + sk + 2) = f64x2{(double)rb[2], (double)rb[3]};
                if (amax_in_part)                           // (uniform) the basis entries this stage consumes, as they pass by
                    inmax = fmax(fmax(inmax, fmax(fabs((double)rb[0]), fabs((double)rb[1]))), fmax(fabs((double)rb[2]), fabs((double)rb[3])));
                if constexpr (FX) {
                    if (rg == 0 && fx_row) {                // (entries beyond K were fetched as 0: zero halves, like the row builder's)
                        typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
                        f16x4_t hv, lv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { _Float16 h_, l_; fs_split_entry<TR>(rb[e], fx_sx, h_, l_); hv[e] = h_; lv[e] = l_; }
                        *reinterpret_cast<f16x4_t*>(fx_row + 32 * s) = hv;
                        *reinterpret_cast<f16x4_t*>(fx_row + 32 * s + 16) = lv;
                    }
                }
                __syncthreads();
                // the buffer written above was last read two stages ago, and every wave has passed a barrier since
                if (f_tile < total) EB_FETCH()
#pragma unroll
                for (int ks = 0; ks < EB_BK / 4; ++ks) {
                    const int kk = ks * 4 + (lane >> 4);
                    double av[RT], bv[4];
#pragma unroll
                    for (int a_ = 0; a_ < RT; ++a_) av[a_] = As[(wave * 16 * RT + a_ * 16 + (lane & 15)) * EB_LD + kk];
#pragma unroll
                    for (int c_ = 0; c_ < 4; ++c_) bv[c_] = Bs[(c_ * 16 + (lane & 15)) * EB_LD + kk];
#pragma unroll
                    for (int a_ = 0; a_ < RT; ++a_)
                        if (rg * RA + wave * 16 * RT + a_ * 16 < kr) {     // (wave-uniform: row blocks beyond kr are skipped)
#pragma unroll
                            for (int c_ = 0; c_ < 4; ++c_) acc[a_][c_] = mfma_f64_16x16x4(av[a_], bv[c_], acc[a_][c_]);
                        }
                }
            }
            // ---- row-group epilogue: optional K-major store, squares into the column sums (fixed order), maxima
#pragma unroll
            for (int c_ = 0; c_ < 4; ++c_) {
                const int j = j0 + c_ * 16 + (lane & 15);
#pragma unroll
                for (int a_ = 0; a_ < RT; ++a_)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = rg * RA + wave * 16 * RT + a_ * 16 + (lane >> 4) + 4 * r;
                        const double v = (i < kr && j < N) ? acc[a_][c_][r] : 0.0;
                        if (STORE && i < kr && j < N) E[(long long)i * Npad + j] = v;
                        csum[c_] = fma(v, v, csum[c_]);
                        amax = fmax(amax, fabs(v));
                    }
            }
        }
        // ---- tile epilogue: the four row groups of a wave (xor tree: the same order in every lane), then the four waves
#pragma unroll
        for (int c_ = 0; c_ < 4; ++c_) {
            csum[c_] += __shfl_xor(csum[c_], 16);
            csum[c_] += __shfl_xor(csum[c_], 32);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off));
        if (wave > 0 && lane < 16) {
#pragma unroll
            for (int c_ = 0; c_ < 4; ++c_) xs[(wave - 1) * 64 + c_ * 16 + lane] = csum[c_];
        }
        if (amax_in_part) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) inmax = fmax(inmax, __shfl_xor(inmax, off));
        }
        if (lane == 0) { wmax[wave] = amax; wmax[4 + wave] = inmax; }
        __syncthreads();
        if (wave == 0 && lane < 16 && nrm) {
#pragma unroll
            for (int c_ = 0; c_ < 4; ++c_) {
                const int cj = c_ * 16 + lane, j = j0 + cj;
                if (j < Npad) nrm[(long long)b * Npad + j] = ((csum[c_] + xs[cj]) + xs[64 + cj]) + xs[128 + cj];
            }
        }
        if (t == 0 && amax_part) amax_part[(long long)b * ntile_j + (j0 >> 6)] = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
        if (t == 0 && amax_in_part) {
            const double im = fmax(fmax(wmax[4], wmax[5]), fmax(wmax[6], wmax[7]));
            amax_in_part[(long long)b * ntile_j + (j0 >> 6)] = im;
            if (fx.amax_copy) fx.amax_copy[(long long)b * ntile_j + (j0 >> 6)] = im;         // (the next call's hint)
        }
        // (xs / wmax are rewritten only after the next tile's stage barriers)
        tile += (int)gridDim.x;
    }
#undef EB_FETCH
#undef EB_SET_TILE
}

template <typename TR>
int dm_launch_embed(dm_ctx* ctx, int B, int N, int kr, int km, const TR* Phi, int ld, const double* Cm, int ldc,
                    long long strideC, int transC, double* embT, int krpad, int Npad, double* nrm, int zero_first,
                    double* amax_part, double* amax_in_part, const dm_embed_fx* fxp) {
    dm_embed_fx fx{nullptr, 0, 0, nullptr, 0, nullptr};
    if (fxp) fx = *fxp;
    if (fx.F && (embT || !amax_in_part || !fx.hint)) return dm_fail(ctx, DM_EINVAL, "embed: the split-row output rides on the norms-only embedding");
    // the K-major buffer is zero padded: rows >= kr and columns >= N must be 0 for the tile kernels
    if (embT && zero_first && N == Npad && kr != krpad) {
        // only the padding ROWS kr .. krpad - 1 of every pair (config 5: k = 200 in a 208-row buffer -- the whole buffer is 872 MB, a
        // 150 us memset per call; its padding 33 MB)
        DM_CHECK_HIP(ctx, hipMemset2DAsync(embT + (size_t)kr * Npad, (size_t)krpad * Npad * sizeof(double), 0, (size_t)(krpad - kr) * Npad * sizeof(double),
                                           (size_t)B, ctx->stream));
    } else if (embT && zero_first && (kr != krpad || N != Npad)) {
        DM_CHECK_HIP(ctx, hipMemsetAsync(embT, 0, (size_t)B * krpad * Npad * sizeof(double), ctx->stream));
    }
    const int RT = kr <= 64 ? 1 : 2;                        // 64 RT rows per row group: two workgroups per CU at RT = 2
    const int nrg = dm_cdiv(kr, 64 * RT);
    const int ntile_j = dm_cdiv(Npad, DM_EMB_COLS), total = B * ntile_j;
    const size_t lds = embed_lds(RT);
    const int ncu = ctx->n_cu > 0 ? ctx->n_cu : 256;
    // resident workgroups per CU: registers (RT x 32 accumulator registers per lane) and LDS
    const int by_regs = RT == 1 ? 3 : 2, by_lds = (int)((size_t)160 * 1024 / lds);
    const int per_cu = by_regs < by_lds ? by_regs : by_lds;
    const int grid = total < ncu * per_cu ? total : ncu * per_cu;
#define EB_LAUNCH1(RT_, ST_, FX_)                                                                                      \
    {                                                                                                                  \
        int rc = dm_grant_lds(ctx, (const void*)embed_tile_kernel<RT_, ST_, TR, FX_>, lds);                            \
        if (rc) return rc;                                                                                             \
        DM_LAUNCH(ctx, "embed_nt_f64", (embed_tile_kernel<RT_, ST_, TR, FX_>), dim3(grid), dim3(256), lds, Cm, strideC, ldc, transC, Phi, \
                  (long long)N * ld, ld, embT, krpad, Npad, kr, N, km, nrm, amax_part, ntile_j, total, nrg, amax_in_part, fx); \
    }
#define EB_LAUNCH(RT_)                                                                                                 \
    {                                                                                                                  \
        if (embT) EB_LAUNCH1(RT_, true, false)                                                                         \
        else if (fx.F) EB_LAUNCH1(RT_, false, true)                                                                    \
        else EB_LAUNCH1(RT_, false, false)                                                                             \
    }
    if (RT == 1) EB_LAUNCH(1) else EB_LAUNCH(2)
#undef EB_LAUNCH
#undef EB_LAUNCH1
    return DM_OK;
}
template int dm_launch_embed<float>(dm_ctx*, int, int, int, int, const float*, int, const double*, int, long long, int, double*, int, int,
                                    double*, int, double*, double*, const dm_embed_fx*);
template int dm_launch_embed<double>(dm_ctx*, int, int, int, int, const double*, int, const double*, int, long long, int, double*, int,
                                     int, double*, int, double*, double*, const dm_embed_fx*);

// =================================================================================================
// fused G tile + arg-reductions
// =================================================================================================
constexpr int GT = 128;    // G tile (rows and columns) per workgroup
constexpr int GBK = 16;    // contraction depth per LDS stage
constexpr int GLD = 144;   // LDS row stride (f64): 288 dwords = 32 (mod 64)
constexpr int GX_WAVE = 64 * 17 + 64 * 17 / 2;   // epilogue exchange buffer per wave (doubles): 64 x 17 values + 64 x 17 indices
static_assert((4 * 2 * 128 + 4 * 2 * 128 / 2 + 4 * GX_WAVE) <= 2 * 2 * GBK * GLD, "epilogue scratch must fit in the staging LDS");

struct gred_params {
    const double* AT; const double* BT;
    const double* n1; const double* n2; const double* mass1;
    // per-tile partial results
    double* rv_knn; int32_t* rj_knn; double* rv_ind; int32_t* rj_ind;   // (B, tilesN, N2pad)
    double* cv_knn; int32_t* ci_knn; double* cv_ind; int32_t* ci_ind;   // (B, tilesM, N1pad)
    int N2, N1, N2pad, N1pad, Kpad, Kloop, tilesM, tilesN, total;
    int prio;     // raise the wave priority during the MFMA main loop
    int stagger;  // number of s_sleep(127) the odd-slot workgroups of the first round wait (0 = off)
    int dbg;      // experiments only (env DM_GRED_DEBUG): 1 = skip the reduction epilogue, 2 = skip the MFMA main loop
};

// Reductions over the wave's 64x64 sub-tile, acc[mt][nt][r] = G[row = i0 + wm*64 + mt*16 + (lane>>4) + 4r]
// [col = j0 + wn*64 + nt*16 + (lane&15)].  MASK = the tile crosses the matrix boundary (padded rows / columns must
// not win); interior tiles take the mask-free instantiation.  ALL = all four reductions (dm_fm_to_p2p; n1, n2 and
// mass1 are then always valid pointers), otherwise only the row arg-min knn21 (ZoomOut).  No run-time flags inside:
// the whole epilogue is one basic block per instantiation.
//  rows:    two passes per row slot: (A) extreme VALUE only (one v_max/v_min per element, DPP butterflies across
//           the 16 lanes of a row), (B) lowest column whose value equals that extreme (32-bit min): the NumPy
//           first-index rule without carrying (value, index) pairs through the butterfly.
//  columns: (value, row) pairs tracked in-lane while sweeping the rows in ascending order, merged across the four
//           16-lane groups at the end.
template <bool MASK, bool ALL>
__device__ __forceinline__ void gred_epilogue(const gred_params& p, const f64x4 (&acc)[4][4], double* sv, int* sj,
                                              double* xch, int b, int i0, int j0, int lane, int wm, int wn) {
    const int cl = lane & 15, rg = lane >> 4;
    // ---- sweep 1: row reductions -------------------------------------------------------------------------
    // (f64 MFMA runs on the vector ALU's double-precision datapath on this part -- tools/ubench_coissue.hip: VALU work
    // of a co-resident wave ADDS to the MFMA time, nothing overlaps -- so the epilogue is priced per VALU instruction.)
    // Per kind: (A) every lane reduces its four columns of a row slot in registers, (value, lowest column) pairs;
    // (B) the 16 partial pairs of each row cross the wave through a private LDS exchange buffer, row-major with a
    // 17-entry row stride (conflict-free both ways); (C) lane L finishes row L: value tree, then the lowest column
    // among the partials that attain it.  About half the VALU instructions of DPP butterflies per row slot.
    {
        const int wave = wm * 2 + wn;
        double* xv = xch + wave * GX_WAVE;                         // [64 rows][17]
        int* xj = reinterpret_cast<int*>(xv + 64 * 17);            // [64 rows][17]
        double a1[4], n1c[4];
        int gcol[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int gj = j0 + wn * 64 + nt * 16 + cl;
            const bool cv = !MASK || gj < p.N1;
            gcol[nt] = cv ? gj : DM_IDX_NONE;
            a1[nt] = (ALL && cv) ? (double)p.mass1[(long long)b * p.N1 + gj] : 0.0;
            n1c[nt] = p.n1[(long long)b * p.N1pad + gj];
        }
#pragma unroll
        for (int kind = (ALL ? 0 : 1); kind < 2; ++kind) {         // 0: ind21 (arg-max of G a1), 1: knn21 (arg-min)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const double g = acc[mt][nt][r];
                        // arg-min of (|x|^2 - 2 <x,y>) is carried as the arg-max of its negation's ordering:
                        // the comparisons below are written per kind, the values are the reference's own
                        v[nt] = (kind == 0) ? g * a1[nt]                        // (Phi2 C Phi1^T) @ A1, convert.py:144
                                            : n1c[nt] - 2.0 * g;                // |x|^2 - 2 <x,y>
                        if (MASK) {
                            const bool cv = gcol[nt] != DM_IDX_NONE;
                            v[nt] = cv ? v[nt] : ((kind == 0) ? -DM_INF_F64 : DM_INF_F64);
                        }
                    }
                    double m01, m23, m;
                    int j01, j23, j;
                    if (kind == 0) {
                        m01 = fmax(v[0], v[1]); j01 = (v[1] > v[0]) ? gcol[1] : gcol[0];
                        m23 = fmax(v[2], v[3]); j23 = (v[3] > v[2]) ? gcol[3] : gcol[2];
                        m = fmax(m01, m23); j = (m23 > m01) ? j23 : j01;
                    } else {
                        m01 = fmin(v[0], v[1]); j01 = (v[1] < v[0]) ? gcol[1] : gcol[0];
                        m23 = fmin(v[2], v[3]); j23 = (v[3] < v[2]) ? gcol[3] : gcol[2];
                        m = fmin(m01, m23); j = (m23 < m01) ? j23 : j01;
                    }
                    const int row = mt * 16 + rg + 4 * r;
                    xv[row * 17 + cl] = m;
                    xj[row * 17 + cl] = j;
                }
            }
            // (same wave wrote what it reads: the LDS queue of a wave is in order, no barrier)
            double w[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) w[c] = xv[lane * 17 + c];
            double m8[8], m4[4];
#pragma unroll
            for (int c = 0; c < 8; ++c) m8[c] = (kind == 0) ? fmax(w[2 * c], w[2 * c + 1]) : fmin(w[2 * c], w[2 * c + 1]);
#pragma unroll
            for (int c = 0; c < 4; ++c) m4[c] = (kind == 0) ? fmax(m8[2 * c], m8[2 * c + 1]) : fmin(m8[2 * c], m8[2 * c + 1]);
            const double ma = (kind == 0) ? fmax(m4[0], m4[1]) : fmin(m4[0], m4[1]);
            const double mb = (kind == 0) ? fmax(m4[2], m4[3]) : fmin(m4[2], m4[3]);
            const double mrow = (kind == 0) ? fmax(ma, mb) : fmin(ma, mb);
            int jrow = DM_IDX_NONE;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int jc = xj[lane * 17 + c];
                jrow = min(jrow, (w[c] == mrow) ? jc : DM_IDX_NONE);
            }
            sv[(kind * 2 + wn) * 128 + wm * 64 + lane] = mrow;
            sj[(kind * 2 + wn) * 128 + wm * 64 + lane] = jrow;
        }
    }
    if (!ALL) return;
    // ---- sweeps 2a / 2b: column reductions --------------------------------------------------------------------
    // 2a: extreme VALUE per column over this lane's 16 rows, then across the four 16-lane groups;
    // 2b: lowest row whose value equals it.  The values are recomputed with the same two operations as before
    // (bit-identical) instead of being kept alive: the opaque asm statements stop the optimiser from re-using
    // (and spilling) the 64 products of the previous sweep.
    {
        double a1[4], cmax_i[4], cmin_k[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int gj = j0 + wn * 64 + nt * 16 + cl;
            a1[nt] = (!MASK || gj < p.N1) ? (double)p.mass1[(long long)b * p.N1 + gj] : 0.0;
            asm volatile("" : "+v"(a1[nt]));
            cmax_i[nt] = -DM_INF_F64; cmin_k[nt] = DM_INF_F64;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = i0 + wm * 64 + mt * 16 + rg + 4 * r;
                const bool rvalid = !MASK || gi < p.N2;
                const double n2r = p.n2[(long long)b * p.N2pad + gi];               // padded buffer: always in bounds
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const double g = acc[mt][nt][r];
                    double vic = g * a1[nt];
                    double vk2 = n2r - 2.0 * g;
                    if (MASK) {
                        vic = rvalid ? vic : -DM_INF_F64;
                        vk2 = rvalid ? vk2 : DM_INF_F64;
                    }
                    cmax_i[nt] = fmax(cmax_i[nt], vic);
                    cmin_k[nt] = fmin(cmin_k[nt], vk2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                cmax_i[nt] = fmax(cmax_i[nt], __shfl_xor(cmax_i[nt], off));
                cmin_k[nt] = fmin(cmin_k[nt], __shfl_xor(cmin_k[nt], off));
            }
            asm volatile("" : "+v"(a1[nt]));
        }
        int ci[4], ck[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { ci[nt] = DM_IDX_NONE; ck[nt] = DM_IDX_NONE; }
#pragma unroll
        for (int mt = 3; mt >= 0; --mt) {
#pragma unroll
            for (int r = 3; r >= 0; --r) {                      // descending rows: the lowest matching row is kept
                const int gi = i0 + wm * 64 + mt * 16 + rg + 4 * r;
                const bool rvalid = !MASK || gi < p.N2;
                const double n2r = p.n2[(long long)b * p.N2pad + gi];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const double g = acc[mt][nt][r];
                    const double vic = g * a1[nt];
                    const double vk2 = n2r - 2.0 * g;
                    ci[nt] = (rvalid && vic == cmax_i[nt]) ? gi : ci[nt];
                    ck[nt] = (rvalid && vk2 == cmin_k[nt]) ? gi : ck[nt];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                ci[nt] = min(ci[nt], __shfl_xor(ci[nt], off));
                ck[nt] = min(ck[nt], __shfl_xor(ck[nt], off));
            }
            if (lane < 16) {
                const int lcol = wn * 64 + nt * 16 + lane;
                sv[(2 * 2 + wm) * 128 + lcol] = cmax_i[nt]; sj[(2 * 2 + wm) * 128 + lcol] = ci[nt];
                sv[(3 * 2 + wm) * 128 + lcol] = cmin_k[nt]; sj[(3 * 2 + wm) * 128 + lcol] = ck[nt];
            }
        }
    }
}

template <bool ALL>
__global__ __launch_bounds__(256, 2) void gred_kernel(gred_params p) {
    __shared__ double smem[2 * 2 * GBK * GLD];   // As[2][GBK][GLD] | Bs[2][GBK][GLD]   (73,728 B)
    double* As = smem;
    double* Bs = smem + 2 * GBK * GLD;

    const int id = xcd_remap(blockIdx.x, p.total);
    const int tiles = p.tilesM * p.tilesN;
    const int b = id / tiles;
    const int tmn = id - b * tiles;
    // tile order inside a pair: bands of 8 tile rows, column-major inside a band, so the ~64 tiles that are resident
    // on an XCD at one time form an 8 x 8 block and share 16 operand panels (3.4 MB at k = 200) instead of one row's
    // 1 + 64 panels -- at N = 8192 the row order re-fetched the B panels from HBM for every tile row (PMC: 57 GB/launch)
    const int band = tmn / (8 * p.tilesN);
    const int brow0 = band * 8, brows = min(8, p.tilesM - brow0);
    const int brem = tmn - band * 8 * p.tilesN;
    const int tn = brem / brows, tm = brow0 + (brem - tn * brows);
    const int i0 = tm * GT, j0 = tn * GT;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // Two workgroups share a CU so that one computes while the other waits for memory.  (Experiment knob, off by
    // default: delay the second resident workgroup of the first dispatch round by half a tile period so the two stay
    // in opposite phases.  It bought 6 % while the epilogue was DPP-bound and nothing since: f64 MFMA and VALU work
    // share the double-precision datapath on this part and cannot overlap, tools/ubench_coissue.hip.)
    if (p.stagger && blockIdx.x < 512) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;      // HW_ID.wave_id bit 0
        if (slot) {
            for (int q = 0; q < p.stagger; ++q) __builtin_amdgcn_s_sleep(127);
        }
    }

    const double* AT = p.AT + (long long)b * p.Kpad * p.N2pad + i0;
    const double* BT = p.BT + (long long)b * p.Kpad * p.N1pad + j0;

    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f64x4{0.0, 0.0, 0.0, 0.0};

    // stage = GBK rows x 128 f64 per operand; pass q: row q*4 + t/64, 16 B per lane, one 1 KiB row per wave
    const int srow = t >> 6, scol = (t & 63) * 2;
    f64x2 ra[4], rb[4];
    // (macros, not lambdas: by-reference lambda captures of the staging arrays end up in scratch)
#define GRED_FETCH(s_)                                                                               \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                  \
        const int kr = (s_) * GBK + q * 4 + srow;                                                    \
        ra[q] = *reinterpret_cast<const f64x2*>(AT + (long long)kr * p.N2pad + scol);              \
        rb[q] = *reinterpret_cast<const f64x2*>(BT + (long long)kr * p.N1pad + scol);              \
    }
#define GRED_STASH(buf_)                                                                             \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                  \
        const int off = (buf_) * GBK * GLD + (q * 4 + srow) * GLD + scol;                            \
        *reinterpret_cast<f64x2*>(As + off) = ra[q];                                               \
        *reinterpret_cast<f64x2*>(Bs + off) = rb[q];                                               \
    }

    const int ns = (p.dbg == 2) ? 1 : p.Kloop / GBK;
    // (experiment knob, off by default: the MFMA main loop outranks the co-resident workgroup's epilogue at the issue port)
    if (p.prio) __builtin_amdgcn_s_setprio(2);
    GRED_FETCH(0)
    GRED_STASH(0)
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        const int buf = s & 1;
        if (s + 1 < ns) { GRED_FETCH(s + 1) }
        const double* A_ = As + buf * GBK * GLD + wm * 64 + (lane & 15);
        const double* B_ = Bs + buf * GBK * GLD + wn * 64 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < GBK / 4; ++ks) {
            const int kk = ks * 4 + (lane >> 4);
            double a[4], bb[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a[mt] = A_[kk * GLD + mt * 16];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bb[nt] = B_[kk * GLD + nt * 16];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma_f64_16x16x4(a[mt], bb[nt], acc[mt][nt]);
        }
        if (s + 1 < ns) { GRED_STASH(buf ^ 1) }
        __syncthreads();
    }
#undef GRED_FETCH
#undef GRED_STASH
    if (p.prio) __builtin_amdgcn_s_setprio(0);

    if (p.dbg == 1) {
        double sacc = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) sacc += acc[a][c][0] + acc[a][c][1] + acc[a][c][2] + acc[a][c][3];
        if (sacc == 1.2345) p.rv_knn[0] = sacc;
        return;
    }
    // ---------------- epilogue: reductions over the 128x128 tile held in registers ----------------
    // LDS is free now (the loop ended with a barrier): reuse it for the cross-wave merges.
    double* sv = smem;                                            // [4 kinds][2 waves][128]
    int* sj = reinterpret_cast<int*>(smem + 4 * 2 * 128);         // [4 kinds][2 waves][128]
    double* xch = smem + 4 * 2 * 128 + 4 * 2 * 128 / 2;         // per-wave exchange buffers, GX_WAVE doubles each
    if ((i0 + GT <= p.N2) && (j0 + GT <= p.N1)) gred_epilogue<false, ALL>(p, acc, sv, sj, xch, b, i0, j0, lane, wm, wn);
    else gred_epilogue<true, ALL>(p, acc, sv, sj, xch, b, i0, j0, lane, wm, wn);
    __syncthreads();
    // merge the two waves that share a row (wn = 0, 1) / a column (wm = 0, 1); lower half first
    if (t < 128) {
        const int gi = i0 + t;
        if (gi < p.N2) {
            const long long o = ((long long)b * p.tilesN + tn) * p.N2pad + gi;
            if (ALL) {
                double v = sv[(0 * 2 + 0) * 128 + t]; int j = sj[(0 * 2 + 0) * 128 + t];
                argmax_merge(v, j, sv[(0 * 2 + 1) * 128 + t], sj[(0 * 2 + 1) * 128 + t]);
                p.rv_ind[o] = v; p.rj_ind[o] = j;
            }
            {
                double v = sv[(1 * 2 + 0) * 128 + t]; int j = sj[(1 * 2 + 0) * 128 + t];
                argmin_merge(v, j, sv[(1 * 2 + 1) * 128 + t], sj[(1 * 2 + 1) * 128 + t]);
                p.rv_knn[o] = v; p.rj_knn[o] = j;
            }
        }
    } else if (ALL) {
        const int c = t - 128;
        const int gj = j0 + c;
        if (gj < p.N1) {
            const long long o = ((long long)b * p.tilesM + tm) * p.N1pad + gj;
            {
                double v = sv[(2 * 2 + 0) * 128 + c]; int i = sj[(2 * 2 + 0) * 128 + c];
                argmax_merge(v, i, sv[(2 * 2 + 1) * 128 + c], sj[(2 * 2 + 1) * 128 + c]);
                p.cv_ind[o] = v; p.ci_ind[o] = i;
            }
            {
                double v = sv[(3 * 2 + 0) * 128 + c]; int i = sj[(3 * 2 + 0) * 128 + c];
                argmin_merge(v, i, sv[(3 * 2 + 1) * 128 + c], sj[(3 * 2 + 1) * 128 + c]);
                p.cv_knn[o] = v; p.ci_knn[o] = i;
            }
        }
    }
}

// final merge over tiles, ascending tile order, strict comparison (lower tile = lower index wins ties)
__global__ __launch_bounds__(256) void gred_merge_kernel(const double* __restrict__ pv, const int32_t* __restrict__ pj,
                                                         int ntiles, int n, int npad, int is_max,
                                                         int32_t* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = is_max ? -DM_INF_F64 : DM_INF_F64;
    int j = DM_IDX_NONE;
    for (int tIdx = 0; tIdx < ntiles; ++tIdx) {
        const long long o = ((long long)b * ntiles + tIdx) * npad + i;
        const double ov = pv[o];
        const int oj = pj[o];
        if (is_max) argmax_merge(v, j, ov, oj); else argmin_merge(v, j, ov, oj);
    }
    out[(long long)b * n + i] = (j == DM_IDX_NONE) ? 0 : j;
}


size_t dm_gred_ws_bytes(int B, int N2, int N1) {
    const int N2pad = pad_to(N2, GT), N1pad = pad_to(N1, GT);
    const size_t tilesM = N2pad / GT, tilesN = N1pad / GT;
    const size_t rows = (size_t)B * tilesN * N2pad, cols = (size_t)B * tilesM * N1pad;
    return 2 * (dm_align_up(rows * 8) + dm_align_up(rows * 4)) + 2 * (dm_align_up(cols * 8) + dm_align_up(cols * 4)) + 4096;
}

int dm_launch_gred(dm_ctx* ctx, const dm_gred_args& a) {
    gred_params p;
    memset(&p, 0, sizeof(p));
    p.AT = a.AT; p.BT = a.BT; p.n1 = a.n1; p.n2 = a.n2; p.mass1 = a.mass1;
    p.N2 = a.N2; p.N1 = a.N1; p.N2pad = a.N2pad; p.N1pad = a.N1pad; p.Kpad = a.Kpad; p.Kloop = a.Kloop;
    p.tilesM = a.N2pad / GT; p.tilesN = a.N1pad / GT;
    p.total = a.B * p.tilesM * p.tilesN;
    p.dbg = dm_knob("DM_GRED_DEBUG", 0);          // DM_EXPERIMENTS builds only (the product passes 0)
    p.stagger = dm_knob("DM_GRED_STAGGER", 0);
    p.prio = dm_knob("DM_GRED_PRIO", 0);
    // two instantiations: all four reductions (needs n1, n2, mass1) or the row arg-min knn21 alone (needs n1)
    const bool all = a.knn12 || a.ind21 || a.ind12;
    if (!a.n1 || (all && (!a.n2 || !a.mass1)))
        return dm_fail(ctx, DM_EINVAL, "gred: missing norms / mass for the requested reductions");
    if (a.N2pad % GT || a.N1pad % GT || a.Kpad % GBK || a.Kloop % GBK || a.Kloop < GBK || a.Kloop > a.Kpad)
        return dm_fail(ctx, DM_EINVAL, "gred: operand padding must be a multiple of the tile (%d) / stage (%d)", GT, GBK);
    const size_t rows = (size_t)a.B * p.tilesN * a.N2pad, cols = (size_t)a.B * p.tilesM * a.N1pad;
    p.rv_knn = (double*)dm_ws_take(ctx, rows * 8); p.rj_knn = (int32_t*)dm_ws_take(ctx, rows * 4);
    p.rv_ind = (double*)dm_ws_take(ctx, rows * 8); p.rj_ind = (int32_t*)dm_ws_take(ctx, rows * 4);
    p.cv_knn = (double*)dm_ws_take(ctx, cols * 8); p.ci_knn = (int32_t*)dm_ws_take(ctx, cols * 4);
    p.cv_ind = (double*)dm_ws_take(ctx, cols * 8); p.ci_ind = (int32_t*)dm_ws_take(ctx, cols * 4);
    if (!p.rv_knn || !p.rj_knn || !p.rv_ind || !p.rj_ind || !p.cv_knn || !p.ci_knn || !p.cv_ind || !p.ci_ind)
        return dm_fail(ctx, DM_ENOMEM, "gred: workspace not reserved");
    if (all) DM_LAUNCH(ctx, "gred_f64", gred_kernel<true>, dim3(p.total), dim3(256), 0, p);
    else DM_LAUNCH(ctx, "gred_f64", gred_kernel<false>, dim3(p.total), dim3(256), 0, p);
    if (a.knn21) {
        DM_LAUNCH(ctx, "gred_merge", gred_merge_kernel, dim3(dm_cdiv(a.N2, 256), a.B), dim3(256), 0, p.rv_knn, p.rj_knn,
                  p.tilesN, a.N2, a.N2pad, 0, a.knn21);
    }
    if (a.ind21) {
        DM_LAUNCH(ctx, "gred_merge", gred_merge_kernel, dim3(dm_cdiv(a.N2, 256), a.B), dim3(256), 0, p.rv_ind, p.rj_ind,
                  p.tilesN, a.N2, a.N2pad, 1, a.ind21);
    }
    if (a.knn12) {
        DM_LAUNCH(ctx, "gred_merge", gred_merge_kernel, dim3(dm_cdiv(a.N1, 256), a.B), dim3(256), 0, p.cv_knn, p.ci_knn,
                  p.tilesM, a.N1, a.N1pad, 0, a.knn12);
    }
    if (a.ind12) {
        DM_LAUNCH(ctx, "gred_merge", gred_merge_kernel, dim3(dm_cdiv(a.N1, 256), a.B), dim3(256), 0, p.cv_ind, p.ci_ind,
                  p.tilesM, a.N1, a.N1pad, 1, a.ind12);
    }
    return DM_OK;
}

// =================================================================================================
// C ABI
// =================================================================================================
template <typename TR>
static int fm_to_p2p_impl(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const TR* Phi1, int ld1,
                          const TR* Phi2, int ld2, const TR* mass1_in, const double* C, int32_t* knn21,
                          int32_t* knn12, int32_t* ind21, int32_t* ind12) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N1 > 0 && N2 > 0 && k1 > 0 && k2 > 0, "sizes must be positive");
    DM_REQUIRE(ctx, Phi1 && Phi2 && C, "null input");
    DM_REQUIRE(ctx, ld1 >= k1 && ld2 >= k2, "eigenvector row stride smaller than the map size");
    DM_REQUIRE(ctx, (!ind21 && !ind12) || mass1_in, "mass1 is needed for the indicator maps");
    if (!knn21 && !knn12 && !ind21 && !ind12) return DM_OK;
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));

    const int N1pad = pad_to(N1, GT), N2pad = pad_to(N2, GT);
    const int Kpad = pad_to(k2, GBK);          // contraction of G = Phi2 (k2) . emb1 (k2)
    const int K1pad = pad_to(k1, GBK);
    const bool all = knn12 || ind21 || ind12;      // anything beyond knn21 takes the four-reduction kernel
    const size_t bytes_E2 = 0;                     // emb2 = Phi2 C is only needed for its row norms: never stored (embed_norm_kernel)
    // all four maps: one pass of the four-key fp16 tile kernel + exact re-evaluation (dm_knnsplit.hip).  That
    // path reads Phi2 where it lies (row-major, fp32 or fp64): no K-major copy of it is made.
    const bool split = knn21 && knn12 && ind21 && ind12 && mass1_in && dm_fm_split_ok(ctx, N2, N1, k2);
    const size_t bytes_AT = split ? 0 : (size_t)B * Kpad * N2pad * 8, bytes_BT = (size_t)B * Kpad * N1pad * 8;
    const int nS = dm_cdiv(N1pad, DM_EMB_COLS), nT = dm_cdiv(N2pad, DM_EMB_COLS);
    const size_t bytes_amax = (size_t)B * (nS + nT) * 8;
    const size_t bytes_zero = split ? dm_fm_split_zero_bytes(B) : 0;    // |Phi2| maxima and the per-pair bounds: one memset
    const size_t need = dm_align_up(bytes_AT) + dm_align_up(bytes_BT) + dm_align_up(bytes_E2) +
                        dm_align_up((size_t)B * N1pad * 8) + dm_align_up((size_t)B * N2pad * 8) +
                        dm_align_up((size_t)B * N1 * 8) + dm_align_up(bytes_amax) + dm_align_up(bytes_zero) +
                        (split ? dm_fm_split_ws_bytes(B, N2, N1, k2) : dm_gred_ws_bytes(B, N2, N1));
    int rc = dm_ws_reserve(ctx, need);
    if (rc) return rc;
    double* AT = bytes_AT ? (double*)dm_ws_take(ctx, bytes_AT) : nullptr;
    double* BT = (double*)dm_ws_take(ctx, bytes_BT);
    double* E2 = bytes_E2 ? (double*)dm_ws_take(ctx, bytes_E2) : nullptr;
    double* n1 = (double*)dm_ws_take(ctx, (size_t)B * N1pad * 8);
    double* n2 = (double*)dm_ws_take(ctx, (size_t)B * N2pad * 8);
    double* amaxS = (double*)dm_ws_take(ctx, bytes_amax);
    void* zeroed = bytes_zero ? dm_ws_take(ctx, bytes_zero) : nullptr;
    double* massbuf = (double*)dm_ws_take(ctx, (size_t)B * N1 * 8);      // float64 masses (or zeros when none were given)
    const double* mass1 = nullptr;
    rc = dm_widen_mass(ctx, (long long)B * N1, mass1_in, massbuf, &mass1);
    if (rc) return rc;
    double* amaxT = amaxS + (size_t)B * nS;            // (B, nT) max |Phi2| per 64 vertices, written by the second embedding
    if ((bytes_AT && !AT) || !BT || (bytes_E2 && !E2) || !n1 || !n2 || !amaxS || (bytes_zero && !zeroed))
        return dm_fail(ctx, DM_ENOMEM, "fm_to_p2p: workspace not reserved");
    if (zeroed) DM_CHECK_HIP(ctx, hipMemsetAsync(zeroed, 0, bytes_zero, ctx->stream));

    // AT = Phi2[:, :k2]^T (K-major f64) for the float64 kernel
    if (!split) {
        rc = dm_launch_phiT(ctx, B, N2, k2, Phi2, ld2, AT, Kpad, N2pad, (double*)nullptr);
        if (rc) return rc;
    }
    // BT = emb1^T, emb1 = Phi1[:, :k1] C^T (N1 x k2): emb1T[c][j] = sum_m C[c][m] Phi1[j][m];  n1_j = |emb1_j|^2
    rc = dm_launch_embed(ctx, B, N1, k2, k1, Phi1, ld1, C, k1, (long long)k2 * k1, 0, BT, Kpad, N1pad, n1, 1, amaxS);
    if (rc) return rc;
    // The four-map path's target rows (split fp16 rows of Phi2) need the power-of-two scale of max |Phi2| before the first row is
    // written: a property of the mesh, not of the map.  The second embedding measures those maxima anyway; a call that finds the
    // maxima a previous call on the same basis left (dm_ctx::basis_stat) lets the embedding write the rows as the basis streams by
    // and skips the row builder's pass (134 MB of reads at config 2).  The pass checks the hint against what was measured.
    dm_ctx::basis_stat* st = (split && ctx->opt_basis_stats) ? dm_stat_entry(ctx, Phi2, B, N2, k2, ld2, (int)sizeof(TR), 1) : nullptr;
    dm_fm_split_pre pre{nullptr, false, nullptr, nullptr};
    dm_embed_fx fx{nullptr, 0, 0, nullptr, 0, nullptr};
    if (split) {
        int D_ = 0, R2_ = 0;
        pre.Fx = dm_fm_split_take_fx(ctx, B, N2, k2, &D_, &R2_);
        if (!pre.Fx) return dm_fail(ctx, DM_ENOMEM, "fm_to_p2p: workspace not reserved");
        if (st) {
            // (a buffer = the pairs' maxima of |Phi2|, reduced from the embedding's per-block maxima by the pass's per-row-term workgroups)
            if (st->valid) { fx.F = pre.Fx; fx.D = D_; fx.rows_out = R2_; fx.hint = st->buf[st->cur]; fx.n_hint = 1; pre.built = true; pre.hint = fx.hint; }
            pre.pair_out = st->buf[st->cur ^ 1];
        }
    }
    if (all) {
        // emb2 = Phi2[:, :k2] C (N2 x k1): emb2T[m][i] = sum_c C[c][m] Phi2[i][c];  only n2_i = |emb2_i|^2 is used
        rc = dm_launch_embed(ctx, B, N2, k1, k2, Phi2, ld2, C, k1, (long long)k2 * k1, 1, (double*)nullptr, K1pad, N2pad, n2, 1,
                             (double*)nullptr, split ? amaxT : nullptr, (split && st) ? &fx : nullptr);
        if (rc) return rc;
        if (st) { st->cur ^= 1; st->valid = true; }          // (stream order: the next call reads what this launch wrote)
    }
    dm_gred_args a;
    a.B = B; a.N2 = N2; a.N1 = N1; a.Kloop = Kpad;
    a.AT = AT; a.N2pad = N2pad; a.BT = BT; a.N1pad = N1pad; a.Kpad = Kpad;
    a.n1 = n1; a.n2 = all ? n2 : nullptr; a.mass1 = mass1;
    if (all && !mass1) {                      // nearest-neighbour maps only: the indicator values are never read
        if (!massbuf) return dm_fail(ctx, DM_ENOMEM, "fm_to_p2p: workspace not reserved");
        DM_CHECK_HIP(ctx, hipMemsetAsync(massbuf, 0, (size_t)B * N1 * 8, ctx->stream));
        a.mass1 = massbuf;
    }
    a.knn21 = knn21; a.knn12 = knn12; a.ind21 = ind21; a.ind12 = ind12;
    ctx->last_flag_counts = nullptr;
    if (split) { a.Ktrue = k2; return dm_launch_fm_split<TR>(ctx, a, amaxS, nS, amaxT, nT, zeroed, Phi2, ld2, &pre); }
    return dm_launch_gred(ctx, a);
}
extern "C" int dm_fm_to_p2p(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const float* Phi1, int ld1,
                            const float* Phi2, int ld2, const float* mass1, const double* C, int32_t* knn21,
                            int32_t* knn12, int32_t* ind21, int32_t* ind12) {
    return fm_to_p2p_impl<float>(ctx, B, N1, N2, k1, k2, Phi1, ld1, Phi2, ld2, mass1, C, knn21, knn12, ind21, ind12);
}
extern "C" int dm_fm_to_p2p_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const double* Phi1, int ld1,
                                const double* Phi2, int ld2, const double* mass1, const double* C, int32_t* knn21,
                                int32_t* knn12, int32_t* ind21, int32_t* ind12) {
    return fm_to_p2p_impl<double>(ctx, B, N1, N2, k1, k2, Phi1, ld1, Phi2, ld2, mass1, C, knn21, knn12, ind21, ind12);
}

// which path dm_fm_to_p2p takes for these sizes on this context (bench.py names the dominant kernel accordingly)
extern "C" int dm_fm_to_p2p_uses_split(const dm_ctx* ctx, int N2, int N1, int k) {
    return (ctx && dm_fm_split_ok(ctx, N2, N1, k)) ? ctx->opt_p2p_split : 0;       // 1: two passes, 2: one pass in both directions
}

// ---- generic exact nearest neighbour (pyFM/spectral/nn_utils.py:4-38, k = 1) ---------------------------------
// out[b][i] = argmin_j |X[b][j] - Y[b][i]|^2 = argmin_j |X_j|^2 - 2 <X_j, Y_i>   (lowest j on ties)
extern "C" int dm_knn_query_f64(dm_ctx* ctx, int B, int nx, int ny, int p, const double* X, const double* Y, int32_t* out) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && nx > 0 && ny > 0 && p > 0, "sizes must be positive");
    DM_REQUIRE(ctx, X && Y && out, "null pointer");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const int nxpad = pad_to(nx, GT), nypad = pad_to(ny, GT), Kpad = pad_to(p, GBK);
    const size_t bA = (size_t)B * Kpad * nypad * 8, bB = (size_t)B * Kpad * nxpad * 8;
    int rc = dm_ws_reserve(ctx, dm_align_up(bA) + dm_align_up(bB) + dm_align_up((size_t)B * nxpad * 8) + dm_gred_ws_bytes(B, ny, nx) +
                                dm_knn_split_prep_bytes(B, ny, p) + dm_knn_split_ws_bytes(B, ny, nx, p) +
                                dm_align_up((size_t)B * (nxpad / 256 + 1) * 8));
    if (rc) return rc;
    double* AT = (double*)dm_ws_take(ctx, bA);
    double* BT = (double*)dm_ws_take(ctx, bB);
    double* n1 = (double*)dm_ws_take(ctx, (size_t)B * nxpad * 8);
    double* amaxS = (double*)dm_ws_take(ctx, (size_t)B * (nxpad / 256 + 1) * 8);
    rc = launch_transpose_f64(ctx, B, ny, p, Y, p, AT, Kpad, nypad);
    if (rc) return rc;
    rc = launch_transpose_f64(ctx, B, nx, p, X, p, BT, Kpad, nxpad);
    if (rc) return rc;
    DM_LAUNCH(ctx, "colnorm", colnorm_kernel, dim3(dm_cdiv(nxpad, 256), B), dim3(256), 0, BT, p, Kpad, nxpad, n1, amaxS);
    dm_knn_split_state knn;
    rc = dm_knn_split_prepare(ctx, B, ny, nypad, Kpad, p, AT, &knn);
    if (rc) return rc;
    dm_gred_args a;
    a.B = B; a.N2 = ny; a.N1 = nx; a.Kloop = Kpad; a.Ktrue = p;
    a.AT = AT; a.N2pad = nypad; a.BT = BT; a.N1pad = nxpad; a.Kpad = Kpad;
    a.n1 = n1; a.n2 = nullptr; a.mass1 = nullptr;
    a.knn21 = out; a.knn12 = nullptr; a.ind21 = nullptr; a.ind12 = nullptr;
    return dm_launch_knn21(ctx, a, knn, amaxS, dm_cdiv(nxpad, 256));
}

// ---- dense mapped indicator (pyFM/spectral/convert.py:144) ------------------------------------------------------
// M[b] = ((Phi2[:, :k2] C) Phi1[:, :k1]^T) * mass1[None, :]   (N2 x N1 float64), for callers that want the matrix
// itself; the arg-max maps never need it (dm_fm_to_p2p).
struct OutRowMajorF64 {
    double* p; long long stride_b; int ld;
    __device__ __forceinline__ void store(int b, int i, int j, double v) const { p[b * stride_b + (long long)i * ld + j] = v; }
};
template <typename TR>
struct OutIndicator {
    double* p; long long stride_b; int ld; const TR* mass1; int N1;
    __device__ __forceinline__ void store(int b, int i, int j, double v) const {
        p[b * stride_b + (long long)i * ld + j] = v * (double)mass1[(long long)b * N1 + j];
    }
};
template <typename TR>
static int mapped_indicator_impl(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const TR* Phi1, int ld1,
                                 const TR* Phi2, int ld2, const TR* mass1, const double* C, double* M) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N1 > 0 && N2 > 0 && k1 > 0 && k2 > 0, "sizes must be positive");
    DM_REQUIRE(ctx, Phi1 && Phi2 && mass1 && C && M, "null pointer");
    DM_REQUIRE(ctx, ld1 >= k1 && ld2 >= k2, "eigenvector row stride smaller than the map size");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (k1 <= 32 && k2 <= 32) {
        // small maps: the shared arithmetic of dm_indicator_dev.h (what the linear assignment evaluates from the factors, bit for bit)
        const int KP = k1 <= 16 ? 16 : 32;
        const size_t bE2 = (size_t)B * N2 * KP * 8, bP1 = (size_t)B * N1 * KP * 8, bA = (size_t)B * N1 * 8;
        int rc = dm_ws_reserve(ctx, dm_align_up(bE2) + dm_align_up(bP1) + dm_align_up(bA) + 4096);
        if (rc) return rc;
        double* E2p = (double*)dm_ws_take(ctx, bE2);
        double* P1p = (double*)dm_ws_take(ctx, bP1);
        double* a1p = (double*)dm_ws_take(ctx, bA);
        if (!E2p || !P1p || !a1p) return dm_fail(ctx, DM_ENOMEM, "mapped_indicator: workspace not reserved");
        DM_LAUNCH(ctx, "indicator_e2", ind_e2_kernel<TR>, dim3(dm_cdiv(N2, 256), B), dim3(256), 0, Phi2, ld2, N2, k1, k2, C, KP, E2p, (int32_t*)nullptr);
        DM_LAUNCH(ctx, "indicator_p1", ind_p1_kernel<TR>, dim3(dm_cdiv(N1, 256), B), dim3(256), 0, Phi1, ld1, mass1, N1, k1, KP, P1p, a1p, (int32_t*)nullptr);
        const dim3 grid(dm_cdiv(N1, 256), dm_cdiv(N2, 64), B);
        if (KP == 16) DM_LAUNCH(ctx, "indicator_dense", ind_dense_kernel<16>, grid, dim3(256), 0, (const double*)E2p, (const double*)P1p, (const double*)a1p, N1, N2, M);
        else DM_LAUNCH(ctx, "indicator_dense", ind_dense_kernel<32>, grid, dim3(256), 0, (const double*)E2p, (const double*)P1p, (const double*)a1p, N1, N2, M);
        return DM_OK;
    }
    const size_t bE = (size_t)B * N2 * k1 * 8;
    int rc = dm_ws_reserve(ctx, bE);
    if (rc) return rc;
    double* E2 = (double*)dm_ws_take(ctx, bE);
    // emb2[i][m] = sum_c Phi2[i][c] C[c][m]  ==  NT product of rows Phi2_i and rows (C^T)_m
    {
        KRows<TR> opa{Phi2, (long long)N2 * ld2, ld2, N2, k2};
        KRowsF64 opb{C, (long long)k2 * k1, k1, k1, k2, 1};
        OutRowMajorF64 out{E2, (long long)N2 * k1, k1};
        dim3 grid(dm_cdiv(N2, NT_T) * dm_cdiv(k1, NT_T), 1, B);
        DM_LAUNCH(ctx, "emb2_nt_f64", (gemm_nt_f64<KRows<TR>, KRowsF64, OutRowMajorF64>), grid, dim3(256), 0, opa, opb, out,
                  N2, k1, k2);
    }
    {
        KRowsF64 opa{E2, (long long)N2 * k1, k1, N2, k1, 0};
        KRows<TR> opb{Phi1, (long long)N1 * ld1, ld1, N1, k1};
        OutIndicator<TR> out{M, (long long)N2 * N1, N1, mass1, N1};
        dim3 grid(dm_cdiv(N2, NT_T) * dm_cdiv(N1, NT_T), 1, B);
        DM_LAUNCH(ctx, "indicator_nt_f64", (gemm_nt_f64<KRowsF64, KRows<TR>, OutIndicator<TR>>), grid, dim3(256), 0, opa, opb, out,
                  N2, N1, k1);
    }
    return DM_OK;
}
extern "C" int dm_mapped_indicator(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const float* Phi1, int ld1,
                                   const float* Phi2, int ld2, const float* mass1, const double* C, double* M) {
    return mapped_indicator_impl<float>(ctx, B, N1, N2, k1, k2, Phi1, ld1, Phi2, ld2, mass1, C, M);
}
extern "C" int dm_mapped_indicator_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const double* Phi1, int ld1,
                                       const double* Phi2, int ld2, const double* mass1, const double* C, double* M) {
    return mapped_indicator_impl<double>(ctx, B, N1, N2, k1, k2, Phi1, ld1, Phi2, ld2, mass1, C, M);
}

// The fused ZoomOut iteration (dm_zoomout.hip: zoomout_impl): four launches per iteration
//
//   1. zo_embed_split   emb1 = Phi1[:, :k] C^T on the float64 matrix cores; the rows leave the kernel three ways at once:
//                       as SPLIT fp16 rows of the search ([16 high | 16 low] halves per 16 contraction indices), as float64
//                       rows for the exact re-evaluation, and as |emb1_j|^2 (float64) / the fp32 bias -|emb1_j|^2 sx sy / 2.
//                       (before: embedding kernel -> K-major float64 copy -> row-build kernel re-reading it)
//   2. simnn1_f16_mfma  the search: dm_simnn_core with one biased key on split rows (dm_simnn.hip)
//   3. zo_merge_exact   top-2 of every target row over the tile pass's partials, and the exact float64 re-evaluation of the
//                       rows whose margin is inside the error bound, in the same workgroup (before: merge launch -> queue in
//                       HBM -> exact launch)
//   4. p2pfm_tn_f64     C' = Phi2^T (a2 * Phi1[p21]) (dm_zoomout.hip: p2pfm_direct_kernel, no split-K partials)
//
// Reference arithmetic: pyFM/refine/zoomout.py:7-44 (one iteration), pyFM/spectral/convert.py:96-147 (FM_to_p2p),
// pyFM/spectral/convert.py:14-51 (p2p_to_FM); oracle/dm_oracle.py: zoomout_refine.
//
// Scale of the source rows.  The split needs max |emb1| * sy in a fixed binade range; the maximum of an iteration is only
// known when its last workgroup ends.  The rows are therefore scaled with the power of two derived from the PREVIOUS
// iteration's maximum (a pre-pass computes it for the first one), every iteration records its own maximum, and the merge
// kernel checks the ratio: inside [1/4, 8) the error bound below holds (the fp16 subnormal term is budgeted four times over);
// outside, every row of the pair takes the exact path for that iteration -- slower, never wrong.
#include "dm_device.h"
#include "dm_internal.h"
#include "dm_split.h"
#include "dm_zoomfuse.h"

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// amax[b * nch + chunk] = max |Phi[b][i][c]| over the rows i = chunk (mod nch), c < k   (row-major basis, any TR)
template <typename TR>
__global__ __launch_bounds__(256) void zo_absmax_rows_kernel(const TR* __restrict__ Phi, int N, int k, int ld, int nch,
                                                             double* __restrict__ amax) {
    __shared__ double sh[4];
    const int chunk = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const TR* P = Phi + (long long)b * N * ld;
    double m = 0.0;
    // thread = (row of a group of 256 / kq rows, column): consecutive threads read consecutive columns
    const int kq = k < 256 ? k : 256;
    const int rpg = 256 / kq;                                  // rows per pass
    const int tr = t / kq, tc = t - tr * kq;
    if (tr < rpg) {
        for (int i = chunk + nch * tr; i < N; i += nch * rpg)
            for (int c = tc; c < k; c += kq) m = fmax(m, fabs((double)P[(long long)i * ld + c]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((t & 63) == 0) sh[t >> 6] = m;
    __syncthreads();
    if (t == 0) amax[b * nch + chunk] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

template <typename TR>
int dm_zo_absmax_rows(dm_ctx* ctx, int B, int N, int k, const TR* Phi, int ld, int nch, double* amax) {
    DM_LAUNCH(ctx, "zo_absmax_rows", zo_absmax_rows_kernel<TR>, dim3(nch, B), dim3(256), 0, Phi, N, k, ld, nch, amax);
    return DM_OK;
}
template int dm_zo_absmax_rows<float>(dm_ctx*, int, int, int, const float*, int, int, double*);
template int dm_zo_absmax_rows<double>(dm_ctx*, int, int, int, const double*, int, int, double*);

// ---------------------------------------------------------------------------------------------------------------------
// 1. embedding + split rows
// Workgroup = ZE_NW waves = 16 ZE_NW source vertices; wave w owns vertices 16 w .. 16 w + 15 and ALL of their (<= 16 NRB) embedding
// entries: NRB accumulator tiles of v_mfma_f64_16x16x4_f64 (A = the wave's rows of Phi1, read straight from global memory one
// stage ahead -- a lane needs four entries per stage of 16 contraction indices --, B = rows of C, shared by the waves
// through two LDS stages, one barrier per stage).  A row is complete inside one wave: norm, maximum, split and all three
// stores need no exchange beyond the wave.
constexpr int ZE_LD = 18;                               // LDS row stride of the C stage (f64): conflict-free fragment reads
constexpr int ZE_NW = 8;                                // waves per workgroup (16 vertices each); 4 (two or three workgroups per CU) measured 3 % slower
static inline size_t zo_embed_lds(int NRB) { return (size_t)2 * NRB * 16 * ZE_LD * 8 + ZE_NW * 1024; }

template <typename TR, int NRB>
__global__ __launch_bounds__(64 * ZE_NW, NRB <= 6 ? 4 : 2) void zo_embed_split_kernel(zo_embed_args<TR> a) {
    extern __shared__ __attribute__((aligned(16))) double ze_sm[];
    double* Cs = ze_sm;                                               // [2][16 NRB][ZE_LD]
    unsigned int* scr = reinterpret_cast<unsigned int*>(ze_sm + 2 * NRB * 16 * ZE_LD);   // [ZE_NW waves][256 dwords]
    constexpr int RPP = 8 * ZE_NW;                                    // rows of C staged per pass of the workgroup
    constexpr int NQ = (NRB * 16 + RPP - 1) / RPP;
    const int b = blockIdx.y, v0 = blockIdx.x * (16 * ZE_NW);
    const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int nrb = NRB;                                          // exact: no guard between the matrix instructions
    const int ns = a.nrb, k = a.k;
    typedef __attribute__((address_space(1))) const f64x2 gf64x2;
    typedef __attribute__((address_space(1))) const TR gTR;
    const double* Cb = a.C + (long long)b * a.strideC;
    const int tr = t >> 3, mc = (t & 7) * 2;                          // C staging: rows tr + RPP q, contraction entries mc, mc + 1
    gTR* arow = (gTR*)(a.Phi1 + (long long)b * a.s1 + (long long)min(v0 + wave * 16 + l15, a.N1 - 1) * a.ld1);

    f64x2 rc[NQ];
    TR an[4], ac[4];
#define ZE_FETCH(s_)                                                                           \
    {                                                                                          \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                       \
            const int row_ = tr + RPP * q;                                                     \
            rc[q] = (row_ < 16 * nrb) ? *(gf64x2*)(Cb + (long long)row_ * a.ldc + 16 * (s_) + mc) : f64x2{0.0, 0.0}; \
        }                                                                                      \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                        \
            an[e] = arow[min(16 * (s_) + 4 * e + g, a.ld1 - 1)];                               \
        }                                                                                      \
    }
    f64x4 acc[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) acc[rb] = f64x4{0.0, 0.0, 0.0, 0.0};
    ZE_FETCH(0)
    for (int s = 0; s < ((a.dbg & 4) ? 1 : ns); ++s) {
        double* Cw = Cs + (s & 1) * (NRB * 16 * ZE_LD);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = tr + RPP * q;
            if (row < 16 * NRB) *reinterpret_cast<f64x2*>(Cw + row * ZE_LD + mc) = rc[q];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ac[e] = an[e];
        __syncthreads();
        // (the buffer written above was last read two stages ago, and every wave has passed a barrier since)
        if (s + 1 < ns) ZE_FETCH(s + 1)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            // (entries beyond the map are zeroed HERE, a stage after their fetch: a select right behind the load made every
            //  fetch wait for its own data)
            const double av = (16 * s + 4 * ks + g < k) ? (double)ac[ks] : 0.0;
            double bv[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) bv[rb] = Cw[(rb * 16 + l15) * ZE_LD + 4 * ks + g];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[rb] = mfma_f64_16x16x4(av, bv[rb], acc[rb]);
        }
    }
#undef ZE_FETCH
    // acc[rb][q] = emb1[vertex v0 + 16 wave + g + 4 q][16 rb + l15]
    double amax = 0.0, ss[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
        if (rb < nrb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double v = acc[rb][q];
                amax = fmax(amax, fabs(v));
                ss[q] = fma(v, v, ss[q]);
            }
        }
    // (vertices beyond N1 repeat the last one: they change no maximum)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off));
    // one atomic per workgroup (the waves meet in the scratch area, which nothing has used yet)
    double* wmax = reinterpret_cast<double*>(scr);
    if (lane == 0) wmax[wave] = amax;
    __syncthreads();
    if (t == 0) {
        double m = wmax[0];
#pragma unroll
        for (int w = 1; w < ZE_NW; ++w) m = fmax(m, wmax[w]);
        atomicMax(a.amax_cur + b, (unsigned long long)__double_as_longlong(m));
    }
    if (a.only_max || (a.dbg & 8)) return;
    __syncthreads();                                  // (the scratch area is reused below)

    double mp = __longlong_as_double((long long)a.amax_prev[b]);
    const double sy = ks_scale(&mp, 1);
    const double sxy = ks_scale(a.amaxT + b * a.nT, a.nT) * sy;
    float bm = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // sum over the 16 lanes of a row group on the vector ALU (DPP): the lane that stores (l15 == 0) adds in one fixed order
        ss[q] += dpp_f64<0xB1>(ss[q]);
        ss[q] += dpp_f64<0x4E>(ss[q]);
        ss[q] += dpp_f64<0x141>(ss[q]);
        ss[q] += dpp_f64<0x140>(ss[q]);
        const int vv = v0 + wave * 16 + g + 4 * q;
        if (l15 == 0 && vv < a.N1) {
            const float bi = (float)(-0.5 * ss[q] * sxy);
            a.n1[(long long)b * a.N1pad + vv] = ss[q];
            a.bias[(long long)b * a.R1 + vv] = bi;
            bm = fmaxf(bm, fabsf(bi));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bm = fmaxf(bm, __shfl_xor(bm, off));
    if (lane == 0) reinterpret_cast<float*>(scr)[wave] = bm;           // one atomic per workgroup
    __syncthreads();
    if (t == 0) {
        float m = 0.f;
#pragma unroll
        for (int w = 0; w < ZE_NW; ++w) m = fmaxf(m, reinterpret_cast<float*>(scr)[w]);
        atomicMax(a.bmax + b, __float_as_uint(m));
    }
    __syncthreads();

    unsigned int* sw = scr + wave * 256;
    const int vloc_r = lane >> 2, chunk_r = lane & 3;                  // the 16-byte piece this lane copies out of the image
    const int vrow = v0 + wave * 16 + vloc_r;
    const bool row_ok = vrow < a.N1 && !(a.dbg & 1);
    _Float16* frow = a.Fy + ((long long)b * a.R1 + min(vrow, a.N1 - 1)) * a.ldS + 8 * chunk_r;
    double* erow[4];
    bool vok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int vv = v0 + wave * 16 + g + 4 * q;
        vok[q] = vv < a.N1 && !(a.dbg & 2);
        erow[q] = a.embr + ((long long)b * a.N1 + min(vv, a.N1 - 1)) * a.Kpad + l15;
    }
    // even lanes store the pair of high halves (r, r + 1), odd lanes the pair of low halves (r - 1, r): one dword per lane
    const bool odd = (l15 & 1) != 0;
    const int swo = (odd ? 8 : 0) + (l15 >> 1);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (vok[q]) erow[q][rb * 16] = acc[rb][q];
            _Float16 h, l;
            split2_hw(acc[rb][q] * sy, h, l);
            const unsigned int hu = (unsigned int)__builtin_bit_cast(unsigned short, h);
            const unsigned int lu = (unsigned int)__builtin_bit_cast(unsigned short, l);
            const unsigned int oh = (unsigned int)dpp_i32<0xB1>((int)hu), ol = (unsigned int)dpp_i32<0xB1>((int)lu);   // lane ^ 1
            sw[(g + 4 * q) * 16 + swo] = odd ? (ol | (lu << 16)) : (hu | (oh << 16));
        }
        // (LDS operations of one wave execute in order: the read below sees the writes above, the next block's writes come
        //  after it.  No fence: a fence would also wait for the global stores in flight, once per block.)
        __builtin_amdgcn_wave_barrier();
        const u32x4_t piece = *reinterpret_cast<const u32x4_t*>(sw + lane * 4);
        if (row_ok) *reinterpret_cast<u32x4_t*>(frow + 32 * rb) = piece;
        __builtin_amdgcn_wave_barrier();
    }
    // halves [32 nrb, D) of every row are zero (contraction depths below the tile kernel's minimum are padded)
    const int pc = (a.D - 32 * nrb) >> 3;
    if (pc > 0) {
        const u32x4_t z = {0u, 0u, 0u, 0u};
        for (int c = lane; c < 16 * pc; c += 64) {
            const int vl = c / pc, qq = c - vl * pc;
            const int vv = v0 + wave * 16 + vl;
            if (vv < a.N1) *reinterpret_cast<u32x4_t*>(a.Fy + ((long long)b * a.R1 + vv) * a.ldS + 32 * nrb + 8 * qq) = z;
        }
    }
}

template <typename TR, int NRB>
static int zo_embed_launch(dm_ctx* ctx, int B, const zo_embed_args<TR>& a) {
    const size_t lds = zo_embed_lds(NRB);
    int rc = dm_grant_lds(ctx, (const void*)zo_embed_split_kernel<TR, NRB>, lds);
    if (rc) return rc;
    DM_LAUNCH(ctx, "zo_embed_split", (zo_embed_split_kernel<TR, NRB>), dim3(dm_cdiv(a.N1, 16 * ZE_NW), B), dim3(64 * ZE_NW), lds, a);
    return DM_OK;
}
template <typename TR>
int dm_zo_embed_split(dm_ctx* ctx, int B, const zo_embed_args<TR>& a) {
    switch (a.nrb) {            // one instantiation per number of 16-column blocks: straight-line matrix code, registers to fit
        case 1: return zo_embed_launch<TR, 1>(ctx, B, a);
        case 2: return zo_embed_launch<TR, 2>(ctx, B, a);
        case 3: return zo_embed_launch<TR, 3>(ctx, B, a);
        case 4: return zo_embed_launch<TR, 4>(ctx, B, a);
        case 5: return zo_embed_launch<TR, 5>(ctx, B, a);
        case 6: return zo_embed_launch<TR, 6>(ctx, B, a);
        case 7: return zo_embed_launch<TR, 7>(ctx, B, a);
        case 8: return zo_embed_launch<TR, 8>(ctx, B, a);
        case 9: return zo_embed_launch<TR, 9>(ctx, B, a);
        case 10: return zo_embed_launch<TR, 10>(ctx, B, a);
        case 11: return zo_embed_launch<TR, 11>(ctx, B, a);
        case 12: return zo_embed_launch<TR, 12>(ctx, B, a);
        case 13: return zo_embed_launch<TR, 13>(ctx, B, a);
        default: return dm_fail(ctx, DM_EINVAL, "zo_embed_split: map size %d beyond 208", a.k);
    }
}
template int dm_zo_embed_split<float>(dm_ctx*, int, const zo_embed_args<float>&);
template int dm_zo_embed_split<double>(dm_ctx*, int, const zo_embed_args<double>&);

// ---------------------------------------------------------------------------------------------------------------------
// 3. merge + exact.  One workgroup = 16 target rows of a pair.  Phase 1: sixteen lanes per row merge the row's partials of the tile
// pass (top-2 with the lowest-index rule, like simnn_merge_kernel), the arg-max is written, and the rows whose margin is
// inside the bound go on a list in LDS.  Phase 2: the whole workgroup re-evaluates the listed rows one by one in float64
// (ks_exact_row: only the 32-candidate blocks the partials cannot rule out).
constexpr int ZM_ROWS = 16;                        // target rows per workgroup: 16 lanes per row in the merge phase
constexpr int ZM_KMAX = 208;                       // largest map of the fused path

// Exact float64 re-evaluation of ONE queued row by ONE wave (no workgroup barrier: the four waves of a workgroup work on four
// rows at once).  Same arithmetic and summation schedule as ks_exact_row<0> (dm_exact.h): per candidate eight partial sums over
// the contraction indices r = p (mod 8), ascending fma chains, added in the order p = 0 .. 7; value |y_j|^2 - 2 g; blocks and
// candidates ascend and comparisons are strict, so the lowest index wins ties.  Two lanes per candidate: lane (c, hh) carries
// the parts 4 hh .. 4 hh + 3 (two 16-byte loads per eight indices), the odd lane hands its four sums to the even one.
//   xw: this wave's K8 = 8 ceil(K / 8) doubles of LDS
template <typename TR>
__device__ __forceinline__ void zo_exact_row_wave(const zo_mx_args<TR>& a, int o, float thr, double* xw) {
    const int lane = threadIdx.x & 63, c = lane >> 1, hh = lane & 1;
    const int K = a.K, K8 = (K + 7) & ~7;
    const int b = o / a.N2, i = o - b * a.N2;
    const TR* __restrict__ trow = a.Phi2 + ((long long)b * a.N2 + i) * a.ld2;
    for (int r = lane; r < K8; r += 64) xw[r] = (r < K) ? (double)trow[r] : 0.0;       // (zero beyond K: the padded products are exact zeros)
    __builtin_amdgcn_wave_barrier();
    const int nparts = a.q.nparts, pw = a.q.pw, nsub = nparts * (pw / 32);
    const double* __restrict__ E = a.embr + (long long)b * a.N1 * a.Kpad;
    const double* __restrict__ n1 = a.n1 + (long long)b * a.N1pad;
    double bv = DM_INF_F64;
    int bj = DM_IDX_NONE;
    for (int sb0 = 0; sb0 < nsub; sb0 += 64) {
        const int sbt = sb0 + lane;
        const bool keep = sbt < nsub && dm_simnn_keep(a.q.pb, a.q.pj, a.q.ps, nparts, pw, a.q.Npad, b, i, sbt, thr);
        unsigned long long mm = __ballot(keep);                   // uniform
        while (mm) {
            const int sb = sb0 + __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const int j = sb * 32 + c;
            const int jc = min(j, a.N1 - 1);
            const double* Cr = E + (long long)jc * a.Kpad + 4 * hh;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int r = 0;
            // (loads of four steps ahead of their ordered fma chains: one L2 round trip per 32 indices instead of per 8)
            for (; r + 32 <= K8; r += 32) {
                f64x2 y[8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    y[2 * u] = *reinterpret_cast<const f64x2*>(Cr + r + 8 * u);
                    y[2 * u + 1] = *reinterpret_cast<const f64x2*>(Cr + r + 8 * u + 2);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f64x2 x0 = *reinterpret_cast<const f64x2*>(xw + r + 8 * u + 4 * hh), x1 = *reinterpret_cast<const f64x2*>(xw + r + 8 * u + 4 * hh + 2);
                    s0 = fma(x0[0], y[2 * u][0], s0); s1 = fma(x0[1], y[2 * u][1], s1);
                    s2 = fma(x1[0], y[2 * u + 1][0], s2); s3 = fma(x1[1], y[2 * u + 1][1], s3);
                }
            }
            for (; r < K8; r += 8) {
                const f64x2 y0 = *reinterpret_cast<const f64x2*>(Cr + r), y1 = *reinterpret_cast<const f64x2*>(Cr + r + 2);
                const f64x2 x0 = *reinterpret_cast<const f64x2*>(xw + r + 4 * hh), x1 = *reinterpret_cast<const f64x2*>(xw + r + 4 * hh + 2);
                s0 = fma(x0[0], y0[0], s0); s1 = fma(x0[1], y0[1], s1); s2 = fma(x1[0], y1[0], s2); s3 = fma(x1[1], y1[1], s3);
            }
            // parts 4 .. 7 from the odd lane (lane ^ 1), then the fixed order 0 .. 7
            const double t0 = dpp_f64<0xB1>(s0), t1 = dpp_f64<0xB1>(s1), t2 = dpp_f64<0xB1>(s2), t3 = dpp_f64<0xB1>(s3);
            const double g = ((((((s0 + s1) + s2) + s3) + t0) + t1) + t2) + t3;
            if (hh == 0 && j < a.N1) {
                const double v = n1[j] - 2.0 * g;                 // |y|^2 - 2 <x, y>
                if (v < bv) { bv = v; bj = j; }
            }
        }
    }
#pragma unroll
    for (int off = 2; off < 64; off <<= 1) {
        const double ov = __shfl_xor(bv, off);
        const int oj = __shfl_xor(bj, off);
        argmin_merge(bv, bj, ov, oj);
    }
    if (lane == 0 && bj != DM_IDX_NONE) a.nn[o] = bj;
    __builtin_amdgcn_wave_barrier();                              // (xw is rewritten by this wave's next row)
}

template <typename TR>
__global__ __launch_bounds__(256) void zo_merge_exact_kernel(zo_mx_args<TR> a) {
    __shared__ __attribute__((aligned(16))) double xrow[4][ZM_KMAX];
    __shared__ int flist[ZM_ROWS];
    __shared__ float fthr[ZM_ROWS];
    __shared__ int fcount;
    const int b = blockIdx.y, t = threadIdx.x;
    const int i = blockIdx.x * ZM_ROWS + (t >> 4), sub = t & 15;
    if (t == 0) fcount = 0;
    __syncthreads();
    float bv = DM_NEG_INF_F32, sv = DM_NEG_INF_F32;
    int bj = DM_IDX_NONE;
    if (i < a.N2) {
        for (int q = sub; q < a.q.nparts; q += 16) {
            const long long o = ((long long)b * a.q.nparts + q) * a.q.Npad + i;
            top2_merge(bv, bj, sv, a.q.pb[o], a.q.pj[o], a.q.ps[o]);
        }
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const float ob = __shfl_xor(bv, off), os = __shfl_xor(sv, off);
        const int oj = __shfl_xor(bj, off);
        top2_merge(bv, bj, sv, ob, oj, os);
    }
    // the scale of the source rows was derived from the previous iteration's maximum: the bound holds while the true one is near it
    const double mp = __longlong_as_double((long long)a.amax_prev[b]), mcur = __longlong_as_double((long long)a.amax_cur[b]);
    const double ratio = mcur * ks_scale(&mp, 1);
    const bool forced = !(ratio >= 0.25 && ratio < 8.0);
    if (i < a.N2 && sub == 0) {
        const long long o = (long long)b * a.N2 + i;
        a.nn[o] = (bj == DM_IDX_NONE) ? 0 : bj;
        const float tau = a.tau_scale * (sqrtf(a.tnorm2[o] * __uint_as_float(a.smax2[b])) + __uint_as_float(a.bmax[b]));
        if (forced || !(bv - sv > tau)) {
            const int pos = atomicAdd(&fcount, 1);
            flist[pos] = (int)o;
            fthr[pos] = forced ? DM_NEG_INF_F32 : bv - tau;
        }
    }
    __syncthreads();
    const int cnt = fcount;
    if (cnt == 0 || a.dbg) return;
    const int wave = t >> 6;
    for (int e = wave; e < cnt; e += 4) zo_exact_row_wave<TR>(a, flist[e], fthr[e], xrow[wave]);
}

template <typename TR>
int dm_zo_merge_exact(dm_ctx* ctx, int B, const zo_mx_args<TR>& a) {
    if (a.K > ZM_KMAX) return dm_fail(ctx, DM_EINVAL, "zo_merge_exact: map size %d beyond %d", a.K, ZM_KMAX);
    DM_LAUNCH(ctx, "zo_merge_exact", zo_merge_exact_kernel<TR>, dim3(dm_cdiv(a.N2, ZM_ROWS), B), dim3(256), 0, a);
    return DM_OK;
}
template int dm_zo_merge_exact<float>(dm_ctx*, int, const zo_mx_args<float>&);
template int dm_zo_merge_exact<double>(dm_ctx*, int, const zo_mx_args<double>&);

__global__ __launch_bounds__(256) void zo_copy_mat_kernel(int rows, int cols, const double* __restrict__ src, int lds, long long ss,
                                                          double* __restrict__ dst, int ldd, long long sd) {
    const int b = blockIdx.y;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)rows * cols) return;
    const int r = (int)(e / cols), c = (int)(e - (long long)r * cols);
    dst[b * sd + (long long)r * ldd + c] = src[b * ss + (long long)r * lds + c];
}
int dm_zo_copy_mat(dm_ctx* ctx, int B, int rows, int cols, const double* src, int lds, long long ss, double* dst, int ldd, long long sd) {
    const long long n = (long long)rows * cols;
    DM_LAUNCH(ctx, "zo_copy_mat", zo_copy_mat_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, rows, cols, src, lds, ss, dst, ldd, sd);
    return DM_OK;
}

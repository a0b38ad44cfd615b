// The fused ZoomOut iteration (dm_zoomout.hip: zoomout_impl): five launches per iteration
//
//   1. zo_embed_split   emb1 = Phi1[:, :k] C^T on the float64 matrix cores; the rows leave the kernel three ways at once:
//                       as SPLIT fp16 rows of the search ([16 high | 16 low] halves per 16 contraction indices), as float64
//                       rows for the exact re-evaluation, and as |emb1_j|^2 (float64) / the fp32 bias -|emb1_j|^2 sx sy / 2.
//                       (before: embedding kernel -> K-major float64 copy -> row-build kernel re-reading it)
//   2. simnn1_f16_mfma  the search: dm_simnn_core with one biased key on split rows (dm_simnn.hip)
//   3. zo_merge         top-2 of every target row over the tile pass's partials (a thread per row), queue of the rows whose
//      zo_exact         margin is inside the error bound; their exact float64 re-evaluation, a workgroup per row with all of a
//                       block's loads in flight (two lean launches: as one kernel of 4096 small workgroups it was bound by
//                       their dispatch, 36 us; as these two 8 + 12)
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
#ifndef ZE_NW_DEF
#define ZE_NW_DEF 8
#endif
constexpr int ZE_NW = ZE_NW_DEF;                        // waves per workgroup (16 vertices each); 4 (two or three workgroups per CU) measured 3 % slower
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
    // the scales of the rows (known before the call: the previous iteration's maximum, the target operand's maxima) are fetched
    // here, under the main loop, not in front of the epilogue that needs them
    double sy = 1.0, sxy = 1.0;
    if (!a.only_max) {
        double mp = __longlong_as_double((long long)a.amax_prev[b]);
        sy = ks_scale(&mp, 1);
        sxy = ks_scale(a.amaxT + b * a.nT, a.nT) * sy;
    }
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
// 3. merge, exact.  zo_merge: top-2 of every target row over the tile pass's partials (lowest-index rule, like simnn_merge_kernel),
// the arg-max is written, the rows whose margin is inside the bound are queued.  zo_exact: a workgroup per queued row
// re-evaluates it in float64 (only the 32-candidate blocks the partials cannot rule out, like ks_exact_row).
constexpr int ZM_KMAX = 208;                       // largest map of the fused path (<= 256: four chunks of 64 indices)

// Exact float64 re-evaluation of ONE queued row by the WHOLE workgroup (r04 second half; before: one wave per row, two lanes per
// candidate, the row's loads in seven dependent rounds -- a flagged row took 10 - 20 us, and the kernel ended with the last one).
// Same arithmetic and summation schedule as ks_exact_row<0> (dm_exact.h): per candidate eight partial sums over the contraction
// indices r = p (mod 8), ascending fma chains, added in the order p = 0 .. 7; value |y_j|^2 - 2 g; blocks and candidates ascend
// and comparisons are strict, so the lowest index wins ties.  Eight lanes per candidate (lane p carries part p: the eight lanes
// read one 64-byte run per eight indices), a wave = 8 candidates, the workgroup = one block of 32; ALL loads of a block are in
// flight at once up to k = 128 (sixteen per lane and round: one L2 / fabric round trip per 128 indices instead of one per 32).
//   xw: 256 doubles of LDS (zero beyond K: the padded products are exact zeros), redv / redj: one slot per wave
template <typename TR>
__device__ __forceinline__ void zo_exact_row_wg(const zo_mx_args<TR>& a, int o, float thr, double* xw, double* redv, int* redj) {
    const int t = threadIdx.x, lane = t & 63, p = lane & 7, c8 = lane >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int K = a.K, K8 = (K + 7) & ~7;
    const int b = o / a.N2, i = o - b * a.N2;
    const TR* __restrict__ trow = a.Phi2 + ((long long)b * a.N2 + i) * a.ld2;
    // Dependent round trips of a flagged row: the target row x, the row's partials (just read by the merge phase: cache hits),
    // the candidates and their norms.  x is requested first and only written to LDS when the first block's candidates are in
    // flight, the norm before the products it is combined with: two round trips instead of four.
    const int nh = (K8 + 127) >> 7;                                // rounds of 128 contraction indices
    const double xr = (t < K) ? (double)trow[t] : 0.0;             // (K <= 256)
    bool x_in_lds = false;                                        // uniform
    const int nparts = a.q.nparts, pw = a.q.pw, nsub = nparts * (pw / 32);
    const double* __restrict__ E = a.embr + (long long)b * a.N1 * a.Kpad;
    const double* __restrict__ n1 = a.n1 + (long long)b * a.N1pad;
    double bv = DM_INF_F64;
    int bj = DM_IDX_NONE;
    for (int sb0 = 0; sb0 < nsub; sb0 += 64) {
        const int sbt = sb0 + lane;                               // (every wave takes the same ballot: no exchange)
        bool keep = false;
        {   // dm_simnn_keep with its three loads side by side (short-circuit evaluation made them three dependent round trips)
            const int q = min((sbt * 32) / pw, nparts - 1);
            const long long oq = ((long long)b * nparts + q) * a.q.Npad + i;
            const float qs = a.q.ps[oq], qb = a.q.pb[oq];
            const int qj = a.q.pj[oq];
            keep = sbt < nsub && (qs >= thr || (qb >= thr && (qj >> 5) == sbt));
        }
        unsigned long long mm = __ballot(keep);                   // uniform
        // four kept blocks per pass, their loads in flight together (a flagged row typically keeps one whole partial = four
        // blocks: one round trip instead of four; the launch has at most two workgroups per CU, registers are free)
        while (mm) {
            constexpr int NB = 4, NU = 16;
            int sb[NB];
            bool on[NB];                                          // uniform
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                on[c] = mm != 0;
                sb[c] = on[c] ? sb0 + __ffsll((long long)mm) - 1 : sb[0];
                mm &= mm - 1;
            }
#ifdef DM_EXPERIMENTS
            if (a.dbg & 2) mm = 0;                                // (the first pass only)
#endif
            int j[NB], jc[NB];
            const double* Cr[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                j[c] = sb[c] * 32 + wave * 8 + c8;
                jc[c] = (a.dbg & 4) ? wave * 8 + c8 : min(j[c], a.N1 - 1);   // (experiments: every block reads the pair's first rows)
                Cr[c] = E + (long long)jc[c] * a.Kpad;
            }
            // sixteen loads per block and round (one round up to k = 128, two beyond)
            double sacc[NB], nj[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) { sacc[c] = 0.0; nj[c] = 0.0; }
            for (int h = 0; h < nh; ++h) {
                double y[NB][NU];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int r = 8 * (NU * h + u) + p;
                    const int rr = r < K8 ? r : p;                // (beyond the row: any finite entry, its x is zero)
#pragma unroll
                    for (int c = 0; c < NB; ++c) y[c][u] = Cr[c][rr];
                }
                if (h == 0) {
#pragma unroll
                    for (int c = 0; c < NB; ++c) nj[c] = n1[jc[c]];
                    if (!x_in_lds) {
                        __syncthreads();                          // (the previous row's readers of xw / red are done)
                        xw[t] = xr;
                        __syncthreads();
                        x_in_lds = true;
                    }
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const double xv = xw[8 * (NU * h + u) + p];
#pragma unroll
                    for (int c = 0; c < NB; ++c) sacc[c] = fma(xv, y[c][u], sacc[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                // parts 1 .. 7 from the lanes p + 1 .. p + 7 of the row (row_shl), added in the fixed order: valid in the lanes p == 0
                double g = sacc[c];
                g += dpp_f64<0x101>(sacc[c]); g += dpp_f64<0x102>(sacc[c]); g += dpp_f64<0x103>(sacc[c]); g += dpp_f64<0x104>(sacc[c]);
                g += dpp_f64<0x105>(sacc[c]); g += dpp_f64<0x106>(sacc[c]); g += dpp_f64<0x107>(sacc[c]);
                if (p == 0 && on[c] && j[c] < a.N1) {             // (blocks ascend: strict comparisons keep the lowest index)
                    const double v = nj[c] - 2.0 * g;             // |y|^2 - 2 <x, y>
                    if (v < bv) { bv = v; bj = j[c]; }
                }
            }
        }
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const double ov = __shfl_xor(bv, off);
        const int oj = __shfl_xor(bj, off);
        argmin_merge(bv, bj, ov, oj);
    }
    if (!x_in_lds) __syncthreads();                               // (no block kept: still behind the previous row's readers of red)
    if (lane == 0) { redv[wave] = bv; redj[wave] = bj; }
    __syncthreads();
    if (t == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) argmin_merge(bv, bj, redv[w], redj[w]);
        if (bj != DM_IDX_NONE) a.nn[o] = bj;
    }
}

// Merge: ONE THREAD per target row reads the row's partials of the tile pass (consecutive rows = consecutive lanes: every load
// instruction reads 256 contiguous bytes), keeps the top-2 with the lowest-index rule, writes the arg-max and appends the rows
// whose margin is inside the error bound to the iteration's queue (one atomic per wave that holds any).  256 workgroups
// instead of 4096: the r04 kernel (16 rows per workgroup, its flagged rows re-evaluated in the same workgroup) spent 20 - 39 us
// on getting its 4096 workgroups dispatched while the flagged ones held their slots (per-workgroup stamps, tools/zo_experiment.py).
template <typename TR>
__global__ __launch_bounds__(256) void zo_merge_kernel(zo_mx_args<TR> a) {
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63;
    const int i = blockIdx.x * 256 + t, ic = min(i, a.N2 - 1);
    const long long orow = (long long)b * a.N2 + ic;
    const float tn2 = a.tnorm2[orow];
    const float sm2 = __uint_as_float(a.smax2[b]), bmx = __uint_as_float(a.bmax[b]);
    const double mp = __longlong_as_double((long long)a.amax_prev[b]), mcur = __longlong_as_double((long long)a.amax_cur[b]);
    float bv = DM_NEG_INF_F32, sv = DM_NEG_INF_F32;
    int bj = DM_IDX_NONE;
    const long long o0 = (long long)b * a.q.nparts * a.q.Npad + ic;
    // eight partials (24 loads) in flight per round; a round past the end re-reads the last partial and merges nothing
    const float* __restrict__ pb = a.q.pb; const int32_t* __restrict__ pj = a.q.pj; const float* __restrict__ ps = a.q.ps;
    for (int q0 = 0; q0 < a.q.nparts; q0 += 8) {
        float vb[8], vs[8];
        int vj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long o = o0 + (long long)min(q0 + u, a.q.nparts - 1) * a.q.Npad;
            vb[u] = pb[o]; vj[u] = pj[o]; vs[u] = ps[o];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (q0 + u < a.q.nparts) top2_merge(bv, bj, sv, vb[u], vj[u], vs[u]);
    }
    // the scale of the source rows was derived from the previous iteration's maximum: the bound holds while the true one is near it
    const double ratio = mcur * ks_scale(&mp, 1);
    const bool forced = !(ratio >= 0.25 && ratio < 8.0);
    const float tau = a.tau_scale * (sqrtf(tn2 * sm2) + bmx);
    const bool flag = i < a.N2 && (forced || !(bv - sv > tau));
    if (i < a.N2) a.nn[orow] = (bj == DM_IDX_NONE) ? 0 : bj;
    const unsigned long long fm = __ballot(flag);
    if (fm) {                                                     // uniform per wave
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(a.qcount, (unsigned int)__popcll(fm));
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        if (flag) {
            const unsigned int pos = base + (unsigned int)__popcll(fm & ((1ull << lane) - 1ull));
            a.qrow[pos] = (int)orow;
            a.qthr[pos] = forced ? DM_NEG_INF_F32 : bv - tau;
        }
    }
}

// Exact: the queue's rows, one per workgroup at a time (a fixed grid strides over the queue: its length is only known on the
// device).
template <typename TR>
__global__ __launch_bounds__(256, 2) void zo_exact_kernel(zo_mx_args<TR> a) {
    __shared__ __attribute__((aligned(16))) double xrow[256];
    __shared__ double redv[4];
    __shared__ int redj[4];
    // (the first entry is requested together with the queue's length: beyond the length it is stale, never out of bounds)
    unsigned int e = blockIdx.x;
    const unsigned int ec = min(e, (unsigned int)a.qcap - 1u);
    int o = a.qrow[ec];
    float thr = a.qthr[ec];
    const unsigned int cnt = *a.qcount;
    if (a.dbg & 1) return;
    if (e >= cnt) return;
    while (true) {
        const unsigned int en = e + gridDim.x;                    // (the next entry is requested before this row's work)
        const bool more = en < cnt;
        const int on = more ? a.qrow[en] : 0;
        const float thrn = more ? a.qthr[en] : 0.f;
        zo_exact_row_wg<TR>(a, o, thr, xrow, redv, redj);
        if (!more) break;
        e = en; o = on; thr = thrn;
    }
}

constexpr int ZX_GRID = 512;                                      // workgroups of the exact launch
template <typename TR>
int dm_zo_merge_exact(dm_ctx* ctx, int B, const zo_mx_args<TR>& a) {
    if (a.K > ZM_KMAX) return dm_fail(ctx, DM_EINVAL, "zo_merge_exact: map size %d beyond %d", a.K, ZM_KMAX);
    DM_LAUNCH(ctx, "zo_merge", zo_merge_kernel<TR>, dim3(dm_cdiv(a.N2, 256), B), dim3(256), 0, a);
    DM_LAUNCH(ctx, "zo_exact", zo_exact_kernel<TR>, dim3(ZX_GRID), dim3(256), 0, a);
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

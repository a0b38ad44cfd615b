// Feature-similarity nearest neighbour (dm_simnn_f16), BASELINE.json config 3.
//
//   nn21[b,i] = argmax_j <Ftgt[b,i,:], Fsrc[b,j,:]>          oracle/dm_oracle.py: simnn
//
// S^T = Fsrc Ftgt^T is produced 128x128 tile by tile on the fp16 matrix cores
// (v_mfma_f32_32x32x16_f16: exact fp16 products, fp32 accumulation) and consumed in registers by
// a top-2 row reduction; S never reaches memory.  The operands are swapped (src is the MFMA "A"
// side) so that each lane owns ONE target row and 16 source candidates per MFMA tile: the
// reduction is in-lane except for one cross-half step.
//
// Exactness: fp32 accumulation can reorder near-ties.  Every row whose (best - second best) is
// within twice the accumulation error bound  D (1 + 1/16) 2^-23 |t_i| max_j |s_j|  is re-evaluated
// in float64 (products of fp16 are exact in f64, the f64 sum is exact to 1e-16 relative), so the
// returned index equals the float64 argmax with the lowest-index tie rule.
#include "dm_device.h"
#include "dm_internal.h"

constexpr int ST = 128;    // tile: 128 target rows x 128 source rows
constexpr int SBK = 64;    // contraction (halves) per LDS stage: one 128-byte line per row
#define DM_NEG_INF_F32 (-__builtin_huge_valf())

__device__ __forceinline__ void top2_merge(float& b, int& i, float& s, float ob, int oi, float os) {
    if (ob > b || (ob == b && oi < i)) { s = fmaxf(b, os); b = ob; i = oi; }
    else { s = fmaxf(s, ob); }
}

// LDS image of a 128 x 64 fp16 tile: row r is one 128-byte line of eight 16-byte chunks; chunk c is
// stored at slot c ^ ((r >> 1) & 7).  Two consecutive rows fill one 256-byte bank row, so the 16
// rows (distinct mod 16) that one ds_read_b128 lane group touches land on 16 different slots.
__device__ __forceinline__ int lds_off_halves(int row, int chunk) {
    return row * SBK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

struct simnn_params {
    const _Float16* Ftgt; const _Float16* Fsrc;
    float* pb; int32_t* pj; float* ps;       // partials (B, tilesS, N2pad)
    float* tnorm2;                           // (B, N2)  |t_i|^2, written by the workgroups of source tile 0
    unsigned int* smax2;                     // (B)      max_j |s_j|^2 as float bits (atomicMax), by target tile 0
    int N2, N1, D, N2pad, tilesT, tilesS, total;
};

__global__ __launch_bounds__(256, 2) void simnn_kernel(simnn_params p) {
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * 2 * ST * SBK];   // T[2] | S[2], 64 KiB
    _Float16* Ts = smem;
    _Float16* Ss = smem + 2 * ST * SBK;

    const int id = xcd_remap(blockIdx.x, p.total);
    const int tiles = p.tilesT * p.tilesS;
    const int b = id / tiles;
    const int tts = id - b * tiles;
    const int tt_ = tts / p.tilesS, ts_ = tts - tt_ * p.tilesS;
    const int i0 = tt_ * ST, j0 = ts_ * ST;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wt = wave >> 1, ws = wave & 1;

    const _Float16* T = p.Ftgt + (long long)b * p.N2 * p.D;
    const _Float16* S = p.Fsrc + (long long)b * p.N1 * p.D;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    // squared row norms for the exactness bound, accumulated from the MFMA fragments by the workgroups that
    // own the first tile of the other operand (every row of T / S is seen exactly once that way)
    const bool do_tn = (ts_ == 0) && (ws == 0);
    const bool do_sn = (tt_ == 0) && (wt == 0);
    float nrm_t[2] = {0.f, 0.f}, nrm_s[2] = {0.f, 0.f};

    const int lrow = t >> 3, lchunk = t & 7;
    uint4 rt[4], rs[4];
    // (macros, not lambdas: by-reference lambda captures of the staging arrays end up in scratch)
#define SIMNN_FETCH(s_)                                                                                          \
    {                                                                                                            \
        const int k_ = (s_) * SBK + lchunk * 8;                                                                  \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                          \
            const int row = q * 32 + lrow;                                                                       \
            const int gi = i0 + row, gj = j0 + row;                                                              \
            rt[q] = (gi < p.N2 && k_ < p.D) ? *reinterpret_cast<const uint4*>(T + (long long)gi * p.D + k_)      \
                                            : uint4{0, 0, 0, 0};                                                 \
            rs[q] = (gj < p.N1 && k_ < p.D) ? *reinterpret_cast<const uint4*>(S + (long long)gj * p.D + k_)      \
                                            : uint4{0, 0, 0, 0};                                                 \
        }                                                                                                        \
    }
#define SIMNN_STASH(buf_)                                                                                        \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
        const int row = q * 32 + lrow;                                                                           \
        const int off = (buf_) * ST * SBK + lds_off_halves(row, lchunk);                                         \
        *reinterpret_cast<uint4*>(Ts + off) = rt[q];                                                             \
        *reinterpret_cast<uint4*>(Ss + off) = rs[q];                                                             \
    }

    const int ns = (p.D + SBK - 1) / SBK;
    SIMNN_FETCH(0)
    SIMNN_STASH(0)
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        const int buf = s & 1;
        if (s + 1 < ns) SIMNN_FETCH(s + 1)
        const _Float16* Tb = Ts + buf * ST * SBK;
        const _Float16* Sb = Ss + buf * ST * SBK;
#pragma unroll
        for (int kk = 0; kk < SBK / 16; ++kk) {
            const int chunk = kk * 2 + (lane >> 5);
            f16x8 fs[2], ft[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                fs[x] = *reinterpret_cast<const f16x8*>(Sb + lds_off_halves(ws * 64 + x * 32 + (lane & 31), chunk));
                ft[x] = *reinterpret_cast<const f16x8*>(Tb + lds_off_halves(wt * 64 + x * 32 + (lane & 31), chunk));
            }
            if (do_tn) {
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int e = 0; e < 8; ++e) nrm_t[x] = fmaf((float)ft[x][e], (float)ft[x][e], nrm_t[x]);
            }
            if (do_sn) {
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int e = 0; e < 8; ++e) nrm_s[x] = fmaf((float)fs[x][e], (float)fs[x][e], nrm_s[x]);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    acc[st][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs[st], ft[tt], acc[st][tt], 0, 0, 0);
        }
        if (s + 1 < ns) { SIMNN_STASH(buf ^ 1) }
        __syncthreads();
    }
#undef SIMNN_FETCH
#undef SIMNN_STASH

    if (do_tn) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const float v = nrm_t[x] + __shfl_xor(nrm_t[x], 32);
            const int gi = i0 + wt * 64 + x * 32 + (lane & 31);
            if (lane < 32 && gi < p.N2) p.tnorm2[(long long)b * p.N2 + gi] = v;
        }
    }
    if (do_sn) {
        float m = 0.f;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const float v = nrm_s[x] + __shfl_xor(nrm_s[x], 32);
            const int gj = j0 + ws * 64 + x * 32 + (lane & 31);
            if (gj < p.N1) m = fmaxf(m, v);
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) atomicMax(p.smax2 + b, __float_as_uint(m));
    }

    // acc[st][tt][r] = <src j, tgt i>,  j = j0 + ws*64 + st*32 + (r&3) + 8*(r>>2) + 4*(lane>>5),
    //                                   i = i0 + wt*64 + tt*32 + (lane&31)
    float* sb = reinterpret_cast<float*>(smem);          // [2 ws][128]
    int* sj = reinterpret_cast<int*>(smem) + 2 * 128;
    float* ss = reinterpret_cast<float*>(smem) + 4 * 128;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        float bv = DM_NEG_INF_F32, sv = DM_NEG_INF_F32;
        int bj = DM_IDX_NONE;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + ws * 64 + st * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = (j < p.N1) ? acc[st][tt][r] : DM_NEG_INF_F32;
                // candidates arrive in ascending j: strict > keeps the lowest index on ties
                const bool up = v > bv;
                sv = up ? bv : fmaxf(sv, v);
                bj = up ? j : bj;
                bv = fmaxf(bv, v);
            }
        const float ob = __shfl_xor(bv, 32);
        const int oj = __shfl_xor(bj, 32);
        const float os = __shfl_xor(sv, 32);
        top2_merge(bv, bj, sv, ob, oj, os);
        if (lane < 32) {
            const int li = wt * 64 + tt * 32 + lane;
            sb[ws * 128 + li] = bv; sj[ws * 128 + li] = bj; ss[ws * 128 + li] = sv;
        }
    }
    __syncthreads();
    if (t < 128) {
        const int gi = i0 + t;
        if (gi < p.N2) {
            float bv = sb[t], sv = ss[t];
            int bj = sj[t];
            top2_merge(bv, bj, sv, sb[128 + t], sj[128 + t], ss[128 + t]);
            const long long o = ((long long)b * p.tilesS + ts_) * p.N2pad + gi;
            p.pb[o] = bv; p.pj[o] = bj; p.ps[o] = sv;
        }
    }
}

__global__ __launch_bounds__(256) void simnn_merge_kernel(const float* __restrict__ pb, const int32_t* __restrict__ pj,
                                                          const float* __restrict__ ps, int tilesS, int N2, int N2pad,
                                                          const float* __restrict__ tnorm2, const unsigned int* __restrict__ smax2,
                                                          float tau_scale, int32_t* __restrict__ nn, float* __restrict__ best,
                                                          float* __restrict__ margin, int32_t* __restrict__ flag_count,
                                                          int32_t* __restrict__ flag_list, float* __restrict__ flag_thr) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N2) return;
    float bv = DM_NEG_INF_F32, sv = DM_NEG_INF_F32;
    int bj = DM_IDX_NONE;
    for (int ts = 0; ts < tilesS; ++ts) {
        const long long o = ((long long)b * tilesS + ts) * N2pad + i;
        top2_merge(bv, bj, sv, pb[o], pj[o], ps[o]);
    }
    const long long o = (long long)b * N2 + i;
    nn[o] = (bj == DM_IDX_NONE) ? 0 : bj;
    const float m = bv - sv;
    if (best) best[o] = bv;
    if (margin) margin[o] = m;
    const float tau = tau_scale * sqrtf(tnorm2[o] * __uint_as_float(smax2[b]));
    if (!(m > tau)) {
        const int pos = atomicAdd(flag_count, 1);
        flag_list[pos] = (int32_t)o;
        flag_thr[pos] = bv - tau;          // candidates scoring below this (in fp32) cannot be the float64 argmax
    }
}

// float64 re-evaluation of the flagged rows: one workgroup per flagged row (grid-stride over the list).  Only the
// source tiles whose fp32 tile maximum reaches (best - tau) can contain the float64 argmax (every fp32 score is
// within tau/2 of the exact one); their 128 candidates are re-scored exactly: fp16 products are exact in f64.
__global__ __launch_bounds__(256) void simnn_fixup_kernel(const _Float16* __restrict__ Ftgt, const _Float16* __restrict__ Fsrc,
                                                          int N2, int N1, int D, const float* __restrict__ pb, int tilesS,
                                                          int N2pad, const int32_t* __restrict__ flag_count,
                                                          const int32_t* __restrict__ flag_list,
                                                          const float* __restrict__ flag_thr, int32_t* __restrict__ nn) {
    extern __shared__ __attribute__((aligned(16))) double trow[];   // D doubles + 4 (value) + 4 ints
    double* wv = trow + D;
    int* wj = reinterpret_cast<int*>(wv + 4);
    const int count = *flag_count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = blockIdx.x; e < count; e += gridDim.x) {
        const int o = flag_list[e];
        const float thr = flag_thr[e];
        const int b = o / N2, i = o - b * N2;
        const _Float16* tr = Ftgt + ((long long)b * N2 + i) * D;
        __syncthreads();
        for (int k = threadIdx.x; k < D; k += 256) trow[k] = (double)tr[k];
        __syncthreads();
        double bv = -DM_INF_F64;
        int bj = DM_IDX_NONE;
        for (int ts = 0; ts < tilesS; ++ts) {
            const float tb = pb[((long long)b * tilesS + ts) * N2pad + i];
            if (!(tb >= thr)) continue;                       // uniform: every thread reads the same word
            // wave w re-scores candidates [ts*128 + w*32, +32) in ascending order
            for (int q = 0; q < 32; ++q) {
                const int j = ts * ST + wave * 32 + q;
                if (j >= N1) break;
                const _Float16* sr = Fsrc + ((long long)b * N1 + j) * D;
                double s = 0.0;
                for (int k = lane * 8; k < D; k += 512) {
                    const f16x8 v = *reinterpret_cast<const f16x8*>(sr + k);
#pragma unroll
                    for (int u = 0; u < 8; ++u) s = fma((double)v[u], trow[k + u], s);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
                if (s > bv) { bv = s; bj = j; }               // ascending j within the wave: strict keeps the lowest
            }
        }
        if (lane == 0) { wv[wave] = bv; wj[wave] = bj; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double v = wv[0];
            int j = wj[0];
            for (int w = 1; w < 4; ++w) argmax_merge(v, j, wv[w], wj[w]);   // index tie-break: waves interleave tiles
            if (j != DM_IDX_NONE) nn[o] = j;
        }
    }
}

static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

extern "C" int dm_simnn_f16(dm_ctx* ctx, int B, int N2, int N1, int D, const void* Ftgt, const void* Fsrc, int32_t* nn21,
                            float* best, float* margin) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N2 > 0 && N1 > 0 && D > 0, "sizes must be positive");
    DM_REQUIRE(ctx, Ftgt && Fsrc && nn21, "null pointer");
    DM_REQUIRE(ctx, D % 8 == 0, "D must be a multiple of 8 (16-byte fp16 rows)");
    DM_REQUIRE(ctx, D <= 16384, "D too large for the float64 fix-up row buffer");
    DM_REQUIRE(ctx, (((uintptr_t)Ftgt | (uintptr_t)Fsrc) & 15) == 0, "feature pointers must be 16-byte aligned");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));

    simnn_params p;
    p.Ftgt = (const _Float16*)Ftgt; p.Fsrc = (const _Float16*)Fsrc;
    p.N2 = N2; p.N1 = N1; p.D = D; p.N2pad = pad_to(N2, ST);
    p.tilesT = p.N2pad / ST; p.tilesS = dm_cdiv(N1, ST);
    p.total = B * p.tilesT * p.tilesS;
    const size_t np = (size_t)B * p.tilesS * p.N2pad;
    const size_t need = 3 * dm_align_up(np * 4) + dm_align_up((size_t)B * N2 * 4) * 3 + dm_align_up((size_t)B * 4) + 8192;
    int rc = dm_ws_reserve(ctx, need);
    if (rc) return rc;
    p.pb = (float*)dm_ws_take(ctx, np * 4);
    p.pj = (int32_t*)dm_ws_take(ctx, np * 4);
    p.ps = (float*)dm_ws_take(ctx, np * 4);
    p.tnorm2 = (float*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    int32_t* flag_list = (int32_t*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    float* flag_thr = (float*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    p.smax2 = (unsigned int*)dm_ws_take(ctx, (size_t)B * 4);
    int32_t* flag_count = (int32_t*)dm_ws_take(ctx, 256);

    DM_CHECK_HIP(ctx, hipMemsetAsync(p.smax2, 0, (size_t)B * 4, ctx->stream));
    DM_CHECK_HIP(ctx, hipMemsetAsync(flag_count, 0, 4, ctx->stream));
    DM_LAUNCH(ctx, "simnn_f16_mfma", simnn_kernel, dim3(p.total), dim3(256), 0, p);
    // twice the fp32 accumulation bound: D exact products, D (1 + 1/16) additions, unit roundoff 2^-23
    // (safe for round-to-nearest and for truncating adders), 1 % slack for the fp32 norms
    const float tau_scale = 2.0f * 1.01f * (float)D * (1.0f + 1.0f / 16.0f) * 1.1920929e-7f;
    DM_LAUNCH(ctx, "simnn_merge", simnn_merge_kernel, dim3(dm_cdiv(N2, 256), B), dim3(256), 0, p.pb, p.pj, p.ps, p.tilesS, N2,
              p.N2pad, p.tnorm2, p.smax2, tau_scale, nn21, best, margin, flag_count, flag_list, flag_thr);
    const size_t lds = (size_t)D * 8 + 64;
    static size_t lds_set = 0;
    if (lds > 65536 && lds > lds_set) {
        DM_CHECK_HIP(ctx, hipFuncSetAttribute((const void*)simnn_fixup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)lds));
        lds_set = lds;
    }
    DM_LAUNCH(ctx, "simnn_fixup_f64", simnn_fixup_kernel, dim3(2048), dim3(256), lds, (const _Float16*)Ftgt,
              (const _Float16*)Fsrc, N2, N1, D, p.pb, p.tilesS, p.N2pad, flag_count, flag_list, flag_thr, nn21);
    return DM_OK;
}

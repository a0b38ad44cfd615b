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
#include <stdlib.h>

#include "dm_device.h"
#include "dm_internal.h"

constexpr int ST = 256;    // tile: 256 target rows x 256 source rows per workgroup
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
    float* pb32;                             // (B, N1pad/32, N2pad) fp32 maximum over each block of 32 source rows (fix-up filter)
    int nsub;                                // N1pad / 32
    float* tnorm2;                           // (B, N2)  |t_i|^2, written by the workgroups of source tile 0
    unsigned int* smax2;                     // (B)      max_j |s_j|^2 as float bits (atomicMax), by target tile 0
    int N2, N1, D, N2pad, tilesT, tilesS, total;
    int ldT, ldS;                            // row strides (halves) of Ftgt / Fsrc, >= D, multiples of 8
    int dbg;      // experiments only (env DM_SIMNN_DEBUG): 1 = skip the epilogue, 2 = one K stage only, 3 = no norms
};

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 struct ends up in scratch as a staging array)

// sum of squares of 8 halves with v_dot2_f32_f16 (fp32 accumulate)
__device__ __forceinline__ float sumsq8(f16x8 v, float acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const f16x2 h = {v[2 * e], v[2 * e + 1]};
        acc = __builtin_amdgcn_fdot2(h, h, acc, false);
    }
    return acc;
}

// Shared epilogue: row norms, top-2 reduction of the accumulators over the tile's 256 source rows, 32-row block
// maxima for the fix-up filter.  Reuses the start of the staging LDS as scratch (callers synchronise before).
//   acc[st][tt][r] = <src j, tgt i>,  j = j0 + wsrc*128 + st*32 + (r&3) + 8*(r>>2) + 4*(lane>>5),
//                                     i = i0 + wtgt*64 + tt*32 + (lane&31)
template <bool FULL>
__device__ __forceinline__ void simnn_tail(const simnn_params& p, f32x16 (&acc)[4][2], float (&nrm_t)[2], float (&nrm_s)[4],
                                           bool do_tn, bool do_sn, int b, int i0, int j0, int ts_, _Float16* smem) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wsrc = wave & 1, wtgt = wave >> 1;
    if (do_tn) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const float v = nrm_t[x] + __shfl_xor(nrm_t[x], 32);
            const int gi = i0 + wtgt * 64 + x * 32 + (lane & 31);
            if (lane < 32 && gi < p.N2) p.tnorm2[(long long)b * p.N2 + gi] = v;
        }
    }
    if (do_sn) {
        float m = 0.f;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float v = nrm_s[x] + __shfl_xor(nrm_s[x], 32);
            const int gj = j0 + wsrc * 128 + x * 32 + (lane & 31);
            if (gj < p.N1) m = fmaxf(m, v);
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) atomicMax(p.smax2 + b, __float_as_uint(m));
    }

    // acc[st][tt][r] = <src j, tgt i>,  j = j0 + wsrc*128 + st*32 + (r&3) + 8*(r>>2) + 4*(lane>>5),
    //                                   i = i0 + wtgt*64 + tt*32 + (lane&31)
    float* sb = reinterpret_cast<float*>(smem);          // [2 wsrc][256]
    int* sj = reinterpret_cast<int*>(smem) + 2 * ST;
    float* ss = reinterpret_cast<float*>(smem) + 4 * ST;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        float bv = DM_NEG_INF_F32, sv = DM_NEG_INF_F32;
        int bj = DM_IDX_NONE;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            float m32 = DM_NEG_INF_F32;                  // maximum over this block of 32 source rows
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + wsrc * 128 + st * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = (FULL || j < p.N1) ? acc[st][tt][r] : DM_NEG_INF_F32;
                // candidates arrive in ascending j: strict > keeps the lowest index on ties
                const bool up = v > bv;
                sv = up ? bv : fmaxf(sv, v);
                bj = up ? j : bj;
                bv = fmaxf(bv, v);
                m32 = fmaxf(m32, v);
            }
            m32 = fmaxf(m32, __shfl_xor(m32, 32));
            const int gi32 = i0 + wtgt * 64 + tt * 32 + (lane & 31);
            if (lane < 32 && gi32 < p.N2)
                p.pb32[((long long)b * p.nsub + (j0 >> 5) + wsrc * 4 + st) * p.N2pad + gi32] = m32;
        }
        const float ob = __shfl_xor(bv, 32);
        const int oj = __shfl_xor(bj, 32);
        const float os = __shfl_xor(sv, 32);
        top2_merge(bv, bj, sv, ob, oj, os);
        if (lane < 32) {
            const int li = wtgt * 64 + tt * 32 + lane;
            sb[wsrc * ST + li] = bv; sj[wsrc * ST + li] = bj; ss[wsrc * ST + li] = sv;
        }
    }
    __syncthreads();
    if (t < ST) {
        const int gi = i0 + t;
        if (gi < p.N2) {
            float bv = sb[t], sv = ss[t];
            int bj = sj[t];
            top2_merge(bv, bj, sv, sb[ST + t], sj[ST + t], ss[ST + t]);
            const long long o = ((long long)b * p.tilesS + ts_) * p.N2pad + gi;
            p.pb[o] = bv; p.pj[o] = bj; p.ps[o] = sv;
        }
    }
}

// FULL: every workgroup tile is interior and D is a multiple of the stage depth -> the main loop carries no
// bounds checks and no address arithmetic beyond two pointer bumps.
//
// Tile = 256 target rows x 256 source rows per 512-thread workgroup (8 waves = 2 source halves x 4 target quarters,
// each wave 128 source x 64 target = 4 x 2 MFMA tiles, 128 accumulator registers).  A 128 x 128 tile moves
// 64 flop per L2 byte, which at the fp16 MFMA rate asks the L2 for more than it can deliver; 256 x 256 halves that.
template <bool FULL>
__global__ __launch_bounds__(512, 2) void simnn_kernel(simnn_params p) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];             // T[2] | S[2], 2 x 2 x 32 KiB
    _Float16* Ts = smem;
    _Float16* Ss = smem + 2 * ST * SBK;

    const int id = xcd_remap(blockIdx.x, p.total);
    const int tiles = p.tilesT * p.tilesS;
    const int b = id / tiles;
    const int tts = id - b * tiles;
    const int tt_ = tts / p.tilesS, ts_ = tts - tt_ * p.tilesS;
    const int i0 = tt_ * ST, j0 = ts_ * ST;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wsrc = wave & 1, wtgt = wave >> 1;

    const _Float16* T = p.Ftgt + (long long)b * p.N2 * p.ldT;
    const _Float16* S = p.Fsrc + (long long)b * p.N1 * p.ldS;

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    // squared row norms for the exactness bound, accumulated from the MFMA fragments by the workgroups that
    // own the first tile of the other operand (every row of T / S is seen exactly once that way)
    const bool do_tn = (ts_ == 0) && (wsrc == 0) && p.dbg != 3;
    const bool do_sn = (tt_ == 0) && (wtgt == 0) && p.dbg != 3;
    float nrm_t[2] = {0.f, 0.f}, nrm_s[4] = {0.f, 0.f, 0.f, 0.f};

    const int lrow = t >> 3, lchunk = t & 7;                                   // staging: rows q*64 + lrow, 16-byte chunk lchunk
    const _Float16* tptr = T + (long long)(i0 + lrow) * p.ldT + lchunk * 8;
    const _Float16* sptr = S + (long long)(j0 + lrow) * p.ldS + lchunk * 8;
    u32x4 rt[4], rs[4];
    // (macros, not lambdas: by-reference lambda captures of the staging arrays end up in scratch)
#define SIMNN_FETCH(s_)                                                                                          \
    {                                                                                                            \
        if (FULL) {                                                                                              \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
                rt[q] = *reinterpret_cast<const u32x4*>(tptr + (long long)q * 64 * p.ldT + (s_) * SBK);         \
                rs[q] = *reinterpret_cast<const u32x4*>(sptr + (long long)q * 64 * p.ldS + (s_) * SBK);         \
            }                                                                                                    \
        } else {                                                                                                 \
            const int k_ = (s_) * SBK + lchunk * 8;                                                              \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
                const int row = q * 64 + lrow;                                                                   \
                const int gi = i0 + row, gj = j0 + row;                                                          \
                rt[q] = (gi < p.N2 && k_ < p.D) ? *reinterpret_cast<const u32x4*>(T + (long long)gi * p.ldT + k_) \
                                                : u32x4{0, 0, 0, 0};                                             \
                rs[q] = (gj < p.N1 && k_ < p.D) ? *reinterpret_cast<const u32x4*>(S + (long long)gj * p.ldS + k_) \
                                                : u32x4{0, 0, 0, 0};                                             \
            }                                                                                                    \
        }                                                                                                        \
    }
#define SIMNN_STASH(buf_)                                                                                        \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
        const int row = q * 64 + lrow;                                                                           \
        const int off = (buf_) * ST * SBK + lds_off_halves(row, lchunk);                                         \
        *reinterpret_cast<u32x4*>(Ts + off) = rt[q];                                                             \
        *reinterpret_cast<u32x4*>(Ss + off) = rs[q];                                                             \
    }

    const int ns = (p.dbg == 2) ? 1 : (p.D + SBK - 1) / SBK;
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
            f16x8 fs[4], ft[2];
#pragma unroll
            for (int x = 0; x < 4; ++x)
                fs[x] = *reinterpret_cast<const f16x8*>(Sb + lds_off_halves(wsrc * 128 + x * 32 + (lane & 31), chunk));
#pragma unroll
            for (int x = 0; x < 2; ++x)
                ft[x] = *reinterpret_cast<const f16x8*>(Tb + lds_off_halves(wtgt * 64 + x * 32 + (lane & 31), chunk));
            if (do_tn) {
#pragma unroll
                for (int x = 0; x < 2; ++x) nrm_t[x] = sumsq8(ft[x], nrm_t[x]);
            }
            if (do_sn) {
#pragma unroll
                for (int x = 0; x < 4; ++x) nrm_s[x] = sumsq8(fs[x], nrm_s[x]);
            }
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    acc[st][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs[st], ft[tt], acc[st][tt], 0, 0, 0);
        }
        if (s + 1 < ns) { SIMNN_STASH(buf ^ 1) }
        __syncthreads();
    }
#undef SIMNN_FETCH
#undef SIMNN_STASH

    if (p.dbg == 1) {
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[a][c][r];
        if (sacc == 1.2345f) p.pb[0] = sacc;
        return;
    }
    simnn_tail<FULL>(p, acc, nrm_t, nrm_s, do_tn, do_sn, b, i0, j0, ts_, smem);
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Interior tiles, D % 64 == 0: same tiling as simnn_kernel, operands staged by LDS-DMA.
// EXP (experiments, env DM_SIMNN_EXP with DM_SIMNN_PIPE=0): 0 = product kernel; 3 = MFMA only (no LDS-DMA, no fragment
// reads in the loop); 7 = LDS-DMA only (no MFMA, no fragment reads).  3 and 7 give wrong results and exist to bound
// the main loop from both sides.
template <int EXP>
__global__ __launch_bounds__(512, 2) void simnn_glds_kernel(simnn_params p) {
    constexpr bool FULL = true;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];             // T[2] | S[2], 2 x 2 x 32 KiB
    _Float16* Ts = smem;
    _Float16* Ss = smem + 2 * ST * SBK;

    const int id = xcd_remap(blockIdx.x, p.total);
    const int tiles = p.tilesT * p.tilesS;
    const int b = id / tiles;
    const int tts = id - b * tiles;
    const int tt_ = tts / p.tilesS, ts_ = tts - tt_ * p.tilesS;
    const int i0 = tt_ * ST, j0 = ts_ * ST;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wsrc = wave & 1, wtgt = wave >> 1;

    const _Float16* T = p.Ftgt + (long long)b * p.N2 * p.ldT;
    const _Float16* S = p.Fsrc + (long long)b * p.N1 * p.ldS;

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    // squared row norms for the exactness bound, accumulated from the MFMA fragments by the workgroups that
    // own the first tile of the other operand (every row of T / S is seen exactly once that way)
    const bool do_tn = (ts_ == 0) && (wsrc == 0) && p.dbg != 3;
    const bool do_sn = (tt_ == 0) && (wtgt == 0) && p.dbg != 3;
    float nrm_t[2] = {0.f, 0.f}, nrm_s[4] = {0.f, 0.f, 0.f, 0.f};

    // Staging by LDS-DMA (global_load_lds, 16 B per lane): one instruction fills 8 consecutive 128-byte rows of the
    // LDS image (wave-uniform base + lane * 16).  The image is swizzled (chunk c of row r lives in slot
    // c ^ ((r >> 1) & 7)), and since the DMA destination is lane-linear the swizzle is applied to the per-lane
    // SOURCE address: lane l fills slot (l & 7) of row (l >> 3), so it fetches chunk (l & 7) ^ ((row >> 1) & 7).
    // Wave w stages row groups 4w .. 4w+3 (8 rows each) of both operands: 8 DMA instructions per stage, no VGPRs,
    // no ds_write.
    const int grow = lane >> 3;
    const _Float16* tsrc[4];
    const _Float16* ssrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (wave * 4 + q) * 8 + grow;
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        tsrc[q] = T + (long long)(i0 + row) * p.ldT + chunk * 8;
        ssrc[q] = S + (long long)(j0 + row) * p.ldS + chunk * 8;
    }
#define SIMNN_DMA(s_, buf_)                                                                                      \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
        const int off = (buf_) * ST * SBK + (wave * 4 + q) * 8 * SBK;                                            \
        __builtin_amdgcn_global_load_lds((gptr_t)(tsrc[q] + (s_) * SBK), (lptr_t)(Ts + off), 16, 0, 0);         \
        __builtin_amdgcn_global_load_lds((gptr_t)(ssrc[q] + (s_) * SBK), (lptr_t)(Ss + off), 16, 0, 0);         \
    }

    const int ns = (p.dbg == 2) ? 1 : p.D / SBK;
    SIMNN_DMA(0, 0)
    __syncthreads();                       // (the barrier's release waits for the outstanding LDS-DMA: vmcnt(0))
    // the k loop exists twice: the few workgroups that also accumulate row norms take the second copy, so the hot
    // copy has no conditional inside a k-step (a branch there splits the basic block and stops the compiler from
    // interleaving the next ds_reads with the MFMAs)
    f16x8 fs[4], ft[2];
#define SIMNN_KLOOP(NORMS)                                                                                             \
    for (int s = 0; s < ns; ++s) {                                                                                     \
        const int buf = s & 1;                                                                                         \
        if (s + 1 < ns && EXP != 3) { SIMNN_DMA(s + 1, buf ^ 1) }                                                      \
        const _Float16* Tb = Ts + buf * ST * SBK;                                                                      \
        const _Float16* Sb = Ss + buf * ST * SBK;                                                                      \
        _Pragma("unroll") for (int kk = 0; kk < SBK / 16; ++kk) {                                                      \
            const int chunk = kk * 2 + (lane >> 5);                                                                    \
            if ((EXP != 3 && EXP != 7) || (s == 0 && kk == 0)) {                                                       \
            _Pragma("unroll") for (int x = 0; x < 4; ++x)                                                              \
                fs[x] = *reinterpret_cast<const f16x8*>(Sb + lds_off_halves(wsrc * 128 + x * 32 + (lane & 31), chunk)); \
            _Pragma("unroll") for (int x = 0; x < 2; ++x)                                                              \
                ft[x] = *reinterpret_cast<const f16x8*>(Tb + lds_off_halves(wtgt * 64 + x * 32 + (lane & 31), chunk));  \
            }                                                                                                          \
            if (NORMS) {                                                                                               \
                if (do_tn) { _Pragma("unroll") for (int x = 0; x < 2; ++x) nrm_t[x] = sumsq8(ft[x], nrm_t[x]); }       \
                if (do_sn) { _Pragma("unroll") for (int x = 0; x < 4; ++x) nrm_s[x] = sumsq8(fs[x], nrm_s[x]); }       \
            }                                                                                                          \
            if (EXP != 7 || s == 0) {                                                                                  \
            _Pragma("unroll") for (int st = 0; st < 4; ++st)                                                           \
                _Pragma("unroll") for (int tt = 0; tt < 2; ++tt)                                                       \
                    acc[st][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs[st], ft[tt], acc[st][tt], 0, 0, 0);        \
            }                                                                                                          \
        }                                                                                                              \
        __syncthreads();                                                                                               \
    }
    if ((ts_ == 0 || tt_ == 0) && p.dbg != 3) { SIMNN_KLOOP(true) } else { SIMNN_KLOOP(false) }
#undef SIMNN_KLOOP
#undef SIMNN_DMA

    if (p.dbg == 1) {
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[a][c][r];
        if (sacc == 1.2345f) p.pb[0] = sacc;
        return;
    }
    simnn_tail<FULL>(p, acc, nrm_t, nrm_s, do_tn, do_sn, b, i0, j0, ts_, smem);
}

// Deep-pipelined variant (default for interior tiles, D % 32 == 0).  Same 256 x 256 tile and wave layout as
// simnn_glds_kernel, but the contraction is staged 32 halves at a time through a ring of FOUR 32 KiB LDS buffers and
// the LDS-DMA of stage s+3 is issued while stage s is computed: a first-touch miss (HBM / Infinity Cache, ~2 us under
// load; every tile has some because one pair's operands, 6 MB, exceed an XCD's 4 MB L2) then has three stages to
// land instead of one.  The wait before each barrier is a COUNTED vmcnt (the two younger stages stay in flight).
//
// LDS image of one operand stage: 256 rows x 64 B; chunk c (16 B) of row r lives in slot c ^ ((r >> 2) & 3), so the
// 16 rows of a ds_read_b128 lane group (four runs of 4 consecutive rows with distinct (r >> 2) & 3) cover all 16
// slots of the 256-byte bank row.  One DMA instruction fills 16 rows (lane l -> row l >> 2, slot l & 3), the swizzle
// is applied on the source address.
constexpr int PBK = 32;                    // halves per stage
constexpr int PNBUF = 4;                   // ring depth
constexpr int PSTAGE = 2 * ST * PBK;       // halves per ring slot: T image then S image
#define DM_WAITCNT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15))     /* vmcnt(n), lgkmcnt / expcnt untouched */
template <int EXP>
__global__ __launch_bounds__(512, 2) void simnn_pipe_kernel(simnn_params p) {
    constexpr bool FULL = true;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];             // 4 x (T 16 KiB | S 16 KiB)

    const int id = xcd_remap(blockIdx.x, p.total);
    const int tiles = p.tilesT * p.tilesS;
    const int b = id / tiles;
    const int tts = id - b * tiles;
    const int tt_ = tts / p.tilesS, ts_ = tts - tt_ * p.tilesS;
    const int i0 = tt_ * ST, j0 = ts_ * ST;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wsrc = wave & 1, wtgt = wave >> 1;

    const _Float16* T = p.Ftgt + (long long)b * p.N2 * p.ldT;
    const _Float16* S = p.Fsrc + (long long)b * p.N1 * p.ldS;

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    const bool do_tn = (ts_ == 0) && (wsrc == 0) && p.dbg != 3;
    const bool do_sn = (tt_ == 0) && (wtgt == 0) && p.dbg != 3;
    float nrm_t[2] = {0.f, 0.f}, nrm_s[4] = {0.f, 0.f, 0.f, 0.f};

    // wave w stages rows 32w .. 32w+31 of both operands: 4 DMA instructions per stage
    const _Float16* tsrc[2];
    const _Float16* ssrc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = wave * 32 + q * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        tsrc[q] = T + (long long)(i0 + row) * p.ldT + chunk * 8;
        ssrc[q] = S + (long long)(j0 + row) * p.ldS + chunk * 8;
    }
#define SIMNN_DMA1(s_, q)                                                                                        \
    {                                                                                                            \
        _Float16* dst = smem + ((s_) & (PNBUF - 1)) * PSTAGE + (wave * 32 + (q) * 16) * PBK;                     \
        __builtin_amdgcn_global_load_lds((gptr_t)(tsrc[q] + (kbase + kstep * (s_)) * PBK), (lptr_t)dst, 16, 0, 0);                \
        __builtin_amdgcn_global_load_lds((gptr_t)(ssrc[q] + (kbase + kstep * (s_)) * PBK), (lptr_t)(dst + ST * PBK), 16, 0, 0);   \
    }
#define SIMNN_DMA(s_) SIMNN_DMA1(s_, 0) SIMNN_DMA1(s_, 1)

    // fragment addresses: row (lane & 31) of a 32-row block, chunk kk*2 + (lane >> 5); the swizzle term depends on
    // the lane only, and kk = 1 flips bit 1 of the slot
    const int swz = (lane >> 2) & 3;
    const int c0 = (lane >> 5) ^ swz;
    const int frow = (lane & 31) * PBK;
    const int foff0 = frow + (c0 << 3), foff1 = frow + ((c0 ^ 2) << 3);
    const int sbase = ST * PBK + wsrc * 128 * PBK, tbase = wtgt * 64 * PBK;

    const int ns = (p.dbg == 2) ? 3 : p.D / PBK;                 // >= 3 (host checks D >= 96)
    // The 32 workgroups of an XCD sweep the contraction roughly in step; consecutive waves of 32 tiles alternate the
    // sweep direction, so a wave starts on the K chunks its predecessor touched last (still in the 4 MB L2) when they
    // share operand panels -- a pair's panels (6 MB) do not fit, and with one direction the LRU has always just
    // evicted the chunk that is needed next.  (Any order gives the same exact result: ties go to the fix-up.)
    const bool krev = ((tts >> 5) & 1) != 0;
    const int kbase = krev ? ns - 1 : 0, kstep = krev ? -1 : 1;
    SIMNN_DMA(0)
    SIMNN_DMA(1)
    SIMNN_DMA(2)
    DM_WAITCNT_VM(8);                                            // stage 0 has landed; stages 1, 2 in flight
    __builtin_amdgcn_s_barrier();

    // The loop is rotated by half a stage: the barrier that publishes stage s+1 sits between the two k-steps of
    // stage s, so the first fragments of stage s+1 are fetched under the MFMAs of (s, kk=1) and no fragment read is
    // exposed after a barrier.  Ring slot (s+3)&3 == (s-1)&3 was last read by the (s-1, kk=1) fragments, complete
    // (lgkmcnt(0)) before the barrier of iteration s-1, which every wave has left before iteration s starts.
    f16x8 fsa[4], fta[2], fsb[4], ftb[2];
#define SIMNN_READ(fs_, ft_, s_, fo_)                                                                                  \
    if (EXP != 7 || (s_) == 0) {                                                                                       \
        const _Float16* Bs = smem + ((s_) & (PNBUF - 1)) * PSTAGE;                                                     \
        _Pragma("unroll") for (int x = 0; x < 4; ++x)                                                                  \
            fs_[x] = *reinterpret_cast<const f16x8*>(Bs + sbase + x * 32 * PBK + (fo_));                               \
        _Pragma("unroll") for (int x = 0; x < 2; ++x)                                                                  \
            ft_[x] = *reinterpret_cast<const f16x8*>(Bs + tbase + x * 32 * PBK + (fo_));                               \
    }
#define SIMNN_MMA(fs_, ft_, NORMS)                                                                                     \
    if (EXP != 7 || s == 0) {                                                                                          \
        if (NORMS) {                                                                                                   \
            if (do_tn) { _Pragma("unroll") for (int x = 0; x < 2; ++x) nrm_t[x] = sumsq8(ft_[x], nrm_t[x]); }          \
            if (do_sn) { _Pragma("unroll") for (int x = 0; x < 4; ++x) nrm_s[x] = sumsq8(fs_[x], nrm_s[x]); }          \
        }                                                                                                              \
        _Pragma("unroll") for (int st = 0; st < 4; ++st)                                                               \
            _Pragma("unroll") for (int tt = 0; tt < 2; ++tt)                                                           \
                acc[st][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs_[st], ft_[tt], acc[st][tt], 0, 0, 0);          \
    }
#define SIMNN_SYNC(n_)                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    __builtin_amdgcn_s_waitcnt(0x0070 | ((n_) & 15));            /* vmcnt(n) lgkmcnt(0) */                             \
    __builtin_amdgcn_s_barrier();                                                                                      \
    __builtin_amdgcn_sched_barrier(0);
#define SIMNN_KLOOP(NORMS)                                                                                             \
    {                                                                                                                  \
        int s = 0;                                                                                                     \
        SIMNN_READ(fsa, fta, 0, foff0)                                                                                 \
        for (; s < ns - 3; ++s) {                                                                                      \
            SIMNN_READ(fsb, ftb, s, foff1)                                                                             \
            SIMNN_DMA1(s + 3, 0)                                                                                       \
            SIMNN_MMA(fsa, fta, NORMS)                                                                                 \
            SIMNN_SYNC(6)                                                                                              \
            SIMNN_READ(fsa, fta, s + 1, foff0)                                                                         \
            SIMNN_DMA1(s + 3, 1)                                                                                       \
            SIMNN_MMA(fsb, ftb, NORMS)                                                                                 \
        }                                                                                                              \
        SIMNN_READ(fsb, ftb, s, foff1)                                                                                 \
        SIMNN_MMA(fsa, fta, NORMS)                                                                                     \
        SIMNN_SYNC(4)                                                                                                  \
        SIMNN_READ(fsa, fta, s + 1, foff0)                                                                             \
        SIMNN_MMA(fsb, ftb, NORMS)                                                                                     \
        ++s;                                                                                                           \
        SIMNN_READ(fsb, ftb, s, foff1)                                                                                 \
        SIMNN_MMA(fsa, fta, NORMS)                                                                                     \
        SIMNN_SYNC(0)                                                                                                  \
        SIMNN_READ(fsa, fta, s + 1, foff0)                                                                             \
        SIMNN_MMA(fsb, ftb, NORMS)                                                                                     \
        ++s;                                                                                                           \
        SIMNN_READ(fsb, ftb, s, foff1)                                                                                 \
        SIMNN_MMA(fsa, fta, NORMS)                                                                                     \
        SIMNN_MMA(fsb, ftb, NORMS)                                                                                     \
    }
    if ((ts_ == 0 || tt_ == 0) && p.dbg != 3) SIMNN_KLOOP(true) else SIMNN_KLOOP(false)
#undef SIMNN_KLOOP
#undef SIMNN_SYNC
#undef SIMNN_MMA
#undef SIMNN_READ
#undef SIMNN_DMA1
#undef SIMNN_DMA
    __syncthreads();                                             // the epilogue reuses the ring as scratch

    if (p.dbg == 1) {
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[a][c][r];
        if (sacc == 1.2345f) p.pb[0] = sacc;
        return;
    }
    simnn_tail<FULL>(p, acc, nrm_t, nrm_s, do_tn, do_sn, b, i0, j0, ts_, smem);
}


__global__ __launch_bounds__(256) void simnn_merge_kernel(const float* __restrict__ pb, const int32_t* __restrict__ pj,
                                                          const float* __restrict__ ps, int tilesS, int N2, int N2pad,
                                                          const float* __restrict__ tnorm2, const unsigned int* __restrict__ smax2,
                                                          float tau_scale, int32_t* __restrict__ nn, float* __restrict__ best,
                                                          float* __restrict__ margin, int32_t* __restrict__ flag_count,
                                                          int32_t* __restrict__ flag_list, float* __restrict__ flag_thr,
                                                          const int32_t* __restrict__ force_flag) {
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
    const bool forced = force_flag && force_flag[b] != 0;   // the caller could not bound the error for this pair: re-score everything
    if (forced || !(m > tau)) {
        const int pos = atomicAdd(flag_count, 1);
        flag_list[pos] = (int32_t)o;
        flag_thr[pos] = forced ? DM_NEG_INF_F32 : bv - tau;   // candidates scoring below this (in fp32) cannot be the float64 argmax
    }
}

// float64 re-evaluation of the flagged rows: one workgroup per flagged row (grid-stride over the list).  Only the
// blocks of 32 source rows whose fp32 maximum reaches (best - tau) can contain the float64 argmax (every fp32 score
// is within tau/2 of the exact one).  Such a block is re-scored exactly by the whole workgroup: 8 lanes per
// candidate, each wave instruction reads 8 x 128 contiguous bytes (fully used cache lines); fp16 products are exact
// in f64 and the summation order is fixed, so duplicated rows give identical scores and the lowest index wins.
__global__ __launch_bounds__(256) void simnn_fixup_kernel(const _Float16* __restrict__ Ftgt, const _Float16* __restrict__ Fsrc,
                                                          int N2, int N1, int D, const float* __restrict__ pb32, int nsub,
                                                          int N2pad, const int32_t* __restrict__ flag_count,
                                                          const int32_t* __restrict__ flag_list,
                                                          const float* __restrict__ flag_thr, int32_t* __restrict__ nn) {
    extern __shared__ __attribute__((aligned(16))) double trow[];   // D doubles + 4 (value) + 4 ints
    double* wv = trow + D;
    int* wj = reinterpret_cast<int*>(wv + 4);
    const int count = *flag_count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cand = threadIdx.x >> 3, part = threadIdx.x & 7;       // 32 candidates x 8 lanes
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
        for (int sb = 0; sb < nsub; ++sb) {
            const float tb = pb32[((long long)b * nsub + sb) * N2pad + i];
            if (!(tb >= thr)) continue;                       // uniform: every thread reads the same word
            const int j = sb * 32 + cand;
            double sacc = 0.0;
            if (j < N1) {
                const _Float16* sr = Fsrc + ((long long)b * N1 + j) * D;
                for (int k = part * 8; k < D; k += 64) {      // D % 8 == 0 is guaranteed by the caller
                    const f16x8 v = *reinterpret_cast<const f16x8*>(sr + k);
#pragma unroll
                    for (int u = 0; u < 8; ++u) sacc = fma((double)v[u], trow[k + u], sacc);
                }
            }
            sacc += __shfl_xor(sacc, 1);
            sacc += __shfl_xor(sacc, 2);
            sacc += __shfl_xor(sacc, 4);
            if (j < N1 && sacc > bv) { bv = sacc; bj = j; }   // blocks ascend: strict keeps the lowest index
        }
        // (all 8 lanes of a candidate hold the same pair; merge over the candidates of the workgroup)
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            const double ov = __shfl_xor(bv, off);
            const int oj = __shfl_xor(bj, off);
            argmax_merge(bv, bj, ov, oj);
        }
        if (lane == 0) { wv[wave] = bv; wj[wave] = bj; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double v = wv[0];
            int j = wj[0];
            for (int w = 1; w < 4; ++w) argmax_merge(v, j, wv[w], wj[w]);
            if (j != DM_IDX_NONE) nn[o] = j;
        }
    }
}

static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

size_t dm_simnn_ws_bytes(int B, int N2, int N1) {
    const size_t N2pad = pad_to(N2, ST), tilesS = dm_cdiv(N1, ST);
    const size_t np = (size_t)B * tilesS * N2pad, np32 = (size_t)B * tilesS * (ST / 32) * N2pad;
    return 3 * dm_align_up(np * 4) + dm_align_up(np32 * 4) + dm_align_up((size_t)B * N2 * 4) * 3 + dm_align_up((size_t)B * 4) + 8192;
}

// Tile kernel + merge: fp32 scores, top-2 per target row, the rows whose margin is inside the error bound queued for an
// exact re-evaluation by the caller.  Workspace comes from the context arena (the caller reserved dm_simnn_ws_bytes).
// rel_extra: additional relative error of a score (in units of |t_i| max_j |s_j|) on top of the fp32 accumulation bound.
int dm_simnn_core(dm_ctx* ctx, int B, int N2, int N1, int D, const _Float16* Ftgt, int ldT, const _Float16* Fsrc, int ldS,
                  float rel_extra,
                  const int32_t* force_flag, int32_t* nn21, float* best, float* margin, dm_simnn_queue* q) {
    simnn_params p;
    p.Ftgt = Ftgt; p.Fsrc = Fsrc;
    p.N2 = N2; p.N1 = N1; p.D = D; p.N2pad = pad_to(N2, ST);
    p.ldT = ldT; p.ldS = ldS;
    p.tilesT = p.N2pad / ST; p.tilesS = dm_cdiv(N1, ST);
    p.total = B * p.tilesT * p.tilesS;
    { const char* e = getenv("DM_SIMNN_DEBUG"); p.dbg = e ? atoi(e) : 0; }
    const size_t np = (size_t)B * p.tilesS * p.N2pad;
    p.nsub = p.tilesS * (ST / 32);
    const size_t np32 = (size_t)B * p.nsub * p.N2pad;
    p.pb = (float*)dm_ws_take(ctx, np * 4);
    p.pj = (int32_t*)dm_ws_take(ctx, np * 4);
    p.ps = (float*)dm_ws_take(ctx, np * 4);
    p.pb32 = (float*)dm_ws_take(ctx, np32 * 4);
    p.tnorm2 = (float*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    int32_t* flag_list = (int32_t*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    float* flag_thr = (float*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    p.smax2 = (unsigned int*)dm_ws_take(ctx, (size_t)B * 4);
    int32_t* flag_count = (int32_t*)dm_ws_take(ctx, 256);
    if (!p.pb || !p.pj || !p.ps || !p.pb32 || !p.tnorm2 || !flag_list || !flag_thr || !p.smax2 || !flag_count)
        return dm_fail(ctx, DM_ENOMEM, "simnn: workspace not reserved");

    DM_CHECK_HIP(ctx, hipMemsetAsync(p.smax2, 0, (size_t)B * 4, ctx->stream));
    DM_CHECK_HIP(ctx, hipMemsetAsync(flag_count, 0, 4, ctx->stream));
    const size_t lds_main = (size_t)2 * 2 * ST * SBK * sizeof(_Float16);       // 128 KiB
    int rc;
    {
        const void* kernels[] = {(const void*)simnn_glds_kernel<0>, (const void*)simnn_glds_kernel<3>, (const void*)simnn_glds_kernel<7>,
                                 (const void*)simnn_kernel<false>, (const void*)simnn_pipe_kernel<0>, (const void*)simnn_pipe_kernel<7>};
        for (const void* kf : kernels) {
            rc = dm_grant_lds(ctx, kf, lds_main);
            if (rc) return rc;
        }
    }
    const bool interior = (N2 % ST == 0 && N1 % ST == 0);
    const char* pe = getenv("DM_SIMNN_PIPE");                                  // 0: the two-buffer kernel (experiments)
    const int pipe = pe ? atoi(pe) : 1;
    if (interior && pipe && D % PBK == 0 && D >= 3 * PBK) {
        const char* xe = getenv("DM_SIMNN_EXP");
        if (xe && atoi(xe) == 7) DM_LAUNCH(ctx, "simnn_f16_mfma", simnn_pipe_kernel<7>, dim3(p.total), dim3(512), lds_main, p);
        else DM_LAUNCH(ctx, "simnn_f16_mfma", simnn_pipe_kernel<0>, dim3(p.total), dim3(512), lds_main, p);
    }
    else if (interior && D % SBK == 0) {
        const char* xe = getenv("DM_SIMNN_EXP");
        const int ex = xe ? atoi(xe) : 0;
        if (ex == 3) DM_LAUNCH(ctx, "simnn_f16_mfma", simnn_glds_kernel<3>, dim3(p.total), dim3(512), lds_main, p);
        else if (ex == 7) DM_LAUNCH(ctx, "simnn_f16_mfma", simnn_glds_kernel<7>, dim3(p.total), dim3(512), lds_main, p);
        else DM_LAUNCH(ctx, "simnn_f16_mfma", simnn_glds_kernel<0>, dim3(p.total), dim3(512), lds_main, p);
    }
    else
        DM_LAUNCH(ctx, "simnn_f16_mfma", simnn_kernel<false>, dim3(p.total), dim3(512), lds_main, p);
    // twice the error bound of a score, relative to |t_i| max_j |s_j|: fp32 accumulation (D exact products,
    // D (1 + 1/16) additions, unit roundoff 2^-23, safe for round-to-nearest and for truncating adders) + the caller's
    // own term; 1 % slack for the fp32 norms
    const float tau_scale = 2.0f * 1.01f * ((float)D * (1.0f + 1.0f / 16.0f) * 1.1920929e-7f + rel_extra);
    DM_LAUNCH(ctx, "simnn_merge", simnn_merge_kernel, dim3(dm_cdiv(N2, 256), B), dim3(256), 0, p.pb, p.pj, p.ps, p.tilesS, N2,
              p.N2pad, p.tnorm2, p.smax2, tau_scale, nn21, best, margin, flag_count, flag_list, flag_thr, force_flag);
    q->pb32 = p.pb32; q->nsub = p.nsub; q->N2pad = p.N2pad;
    q->flag_count = flag_count; q->flag_list = flag_list; q->flag_thr = flag_thr;
    return DM_OK;
}

extern "C" int dm_simnn_f16(dm_ctx* ctx, int B, int N2, int N1, int D, const void* Ftgt, const void* Fsrc, int32_t* nn21,
                            float* best, float* margin) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N2 > 0 && N1 > 0 && D > 0, "sizes must be positive");
    DM_REQUIRE(ctx, Ftgt && Fsrc && nn21, "null pointer");
    DM_REQUIRE(ctx, D % 8 == 0, "D must be a multiple of 8 (16-byte fp16 rows)");
    DM_REQUIRE(ctx, D <= 16384, "D too large for the float64 fix-up row buffer");
    DM_REQUIRE(ctx, (((uintptr_t)Ftgt | (uintptr_t)Fsrc) & 15) == 0, "feature pointers must be 16-byte aligned");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = dm_ws_reserve(ctx, dm_simnn_ws_bytes(B, N2, N1));
    if (rc) return rc;
    dm_simnn_queue q;
    rc = dm_simnn_core(ctx, B, N2, N1, D, (const _Float16*)Ftgt, D, (const _Float16*)Fsrc, D, 0.0f, nullptr, nn21, best, margin, &q);
    if (rc) return rc;
    const size_t lds = (size_t)D * 8 + 64;
    if (lds > 65536) {
        rc = dm_grant_lds(ctx, (const void*)simnn_fixup_kernel, lds);
        if (rc) return rc;
    }
    DM_LAUNCH(ctx, "simnn_fixup_f64", simnn_fixup_kernel, dim3(2048), dim3(256), lds, (const _Float16*)Ftgt,
              (const _Float16*)Fsrc, N2, N1, D, q.pb32, q.nsub, q.N2pad, q.flag_count, q.flag_list, q.flag_thr, nn21);
    return DM_OK;
}

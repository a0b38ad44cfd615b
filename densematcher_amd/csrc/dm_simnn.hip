// Feature-similarity nearest neighbour (dm_simnn_f16), BASELINE.json config 3.
//
//   nn21[b,i] = argmax_j <Ftgt[b,i,:], Fsrc[b,j,:]>          oracle/dm_oracle.py: simnn
//
// S^T = Fsrc Ftgt^T is produced 256x256 tile by tile on the fp16 matrix cores
// (v_mfma_f32_32x32x16_f16: exact fp16 products, fp32 accumulation) and consumed in registers by
// a top-2 row reduction; S never reaches memory.  The operands are swapped (src is the MFMA "A"
// side) so that each lane owns ONE target row and 16 source candidates per MFMA tile: the
// reduction is in-lane except for one cross-half step.
//
// Exactness: fp32 accumulation can reorder near-ties, and the in-register reduction compares scores whose low 4
// mantissa bits carry the candidate's position.  Every row whose (best - second best) is within
// (2 D (1 + 1/16) 2^-23 + 3 * 2^-19) |t_i| max_j |s_j|  is re-evaluated in float64 (products of fp16 are exact in
// f64, the f64 sum is exact to 1e-16 relative), so the returned index equals the float64 argmax with the
// lowest-index tie rule.
#include <string.h>

#include "dm_device.h"
#include "dm_internal.h"
#include "dm_split.h"

constexpr int ST = 256;    // tile: 256 target rows x 256 source rows per workgroup
constexpr int SBK = 64;    // contraction (halves) per LDS stage of the register-staged kernel
#define DM_KEY_NONE (-3.0e38f)   // finite "no candidate" score: its bit pattern stays finite with position bits OR-ed in

// LDS image of a 128 x 64 fp16 tile: row r is one 128-byte line of eight 16-byte chunks; chunk c is
// stored at slot c ^ ((r >> 1) & 7).  Two consecutive rows fill one 256-byte bank row, so the 16
// rows (distinct mod 16) that one ds_read_b128 lane group touches land on 16 different slots.
__device__ __forceinline__ int lds_off_halves(int row, int chunk) {
    return row * SBK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

struct simnn_params {
    const _Float16* Ftgt; const _Float16* Fsrc;
    float* pb; int32_t* pj; float* ps;       // partials (B, 2 tilesS, N2pad): top-2 of a target row over each half (128 source rows)
                                             // of a tile = the candidates one wave holds; they also prune the exact fix-up (dm_simnn_queue)
    float* tnorm2;                           // (B, N2)  |t_i|^2, written by the workgroups of source tile 0
    unsigned int* smax2;                     // (B)      max_j |s_j|^2 as float bits (atomicMax), by target tile 0
    int N2, N1, D, N2pad, tilesT, tilesS, total;
    int ldT, ldS;                            // row strides (halves) of Ftgt / Fsrc, >= D, multiples of 8
    int rowsT, rowsS;                        // rows per pair of Ftgt / Fsrc and of the per-row term arrays: N2 / N1, or the padded
                                             // counts when the caller's buffers are padded to whole tiles (key-set passes on any size)
    int band;                                // tile rows per band of the tile order (simnn_decode)
    int rt0, rnT, rs0, rnS;                  // the rectangle of tiles (per pair) this launch walks: rows rt0 .. rt0 + rnT - 1 of the
                                             // tilesT x tilesS grid, columns rs0 .. rs0 + rnS - 1; total = B rnT rnS
    // two reductions of the same products (DUAL kernels, dm_knnsplit.hip: dm_launch_fm_split):
    //   key A = score + bias[j]  -> pb / pj / ps;   key B = score * scale[j] (DUAL 1) or score (DUAL 2) -> the *_2 arrays
    const float* bias; const float* scale;   // (B, N1) per source row
    float* pb_2; int32_t* pj_2; float* ps_2;
    // both directions in one pass (DUAL 3): besides the two row reductions above, every SOURCE row j gets two reductions
    // over the targets: key A' = score + biasT[i], key B' = score.  Partials per (tile row, target quarter of the tile = 64
    // targets = one wave): (B, N2pad / 64, N1pad); |s_j|^2 and max_i |t_i|^2 for the bound
    const float* biasT;                      // (B, N2) per target row
    float* cb[2]; int32_t* cj[2]; float* cs[2];
    float* snorm2; unsigned int* tmax2;
    int N1pad;
    int dbg;                                 // DM_EXPERIMENTS builds only (0 in the product): see simnn_pipe_kernel
    unsigned long long* trace;               // DM_EXPERIMENTS builds only: stage timeline of workgroup 0 (XV bit 1024)
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

// single-instruction float helpers for the reduction (the operands are scores with position bits in the low
// mantissa: plain fmaxf would add a canonicalising v_max in front of every use)
__device__ __forceinline__ float k_max(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float k_min(float a, float b) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float k_max3(float a, float b, float c) { float d; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float k_med3(float a, float b, float c) { float d; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// value of the other half-wave's lane (lane ^ 32) with one v_permlane32_swap (VALU) instead of a ds_bpermute round trip
__device__ __forceinline__ unsigned xhalf_u32(unsigned v, bool upper) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // r[0] = [lo | lo], r[1] = [hi | hi]
    return upper ? r[0] : r[1];
}
__device__ __forceinline__ float xhalf(float v, bool upper) { return __uint_as_float(xhalf_u32(__float_as_uint(v), upper)); }
__device__ __forceinline__ int xhalf(int v, bool upper) { return (int)xhalf_u32((unsigned)v, upper); }
// maximum over the two half-waves (keys)
__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(__uint_as_float(r[0])), "v"(__uint_as_float(r[1])));
    return d;
}
// score with its low 4 mantissa bits replaced by `code` (v_and_or_b32); code = 15 - position, so that among equal
// (truncated) positive scores the earliest position is the largest key
__device__ __forceinline__ float k_key(float v, int code) { return __int_as_float((__float_as_int(v) & ~15) | code); }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// the lane id, recomputed where it is used: values derived from the kernel's own `lane` are loop invariants of the tile loop,
// and the register allocator spilled some of them to scratch around the main loop -- every scratch reload then waits
// vmcnt(0), i.e. for the whole LDS-DMA queue.  An asm volatile cannot be hoisted or merged.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// Squared row norms for the exactness bound, written right behind the tile loop that accumulated them (the tiles of source
// tile column 0 / target tile row 0): |t_i|^2 per target row and max_j |s_j|^2 per pair (row direction), and, for a pass in both
// directions (COLS), |s_j|^2 per source row and max_i |t_i|^2.  Kept out of the key epilogues: six more live registers there
// were spilled to scratch, and every scratch reload waits vmcnt(0) -- for the whole LDS-DMA queue of the next tile.
// maximum over the 16 lanes of a DPP row, on the vector ALU (a __shfl needs the lane id: the compiler kept the kernel's own copy of
// it alive across the whole tile loop for that, in scratch).  The four row leaders of a wave then issue one atomic each.
__device__ __forceinline__ float row16_max(float m) {
    m = fmaxf(m, __int_as_float(dpp_i32<0xB1>(__float_as_int(m))));
    m = fmaxf(m, __int_as_float(dpp_i32<0x4E>(__float_as_int(m))));
    m = fmaxf(m, __int_as_float(dpp_i32<0x141>(__float_as_int(m))));
    m = fmaxf(m, __int_as_float(dpp_i32<0x140>(__float_as_int(m))));
    return m;
}
template <bool COLS, int TB>
__device__ __forceinline__ void simnn_norms(const simnn_params& p, float (&nrm_t)[TB], float (&nrm_s)[4], bool do_tn, bool do_sn,
                                            int b, int i0, int j0, int wsrc, int wtgt) {
    const int lane = fresh_lane();
    const int hi = lane >> 5;
    if (do_tn) {
        float m = 0.f;
#pragma unroll
        for (int x = 0; x < TB; ++x) {
            const float v = nrm_t[x] + xhalf(nrm_t[x], hi != 0);
            const int gi = i0 + wtgt * (32 * TB) + x * 32 + (lane & 31);
            if (lane < 32 && gi < p.N2) p.tnorm2[(long long)b * p.N2 + gi] = v;
            m = fmaxf(m, v);                             // (rows beyond N2 are zero rows of padded operands)
        }
        if (COLS) {
            m = row16_max(m);
            if ((lane & 15) == 0) atomicMax(p.tmax2 + b, __float_as_uint(m));
        }
    }
    if (do_sn) {
        float m = 0.f;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float v = nrm_s[x] + xhalf(nrm_s[x], hi != 0);
            const int gj = j0 + wsrc * 128 + x * 32 + (lane & 31);
            if (gj < p.N1) m = fmaxf(m, v);
            if (COLS && lane < 32 && gj < p.N1) p.snorm2[(long long)b * p.N1 + gj] = v;
        }
        m = row16_max(m);
        if ((lane & 15) == 0) atomicMax(p.smax2 + b, __float_as_uint(m));
    }
}

// Shared epilogue: row norms and, per wave, the top-2 of every target row over the wave's 128 source rows.  Each wave writes
// its own partial (no exchange between the two source halves of a tile: no LDS scratch, no barrier in the epilogue); the
// merge kernel reduces the 2 tilesS partials of a row.
//   acc[st][tt][r] = <src j, tgt i>,  j = j0 + wsrc*128 + st*32 + (r&3) + 8*(r>>2) + 4*(lane>>5),
//                                     i = i0 + wtgt*64 + tt*32 + (lane&31)
// Per lane and target row the 64 candidates are reduced as KEYS (k_key): 16 v_and_or + 23 max/med3 per block of
// 16, instead of compare/select chains on (value, index) pairs.  A key differs from its score by < 2^-19 relative;
// for any candidate j other than the winner  fp32(best) - fp32(j) >= (bv - sv) - 3 * 2^-19 |t||s|  (one for bv, one
// for the second key, one for its truncation), which is part of the bound that sends a row to the exact fix-up.
// DUAL (1 / 2): two key sets from the same accumulators, A = score + bias[j], B = score * scale[j] (1) or score (2);
// `bsl` = this tile's 256 bias values followed by its 256 scale values, in LDS.  The per-source terms are applied two
// accumulators at a time (v_pk_add_f32 / v_pk_mul_f32).
// TB = 32-row target blocks per wave (2: eight waves x 128 x 64; 4: four waves x 128 x 128, accumulators in AGPRs)
template <bool FULL, int TT, int SKIP = 0, int DUAL = 0, int TB = 2>     // SKIP (experiments): 4 no partial stores
__device__ __forceinline__ void simnn_tail(const simnn_params& p, f32x16 (&acc)[4][TB], float (&nrm_t)[TB], float (&nrm_s)[4],
                                           bool do_tn, bool do_sn, int b, int i0, int j0, int ts_,
                                           int lane_, int wsrc, int wtgt, const float* bsl = nullptr) {
    (void)lane_;
    const int lane = fresh_lane();
    const int hi = lane >> 5;
    (void)nrm_t; (void)nrm_s; (void)do_tn; (void)do_sn;      // (the row norms leave the kernel in simnn_norms, right behind the tile loop)

    constexpr int NKIND = (DUAL && DUAL != 4) ? 2 : 1;       // DUAL 4: key A alone (score + bias[j])
#pragma unroll
    for (int kind = 0; kind < NKIND; ++kind) {
#pragma unroll
    for (int tt = 0; tt < TB; ++tt) {
        float Bk = DM_KEY_NONE, Sk = DM_KEY_NONE;        // running best / second-best key of this lane
        int Bst = 0;                                     // block (of 16 candidates) the best key came from
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            float k[16];
            f32x4 w4[4];
            if (DUAL && (kind == 0 || DUAL == 1)) {      // the per-source terms of this block's 16 candidates
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    w4[q] = *reinterpret_cast<const f32x4*>(bsl + kind * 256 + wsrc * 128 + st * 32 + 4 * hi + 8 * q);
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 v = {acc[st][tt][r], acc[st][tt][r + 1]};
                const f32x2 w = {w4[r >> 2][r & 3], w4[r >> 2][(r & 3) + 1]};
                if (DUAL && kind == 0) v = v + w;
                if (DUAL == 1 && kind == 1) v = v * w;
                if (!FULL) {
                    const int j = j0 + wsrc * 128 + st * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    v[0] = (j < p.N1) ? v[0] : DM_KEY_NONE;
                    v[1] = (j + 1 < p.N1) ? v[1] : DM_KEY_NONE;
                }
                k[r] = k_key(v[0], 15 - r);              // position r ascends with j inside the block
                k[r + 1] = k_key(v[1], 14 - r);
            }
            float bk = k_max(k[0], k[1]), sk = k_min(k[0], k[1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) {
                const float m = k_med3(bk, k[r], k[r + 1]);
                bk = k_max3(bk, k[r], k[r + 1]);
                sk = k_max(sk, m);
            }
            Sk = k_max3(Sk, sk, k_min(Bk, bk));
            Bst = (bk > Bk) ? st : Bst;                  // (equal truncated scores: either block; such a row is re-scored exactly)
            Bk = k_max(Bk, bk);
        }
        // keys -> (value, source index, second value), merged over the two half-waves
        const int kb = __float_as_int(Bk);
        const int r_ = 15 - (kb & 15);
        float bv = __int_as_float(kb & ~15), sv = __int_as_float(__float_as_int(Sk) & ~15);
        int bj = j0 + wsrc * 128 + Bst * 32 + (r_ & 3) + 8 * (r_ >> 2) + 4 * hi;
        if (!FULL && !(bv > -1.0e38f)) { bv = DM_NEG_INF_F32; bj = DM_IDX_NONE; }
        if (!FULL && !(sv > -1.0e38f)) sv = DM_NEG_INF_F32;
        const float ob = xhalf(bv, hi != 0);
        const int oj = xhalf(bj, hi != 0);
        const float os = xhalf(sv, hi != 0);
        top2_merge(bv, bj, sv, ob, oj, os);
        if ((SKIP & 4) && bv == 1.2345f) p.pb[lane] = sv + bj;
        const int gi = i0 + wtgt * (32 * TB) + tt * 32 + lane;
        if (!(SKIP & 4) && lane < 32 && (FULL || gi < p.N2)) {
            const long long o = ((long long)b * (2 * p.tilesS) + 2 * ts_ + wsrc) * p.N2pad + gi;
            if (kind == 0) { p.pb[o] = bv; p.pj[o] = bj; p.ps[o] = sv; }
            else { p.pb_2[o] = bv; p.pj_2[o] = bj; p.ps_2[o] = sv; }
        }
    }
    }
}

// Column direction of a tile (DUAL 3, interior tiles): for every source row of the tile the top-2 over the tile's target
// rows, for two keys (score + biasT[i], score).  A lane owns target columns of the accumulator tiles, so each 32 x 32 tile
// goes through a wave-private LDS buffer (36-float row stride: 16-byte writes stay aligned, the transposed 4-byte reads are
// conflict-free) and comes back with the lane owning a SOURCE row and 16 targets n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
// in registers -- the layout of the row direction with the roles swapped, reduced with the same key arithmetic.  Partials
// are written per wave (target quarter of the tile): no cross-wave exchange.
//   tb: this wave's 32 x 36 float buffer; bT: the tile's 256 target biases in LDS
template <int NWT, int TB, bool FULL>  // NWT waves along the targets, each TB blocks of 32: partials per (tile row, wave);
                                       // FULL = false: the tile reaches into the padding (targets >= N2 / sources >= N1 are masked)
__device__ __forceinline__ void simnn_tail_cols(const simnn_params& p, f32x16 (&acc)[4][TB], float (&nrm_t)[TB], float (&nrm_s)[4],
                                                bool do_tn, bool do_sn, int b, int i0, int j0, int tt_, float* tb,
                                                const float* bT, int lane_, int wsrc, int wtgt) {
    const int lane = fresh_lane();
    const int hi = lane >> 5, l31 = lane & 31;
    (void)nrm_t; (void)nrm_s; (void)do_tn; (void)do_sn;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int gj = j0 + wsrc * 128 + st * 32 + l31;  // this lane's source row
        float Bk[2] = {DM_KEY_NONE, DM_KEY_NONE}, Sk[2] = {DM_KEY_NONE, DM_KEY_NONE};
        int Bt[2] = {0, 0};
#pragma unroll
        for (int tt = 0; tt < TB; ++tt) {
            // transpose: lane (n, hi) holds sources m = 8 q + 4 hi + e of target n  ->  tb[n][m]
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(tb + l31 * 36 + 8 * q + 4 * hi) =
                    f32x4{acc[st][tt][4 * q], acc[st][tt][4 * q + 1], acc[st][tt][4 * q + 2], acc[st][tt][4 * q + 3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float tr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[r] = tb[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + l31];
            f32x4 w4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w4[q] = *reinterpret_cast<const f32x4*>(bT + wtgt * (32 * TB) + tt * 32 + 4 * hi + 8 * q);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();              // (the buffer is rewritten by the next tile: reads first)
#pragma unroll
            for (int kind = 0; kind < 2; ++kind) {
                float k[16];
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 v = {tr[r], tr[r + 1]};
                    if (kind == 0) v = v + f32x2{w4[r >> 2][r & 3], w4[r >> 2][(r & 3) + 1]};
                    if (!FULL) {
                        const int i = i0 + wtgt * (32 * TB) + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        v[0] = (i < p.N2) ? v[0] : DM_KEY_NONE;
                        v[1] = (i + 1 < p.N2) ? v[1] : DM_KEY_NONE;
                    }
                    k[r] = k_key(v[0], 15 - r);
                    k[r + 1] = k_key(v[1], 14 - r);
                }
                float bk = k_max(k[0], k[1]), sk = k_min(k[0], k[1]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) {
                    const float m = k_med3(bk, k[r], k[r + 1]);
                    bk = k_max3(bk, k[r], k[r + 1]);
                    sk = k_max(sk, m);
                }
                Sk[kind] = k_max3(Sk[kind], sk, k_min(Bk[kind], bk));
                Bt[kind] = (bk > Bk[kind]) ? tt : Bt[kind];
                Bk[kind] = k_max(Bk[kind], bk);
            }
        }
#pragma unroll
        for (int kind = 0; kind < 2; ++kind) {
            const int kb = __float_as_int(Bk[kind]);
            const int r_ = 15 - (kb & 15);
            float bv = __int_as_float(kb & ~15), sv = __int_as_float(__float_as_int(Sk[kind]) & ~15);
            int bi = i0 + wtgt * (32 * TB) + Bt[kind] * 32 + (r_ & 3) + 8 * (r_ >> 2) + 4 * hi;
            if (!FULL && !(bv > -1.0e38f)) { bv = DM_NEG_INF_F32; bi = DM_IDX_NONE; }
            if (!FULL && !(sv > -1.0e38f)) sv = DM_NEG_INF_F32;
            const float ob = xhalf(bv, hi != 0);
            const int oi = xhalf(bi, hi != 0);
            const float os = xhalf(sv, hi != 0);
            top2_merge(bv, bi, sv, ob, oi, os);
            if (lane < 32 && (FULL || gj < p.N1)) {
                const long long o = ((long long)b * (p.tilesT * NWT) + tt_ * NWT + wtgt) * p.N1pad + gj;
                p.cb[kind][o] = bv; p.cj[kind][o] = bi; p.cs[kind][o] = sv;
            }
        }
    }
}

// Bounds-checked kernel for edge tiles and contraction depths that are not a multiple of 32: one workgroup per
// tile, operands staged through registers (two 64-deep LDS buffers), zero fill outside the matrices.
// Tile = 256 target rows x 256 source rows per 512-thread workgroup (8 waves = 2 source halves x 4 target quarters,
// each wave 128 source x 64 target = 4 x 2 MFMA tiles, 128 accumulator registers).
__global__ __launch_bounds__(512, 2) void simnn_edge_kernel(simnn_params p) {
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
    const bool do_tn = (ts_ == 0) && (wsrc == 0);
    const bool do_sn = (tt_ == 0) && (wtgt == 0);
    float nrm_t[2] = {0.f, 0.f}, nrm_s[4] = {0.f, 0.f, 0.f, 0.f};

    const int lrow = t >> 3, lchunk = t & 7;                                   // staging: rows q*64 + lrow, 16-byte chunk lchunk
    u32x4 rt[4], rs[4];
    // (macros, not lambdas: by-reference lambda captures of the staging arrays end up in scratch)
#define SIMNN_FETCH(s_)                                                                                          \
    {                                                                                                            \
        const int k_ = (s_) * SBK + lchunk * 8;                                                                  \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                          \
            const int row = q * 64 + lrow;                                                                       \
            const int gi = i0 + row, gj = j0 + row;                                                              \
            rt[q] = (gi < p.N2 && k_ < p.D) ? *reinterpret_cast<const u32x4*>(T + (long long)gi * p.ldT + k_)    \
                                            : u32x4{0, 0, 0, 0};                                                 \
            rs[q] = (gj < p.N1 && k_ < p.D) ? *reinterpret_cast<const u32x4*>(S + (long long)gj * p.ldS + k_)    \
                                            : u32x4{0, 0, 0, 0};                                                 \
        }                                                                                                        \
    }
#define SIMNN_STASH(buf_)                                                                                        \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
        const int row = q * 64 + lrow;                                                                           \
        const int off = (buf_) * ST * SBK + lds_off_halves(row, lchunk);                                         \
        *reinterpret_cast<u32x4*>(Ts + off) = rt[q];                                                             \
        *reinterpret_cast<u32x4*>(Ss + off) = rs[q];                                                             \
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
    simnn_norms<false, 2>(p, nrm_t, nrm_s, do_tn, do_sn, b, i0, j0, wsrc, wtgt);
    simnn_tail<false, ST>(p, acc, nrm_t, nrm_s, do_tn, do_sn, b, i0, j0, ts_, lane, wsrc, wtgt);
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Interior tiles, D % 32 == 0, D >= 96: the main kernel, in two shapes (template WT = target quarters per workgroup):
//   WT = 2: 4 waves, tile = 128 target x 256 source rows, ring of 3 stages (72 KiB) -> TWO workgroups per CU.  The two
//           waves of a SIMD then belong to different workgroups: one workgroup's barrier / fragment-read bubbles and
//           its whole reduction epilogue are covered by the other's MFMAs.  (the four-map pass while a pair's operands fit an XCD's L2)
//   WT = 4: 8 waves, tile = 256 x 256, ring of 4 stages (128 KiB), one workgroup per CU: a third less L2 -> LDS traffic,
//           but both waves of a SIMD meet every barrier together and nothing covers the epilogue.  (everything else)
//
// * Operands go L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no VGPRs, no ds_write) through a ring of
//   32-halves-deep stages; the DMA of stage g + NBUF - 1 is issued while stage g is computed and the wait before each
//   barrier is a COUNTED vmcnt (the younger stages stay in flight).
// * The ring is indexed by a stage counter that runs across tiles: a workgroup walks a LIST of tiles (launched one or
//   two per CU: "persistent"; launched one per tile the list has one entry) and the first stages of the next tile are
//   in flight during the last stages and the reduction epilogue of the current one.  The tiles a workgroup walks are
//   those of its XCD (block b runs on XCD b % 8; each XCD gets a contiguous range of tile ids = whole mesh pairs,
//   whose operand panels then meet in one 4 MiB L2), consecutive ids per round.
// * K-staggered sweep: the workgroups that share a target panel (same tt) or a source panel (same ts) run in
//   lockstep, so with a common sweep order every one of them sees every first touch of a line as an L2 miss.
//   Tile (tt, ts) starts its (cyclic) sweep at a stage that depends on (tt, ts) instead: after its first stages each
//   workgroup trails a neighbour that has already pulled the lines into the XCD's L2.  The sum is the same up to fp32
//   rounding, which the exact fix-up absorbs.
//
// LDS image of one operand stage: rows x 64 B; chunk c (16 B) of row r lives in slot c ^ ((r >> 2) & 3), so the
// 16 rows of a ds_read_b128 lane group (four runs of 4 consecutive rows with distinct (r >> 2) & 3) cover all 16
// slots of the 256-byte bank row.  One DMA instruction fills 16 rows (lane l -> row l >> 2, slot l & 3), the swizzle
// is applied on the source address.
//
// XV = compile-time variant bits; the product instantiates SIMNN_PRODUCT_XV only, a DM_EXPERIMENTS build a list of them
// (the ablations give WRONG results): low 4 bits: 1 skip the epilogue, 3 no norms, 5 no stage barriers, 7 LDS-DMA only, 8 no LDS-DMA, 9 = 8 + 1;
// 16 / 32: K stagger by one stage / spread over the whole sweep; 64: fragment reads pinned in front of the MFMAs;
// 128: the second wave of every SIMD runs its MFMAs first and its reads / DMA last inside each barrier interval.
// Measured on config 3 (tools/simnn_experiment.py, profiles/r02_simnn_variants.txt): stagger helps the DMA-only
// ablation (366 -> 268 us) but not the whole kernel; flipping the second wave changes nothing; the 4-wave shape hides
// the epilogue (21 instead of 57 us) and loses more to its 1.5x DMA traffic.
constexpr int PBK = 32;                    // halves per stage
constexpr int SIMNN_PRODUCT_XV = 64;
// the one-key kernel (config 3) issues its fragment reads and DMA between the matrix instructions of a k-step (bit 4096): -1.3 %
// on features, -3 % on zero operands (tools/simnn_power_test.py); the key-set kernels keep them in front (+2 % with it, simnn1)
constexpr int SIMNN_PRODUCT_ILV0 = 4096;
constexpr int SIMNN_PRODUCT_WT = 4;
#define DM_WAIT_VM_LGKM0(n) __builtin_amdgcn_s_waitcnt(0x0070 | ((n) & 15) | ((((n) >> 4) & 3) << 14))    /* vmcnt(n) lgkmcnt(0) */
static inline size_t simnn_pipe_lds(int WT, int dual = 0) {
    const int TT = 64 * WT, NBUF = dual == -2 ? 5 : (WT == 4 ? 4 : 3);     // (-2: the five-slot ring of the one-key kernel)
    // ring | DUAL: two slots of (256 bias + 256 scale [+ TT target bias]) floats | DUAL 3, 8 waves: the eighth wave's transpose
    // buffer (the other seven use the ring slot that is free during an epilogue)
    size_t n = (size_t)NBUF * (TT + ST) * PBK * sizeof(_Float16);
    if (dual == -1) n += 8 * 256 * 8;                  // experiments: the stage timeline's log
    if (dual > 0) n += (size_t)2 * (dual == 3 ? 512 + TT : 512) * 4;
    if (dual == 3 && WT == 4) n += 32 * 36 * 4;
    return n;
}

// Tile order inside a pair: bands of p.band tile rows, column-major inside a band.  The ~32 workgroups of an XCD work on
// consecutive ids, i.e. on a (band x 32 / band) block of tiles that shares band + 32 / band operand panels (12 for band = 4,
// 3.7 MB at a contraction depth of 608: inside the XCD's 4 MiB L2) and walks along the band, so a band's target panels stay
// resident while each source panel is fetched once per band.  Row-major order (one tile row x 32 columns = 33 panels, 10 MB)
// re-fetched every source panel for every tile row: 22.8 GB per launch at N = 8192 (profiles/r02_stress_hbm_traffic_pmc.csv).
// The pass is NOT bound by that traffic, though: the banded order is 5 % faster at N = 8192 (dm_set_option "simnn_band");
// the four reductions of the epilogue are half of the pass (tools/simnn4_experiment.py).
__device__ __forceinline__ void simnn_decode(const simnn_params& p, int id, int& b, int& tt_, int& ts_) {
    const int tiles = p.rnT * p.rnS;
    b = id / tiles;
    const int tts = id - b * tiles;
    const int per_band = p.band * p.rnS;
    const int band = tts / per_band;
    const int brow0 = band * p.band, brows = min(p.band, p.rnT - brow0);
    const int brem = tts - band * per_band;
    const int cs = brem / brows;
    ts_ = p.rs0 + cs;
    tt_ = p.rt0 + brow0 + (brem - cs * brows);
}

// TB = 32-row target blocks per wave.  TB = 2: waves of 128 source x 64 target rows (128 accumulator registers, two waves per
// SIMD).  TB = 4 (WT = 4 only): FOUR waves of 128 x 128, 256 accumulators per lane in the AGPR half of the register file, one
// wave per SIMD: 8 fragment reads per 16 matrix instructions instead of 6 per 8 -- a third less traffic on the LDS pipe,
// which is what the main loop is bound by.
// EDGE: the launch walks tiles that reach into the padding of operands padded to whole tiles (key-set passes on sizes that
// are not multiples of 256): its reductions mask targets >= N2 and sources >= N1.  Whole tiles run the EDGE = false kernel,
// the same code as for aligned sizes.
template <int XV, int WT, int DUAL = 0, int TB = 2, bool EDGE = false>
__global__ __launch_bounds__(2 * 64 * (2 * WT / TB), TB == 4 ? 1 : 2) void simnn_pipe_kernel(simnn_params p) {
    static_assert(TB == 2 || (TB == 4 && WT == 4), "128 x 128 waves only for the 256-row tile");
    constexpr int TT = 64 * WT;                  // target rows per tile
    constexpr int NWT = TT / (32 * TB);          // waves along the target rows
    constexpr int NW = 2 * NWT;                  // waves
    constexpr bool EARLY = (XV & 2048) != 0;     // five-slot ring, barrier at the END of a stage (see SIMNN_STAGE)
    static_assert(!EARLY || (DUAL == 0 && WT == 4 && TB == 2), "the five-slot ring fills the LDS: one-key kernel, 8 waves");
    constexpr int NBUF = EARLY ? 5 : (WT == 4 ? 4 : 3);   // ring depth
    constexpr int PD = NBUF - 1;                 // stages the DMA runs ahead
    constexpr int PSTAGE = (TT + ST) * PBK;      // halves per ring slot: T image then S image
    constexpr int NSI = 16 / NW;                 // DMA instructions per wave and stage for S ...
    constexpr int NTI = TT / 16 / NW;            // ... and for T
    constexpr int VM_STEADY = EARLY ? 2 * (NTI + NSI) : (PD - 2) * (NTI + NSI) + (NTI + NSI) / 2;   // loads that may stay in flight at the barrier
    constexpr int dbg = XV & 15;
    constexpr int STAG = (XV >> 4) & 3;
    constexpr bool PINR = (XV & 64) != 0;
    constexpr bool FLIP = (XV & 128) != 0;       // second wave of each SIMD: MFMAs first, then the reads / DMA of the half-stage
    constexpr bool TRACE = (XV & 1024) != 0;     // experiments: s_memtime stamps of every stage of workgroup 0's third tile
    constexpr bool ILV = (XV & 4096) != 0;       // fragment reads and DMA issued BETWEEN the matrix instructions of a k-step
    constexpr bool PRIO = (XV & 8192) != 0;      // s_setprio(1) around every cluster of matrix instructions: the SIMD's arbiter prefers the wave
                                                 // that is feeding the matrix pipe over the other wave's reads / DMA / reduction epilogue
    constexpr bool SPLIT = DUAL != 0;            // the key-set kernels read split rows [16 high | 16 low] per stage (dm_knnsplit.hip)
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];             // NBUF x (T | S) | per-tile terms | transpose buffer
    // Behind the ring: DUAL: two slots of per-tile terms | DUAL 3, 8 waves: the eighth wave's transpose buffer (the transposes of
    // the other waves go through the ring slot that is free during an epilogue; the 4-wave both-directions kernel, two
    // workgroups per CU with 80 KiB each, has room for all of its four there).
    constexpr int BSLOT = DUAL == 3 ? 512 + TT : 512;  // floats per slot: 256 bias | 256 scale | TT target bias
    float* bias_lds = reinterpret_cast<float*>(smem + NBUF * PSTAGE);   // filled by LDS-DMA one tile ahead
    float* tb_extra = bias_lds + 2 * BSLOT;      // DUAL 3, 8 waves: transpose buffer of wave 7

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wsrc = wave & 1, wtgt = wave >> 1;
    static_assert(NTI % 2 == 0 && NSI % 2 == 0, "the DMA of a stage is issued in two halves");

    // tiles of this workgroup: ids base + slot, base + slot + nslot, ... of the XCD's range [base, base + cnt)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nslot = ((int)gridDim.x - xcd + 7) >> 3;
    const int q8 = p.total >> 3, r8 = p.total & 7;
    const int base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    if (slot >= cnt) return;
    const int ntile = (cnt - slot + nslot - 1) / nslot;
    const int ns = p.D / PBK;                                    // >= NBUF + 1 (host)

    // LDS-DMA source: per-lane part (row l >> 2 of a 16-row group, swizzled chunk) + uniform part (tile, wave, stage)
    const unsigned voffT = (unsigned)(((lane >> 2) * p.ldT + (((lane & 3) ^ ((lane >> 4) & 3)) << 3)) * 2);
    const unsigned voffS = (unsigned)(((lane >> 2) * p.ldS + (((lane & 3) ^ ((lane >> 4) & 3)) << 3)) * 2);
    // experiment STAG == 3 (wrong data placement, same bytes): every piece reads 8 rows x 128 bytes (whole cache lines) instead of
    // 16 rows x 64; two consecutive stages cover the 16 rows of a piece
    const unsigned voffT_x = (unsigned)(((lane >> 3) * p.ldT + ((lane & 7) << 3)) * 2);
    const unsigned voffS_x = (unsigned)(((lane >> 3) * p.ldS + ((lane & 7) << 3)) * 2);
    // the stage stream being fetched (runs PD stages ahead of the one being computed)
    int d_tile = 0, d_s = 0, d_kp = 0, d_slot = 0;
    const char* d_T = nullptr;                   // rows wave*32 .. of the tile's target panel (2 x 16 rows per stage)
    const char* d_S = nullptr;                   // rows wave*16*NSI .. of the tile's source panel (NSI x 16 rows)
#define SIMNN_DMA_TILE()                                                                                               \
    {                                                                                                                  \
        int b_, tt_, ts_;                                                                                              \
        simnn_decode(p, base + slot + d_tile * nslot, b_, tt_, ts_);                                                   \
        d_T = reinterpret_cast<const char*>(p.Ftgt + ((long long)b_ * p.rowsT + tt_ * TT + wave * 16 * NTI) * p.ldT);  \
        d_S = reinterpret_cast<const char*>(p.Fsrc + ((long long)b_ * p.rowsS + ts_ * ST + wave * 16 * NSI) * p.ldS);  \
        d_kp = (STAG == 0 || STAG == 3) ? 0 : (STAG == 1 ? (ts_ + tt_) % ns : ((ts_ + tt_) * ns / p.tilesS) % ns);                    \
        d_s = 0;                                                                                                       \
        if (DUAL && wave == 0) {      /* the tile's per-source terms: one 1 KiB piece each, landed long before its epilogue */ \
            float* dstB = bias_lds + (d_tile & 1) * BSLOT;                                                             \
            const int lane16 = fresh_lane() * 16;                                                                      \
            const char* gb = reinterpret_cast<const char*>(p.bias + (long long)b_ * p.rowsS + ts_ * ST) + lane16;      \
            __builtin_amdgcn_global_load_lds((gptr_t)gb, (lptr_t)dstB, 16, 0, 0);                                      \
            if (DUAL == 1 || DUAL == 3) {                                                                              \
                const char* gs = reinterpret_cast<const char*>(p.scale + (long long)b_ * p.rowsS + ts_ * ST) + lane16; \
                __builtin_amdgcn_global_load_lds((gptr_t)gs, (lptr_t)(dstB + 256), 16, 0, 0);                          \
            }                                                                                                          \
            if (DUAL == 3 && lane16 < TT * 4) {                                                                        \
                const char* gt2 = reinterpret_cast<const char*>(p.biasT + (long long)b_ * p.rowsT + tt_ * TT) + lane16; \
                __builtin_amdgcn_global_load_lds((gptr_t)gt2, (lptr_t)(dstB + 512), 16, 0, 0);                         \
            }                                                                                                          \
        }                                                                                                              \
    }
    // half H_ (0 / 1) of the DMA instructions of the stage stream's current stage: T piece H_, S pieces H_*NSI/2 ..
#define SIMNN_DMA1(H_)                                                                                                 \
    if (!(dbg & 8) || d_tile == 0) {                                                                                   \
        _Pragma("unroll") for (int u = 0; u < NTI / 2; ++u) {                                                          \
            const int piece = (H_) * (NTI / 2) + u;                                                                    \
            _Float16* dstT = smem + d_slot * PSTAGE + (wave * 16 * NTI + piece * 16) * PBK;                            \
            const char* gt = STAG == 3 ? d_T + ((long long)(piece * 16 + (d_kp & 1) * 8) * p.ldT + (d_kp >> 1) * 64) * 2    \
                                       : d_T + ((long long)(piece * 16) * p.ldT + d_kp * PBK) * 2;                     \
            __builtin_amdgcn_global_load_lds((gptr_t)(gt + (STAG == 3 ? voffT_x : voffT)), (lptr_t)dstT, 16, 0, 0);    \
        }                                                                                                              \
        _Pragma("unroll") for (int u = 0; u < NSI / 2; ++u) {                                                          \
            const int piece = (H_) * (NSI / 2) + u;                                                                    \
            _Float16* dstS = smem + d_slot * PSTAGE + (TT + wave * 16 * NSI + piece * 16) * PBK;                       \
            const char* gs = STAG == 3 ? d_S + ((long long)(piece * 16 + (d_kp & 1) * 8) * p.ldS + (d_kp >> 1) * 64) * 2    \
                                       : d_S + ((long long)(piece * 16) * p.ldS + d_kp * PBK) * 2;                     \
            __builtin_amdgcn_global_load_lds((gptr_t)(gs + (STAG == 3 ? voffS_x : voffS)), (lptr_t)dstS, 16, 0, 0);    \
        }                                                                                                              \
    }
#define SIMNN_DMA_NEXT()                                                                                               \
    {                                                                                                                  \
        ++d_s;                                                                                                         \
        d_kp = (d_kp + 1 == ns) ? 0 : d_kp + 1;                                                                        \
        d_slot = (d_slot + 1 == NBUF) ? 0 : d_slot + 1;                                                                \
        if (d_s == ns) { ++d_tile; if (d_tile < ntile) SIMNN_DMA_TILE() }                                              \
    }

    // fragment addresses: row (lane & 31) of a 32-row block, chunk kk*2 + (lane >> 5); the swizzle term depends on
    // the lane only, and kk = 1 flips bit 1 of the slot
    const int swz = (lane >> 2) & 3;
    const int c0 = (lane >> 5) ^ swz;
    const int frow = (lane & 31) * PBK;
    const int foff0 = frow + (c0 << 3), foff1 = frow + ((c0 ^ 2) << 3);
    const int sbase = TT * PBK + wsrc * 128 * PBK, tbase = wtgt * (32 * TB) * PBK;

    SIMNN_DMA_TILE()
#pragma unroll
    for (int q = 0; q < PD; ++q) { SIMNN_DMA1(0) SIMNN_DMA1(1) SIMNN_DMA_NEXT() }
    {                                                                // vmcnt: stage 0 has landed; the others in flight
        constexpr int n0 = (EARLY ? PD - 2 : PD - 1) * (NTI + NSI);      // (EARLY: stages 0 and 1 have landed)
        __builtin_amdgcn_s_waitcnt(0x0F70 | (n0 & 15) | (((n0 >> 4) & 3) << 14));
    }
    __builtin_amdgcn_s_barrier();

    // The stage loop is rotated by half a stage: the barrier that publishes stage g+1 sits between the two k-steps of
    // stage g, so the first fragments of stage g+1 are fetched under the MFMAs of (g, kk=1) and no fragment read is
    // exposed after a barrier.  Ring slot of stage g+PD == slot of stage g-1, last read by the (g-1, kk=1) fragments,
    // complete (lgkmcnt(0)) before the barrier of iteration g-1, which every wave has left before iteration g starts.
    // vmcnt before the barrier of iteration g: stage g+1 must have landed; younger are the stages g+2 .. g+PD-1 and the
    // first half of stage g+PD, fewer at the very end of the walk.  Stores of an epilogue in between only make the
    // count conservative (it bounds loads + stores in flight).
    f16x8 fsa[4], fta[TB], fsb[4], ftb[TB], ftn[TB];
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    constexpr bool NOEPI = (dbg & 7) == 1 || (dbg & 7) == 7;
    // stores EVERY wave issues in an epilogue: its row partials (2 target blocks x 3 arrays per key set) and, both directions,
    // its column partials (4 source blocks x 2 key sets x 3 arrays)
    constexpr int EPI_ST = ((DUAL && DUAL != 4) ? 2 : 1) * 3 * TB + (DUAL == 3 ? 24 : 0);
    int r_slot = 0;                               // ring slot of the stage being computed
    const bool late = FLIP && wave >= NW / 2;
#define SIMNN_READ(fs_, ft_, slot_, fo_)                                                                               \
    if ((dbg & 7) != 7) {                                                                                              \
        const _Float16* Bs = smem + (slot_) * PSTAGE;                                                                  \
        _Pragma("unroll") for (int x = 0; x < 4; ++x)                                                                  \
            fs_[x] = *reinterpret_cast<const f16x8*>(Bs + sbase + x * 32 * PBK + (fo_));                               \
        _Pragma("unroll") for (int x = 0; x < TB; ++x)                                                                 \
            ft_[x] = *reinterpret_cast<const f16x8*>(Bs + tbase + x * 32 * PBK + (fo_));                               \
    }
#define SIMNN_MMA(fs_, ft_, NORMS, ZERO_)                                                                              \
    if ((dbg & 7) != 7) {                                                                                              \
        if (NORMS) {                                                                                                   \
            if (do_tn) { _Pragma("unroll") for (int x = 0; x < TB; ++x) nrm_t[x] = sumsq8(ft_[x], nrm_t[x]); }         \
            if (do_sn) { _Pragma("unroll") for (int x = 0; x < 4; ++x) nrm_s[x] = sumsq8(fs_[x], nrm_s[x]); }          \
        }                                                                                                              \
        if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int st = 0; st < 4; ++st)                                                               \
            _Pragma("unroll") for (int tt = 0; tt < TB; ++tt)                                                          \
                acc[st][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs_[st], ft_[tt], (ZERO_) ? zero16 : acc[st][tt], 0, 0, 0); \
        if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                       \
    }
    // a k-step whose fragment reads (next k-step's operands: rs_ / rt_ from ring slot rslot_ at rfo_) and DMA half H_ are issued in
    // the shadow of its own matrix instructions -- two reads behind each of the first MFMAs, then the DMA -- instead of in
    // front of them: the matrix pipe starts right behind the barrier
#define SIMNN_MMA_ILV(fs_, ft_, ZERO_, RDS_, rs_, RDT_, rt_, rslot_, rfo_, DODMA_, H_)                                 \
    if ((dbg & 7) != 7) {                                                                                              \
        const _Float16* Br = smem + (rslot_) * PSTAGE;                                                                 \
        if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int st = 0; st < 4; ++st)                                                               \
            _Pragma("unroll") for (int tt = 0; tt < TB; ++tt) {                                                        \
                acc[st][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs_[st], ft_[tt], (ZERO_) ? zero16 : acc[st][tt], 0, 0, 0); \
                const int mi = st * TB + tt;                                                                           \
                const int ns_ = (RDS_) ? 4 : 0, nt_ = (RDT_) ? TB : 0;                                                 \
                _Pragma("unroll") for (int r = 2 * mi; r < 2 * mi + 2; ++r) {                                          \
                    if (r < ns_) rs_[r < 4 ? r : 0] = *reinterpret_cast<const f16x8*>(Br + sbase + r * 32 * PBK + (rfo_));  \
                    else if (r < ns_ + nt_) rt_[(r - ns_) < TB ? (r - ns_) : 0] = *reinterpret_cast<const f16x8*>(Br + tbase + (r - ns_) * 32 * PBK + (rfo_)); \
                }                                                                                                      \
                if ((DODMA_) && mi == (ns_ + nt_ + 1) / 2) { SIMNN_DMA1(H_) }                                           \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
        if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                       \
    }
#define SIMNN_SYNC(n_, AFTER_EPI_)                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    if ((AFTER_EPI_) && n > 0 && !NOEPI) DM_WAIT_VM_LGKM0((n_) + EPI_ST);                                              \
    else DM_WAIT_VM_LGKM0(n_);                                                                                         \
    if ((dbg & 15) != 5) __builtin_amdgcn_s_barrier();      /* (ablation 5: no stage barriers -- wrong results) */    \
    __builtin_amdgcn_sched_barrier(0);
#define SIMNN_PIN() if (PINR) __builtin_amdgcn_sched_barrier(0);
    // TRACE: seven s_memtime stamps per stage, taken without waiting for them (the scalar memory unit executes them in program
    // order; the stage's own lgkmcnt(0) waits cover them) and written by lane 0 to this wave's part of an LDS log at the end of
    // the stage; the log of workgroup 0's third tile is copied out when the tile is done
    unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0, tq5 = 0, tq6 = 0;
    int tr_n = 0;
    bool tr_on = false;
    unsigned long long* tr_lds = reinterpret_cast<unsigned long long*>(smem + NBUF * PSTAGE) + wave * 256;
#define SIMNN_STAMP(v_) if constexpr (TRACE) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(v_)); __builtin_amdgcn_sched_barrier(0); }
#define SIMNN_TRACE_FLUSH()                                                                                            \
    if constexpr (TRACE) {                                                                                             \
        if (tr_on && tr_n + 8 <= 256 && lane == 0) {                                                                   \
            tr_lds[tr_n] = tq0; tr_lds[tr_n + 1] = tq1; tr_lds[tr_n + 2] = tq2; tr_lds[tr_n + 3] = tq3;                \
            tr_lds[tr_n + 4] = tq4; tr_lds[tr_n + 5] = tq5; tr_lds[tr_n + 6] = tq6; tr_lds[tr_n + 7] = 0;              \
        }                                                                                                              \
        if (tr_on) tr_n += 8;                                                                                          \
    }
    // one stage: DMA_ = 1 in the steady state (stage g+PD exists), VM_ = loads allowed to stay in flight at the barrier,
    // NEXT_ = fetch the first fragments of the next stage (in the steady state also across a tile boundary: they wait
    // in registers while the epilogue runs)
#define SIMNN_STAGE(NORMS, DMA_, VM_, NEXT_, ZERO_, AFTER_EPI_)                                                        \
    {                                                                                                                  \
        const int n_slot = (r_slot + 1 == NBUF) ? 0 : r_slot + 1;                                                      \
        if constexpr (EARLY) {                                                                                         \
        /* Five slots, the DMA four stages ahead, ONE barrier at the END of the stage: it publishes stage g + 2, so the     \
           first fragments of stage g + 1 are requested BEFORE it, under this stage's second k-step, and the first matrix   \
           instructions behind the barrier have their operands in registers.  (Four slots: the barrier sat between the two  \
           k-steps, every wave requested its next fragments in one burst right behind it and waited 350-470 cycles for      \
           them at the end of the stage: profiles/r04_simnn_stage_timeline.txt.)  Slot of stage g + 4 = slot of g - 1: its   \
           last readers finished (lgkmcnt(0)) before the barrier of stage g - 1. */                                        \
        SIMNN_READ(fsb, ftb, r_slot, foff1)                                                                            \
        if (DMA_) { SIMNN_DMA1(0) }                                                                                    \
        SIMNN_PIN()                                                                                                    \
        SIMNN_MMA(fsa, fta, NORMS, ZERO_)                                                                              \
        if (NEXT_) { SIMNN_READ(fsa, fta, n_slot, foff0) }                                                             \
        if (DMA_) { SIMNN_DMA1(1) }                                                                                    \
        SIMNN_PIN()                                                                                                    \
        SIMNN_MMA(fsb, ftb, NORMS, false)                                                                              \
        SIMNN_SYNC(VM_, AFTER_EPI_)                                                                                    \
        } else if constexpr (ILV && !SPLIT) {                                                                          \
        if (NORMS) {                                                                                                   \
            if (do_tn) { _Pragma("unroll") for (int x = 0; x < TB; ++x) nrm_t[x] = sumsq8(fta[x], nrm_t[x]); }         \
            if (do_sn) { _Pragma("unroll") for (int x = 0; x < 4; ++x) nrm_s[x] = sumsq8(fsa[x], nrm_s[x]); }          \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        SIMNN_MMA_ILV(fsa, fta, ZERO_, true, fsb, true, ftb, r_slot, foff1, DMA_, 0)                                   \
        SIMNN_SYNC(VM_, AFTER_EPI_)                                                                                    \
        if (NORMS) {                                                                                                   \
            if (do_tn) { _Pragma("unroll") for (int x = 0; x < TB; ++x) nrm_t[x] = sumsq8(ftb[x], nrm_t[x]); }         \
            if (do_sn) { _Pragma("unroll") for (int x = 0; x < 4; ++x) nrm_s[x] = sumsq8(fsb[x], nrm_s[x]); }          \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        SIMNN_MMA_ILV(fsb, ftb, false, NEXT_, fsa, NEXT_, fta, n_slot, foff0, DMA_, 1)                                 \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
        } else if constexpr (ILV) {                                                                                    \
        if (NORMS) {                                                                                                   \
            if (do_tn) { _Pragma("unroll") for (int x = 0; x < TB; ++x) nrm_t[x] = sumsq8(fta[x], sumsq8(fta[x], nrm_t[x])); } \
            if (do_sn) { _Pragma("unroll") for (int x = 0; x < 4; ++x) nrm_s[x] = sumsq8(fsa[x], sumsq8(fsa[x], nrm_s[x])); } \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        SIMNN_MMA_ILV(fsa, fta, ZERO_, true, fsb, true, ftb, r_slot, foff1, DMA_, 0)                                   \
        SIMNN_SYNC(VM_, AFTER_EPI_)                                                                                    \
        if (NORMS) {                                                                                                   \
            if (do_tn) { _Pragma("unroll") for (int x = 0; x < TB; ++x) nrm_t[x] = sumsq8(ftb[x], nrm_t[x]); }         \
            if (do_sn) { _Pragma("unroll") for (int x = 0; x < 4; ++x) nrm_s[x] = sumsq8(fsb[x], nrm_s[x]); }          \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        SIMNN_MMA_ILV(fsa, ftb, false, false, fsa, NEXT_, ftn, n_slot, foff0, DMA_, 1)                                 \
        SIMNN_MMA_ILV(fsb, fta, false, NEXT_, fsa, false, ftn, n_slot, foff0, false, 1)                                \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
        if (NEXT_) { _Pragma("unroll") for (int x = 0; x < TB; ++x) fta[x] = ftn[x]; }                                 \
        } else if constexpr (!SPLIT) {                                                                                 \
        SIMNN_STAMP(tq0)                                                                                               \
        if (!late) { SIMNN_READ(fsb, ftb, r_slot, foff1) if (DMA_) { SIMNN_DMA1(0) } SIMNN_PIN() }                       \
        SIMNN_STAMP(tq1)                                                                                               \
        SIMNN_MMA(fsa, fta, NORMS, ZERO_)                                                                              \
        if (late) { SIMNN_PIN() SIMNN_READ(fsb, ftb, r_slot, foff1) if (DMA_) { SIMNN_DMA1(0) } }                        \
        SIMNN_STAMP(tq2)                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if ((AFTER_EPI_) && n > 0 && !NOEPI) DM_WAIT_VM_LGKM0((VM_) + EPI_ST); else DM_WAIT_VM_LGKM0(VM_);             \
        SIMNN_STAMP(tq3)                                                                                               \
        if ((dbg & 15) != 5) __builtin_amdgcn_s_barrier();                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        SIMNN_STAMP(tq4)                                                                                               \
        if (!late) { if (NEXT_) { SIMNN_READ(fsa, fta, n_slot, foff0) } if (DMA_) { SIMNN_DMA1(1) } SIMNN_PIN() }        \
        SIMNN_STAMP(tq5)                                                                                               \
        SIMNN_MMA(fsb, ftb, NORMS, false)                                                                              \
        if (late) { SIMNN_PIN() if (NEXT_) { SIMNN_READ(fsa, fta, n_slot, foff0) } if (DMA_) { SIMNN_DMA1(1) } }         \
        SIMNN_STAMP(tq6)                                                                                               \
        /* the fragments of the next half-stage were requested eight MFMAs ago: make their arrival explicit here, or    \
           the compiler, merging the loop back-edge, waits lgkmcnt(0) in FRONT of the next MFMAs -- i.e. for the reads   \
           that were only just issued there */                                                                         \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
        SIMNN_TRACE_FLUSH()                                                                                            \
        } else {                                                                                                       \
        /* split rows: a stage holds 16 contraction indices as [16 high halves | 16 low halves] of both operands and     \
           feeds THREE k-steps, hs.ht + hs.lt + ls.ht, from the same 12 fragment reads and the same 32 KiB of LDS-DMA    \
           that fed two before (the (h,h,l) x (h,l,h) rows stored every high half twice): the LDS pipe -- fragment     \
           reads + DMA writes -- is what the main loop is bound by.  fsa / fta = high parts, fsb / ftb = low parts.     \
           The next stage's target high parts land in ftn (fta is an operand of the stage's last k-step).  */          \
        SIMNN_READ(fsb, ftb, r_slot, foff1)                                                                            \
        if (DMA_) { SIMNN_DMA1(0) }                                                                                    \
        SIMNN_PIN()                                                                                                    \
        if (NORMS) {   /* |row|^2 of the rows the products stand for: (h, h, l) resp. (h, l, h) */                     \
            if (do_tn) { _Pragma("unroll") for (int x = 0; x < TB; ++x) nrm_t[x] = sumsq8(fta[x], sumsq8(fta[x], nrm_t[x])); } \
            if (do_sn) { _Pragma("unroll") for (int x = 0; x < 4; ++x) nrm_s[x] = sumsq8(fsa[x], sumsq8(fsa[x], nrm_s[x])); } \
        }                                                                                                              \
        SIMNN_MMA(fsa, fta, false, ZERO_)                                                                              \
        SIMNN_SYNC(VM_, AFTER_EPI_)                                                                                    \
        if (NORMS) {                                                                                                   \
            if (do_tn) { _Pragma("unroll") for (int x = 0; x < TB; ++x) nrm_t[x] = sumsq8(ftb[x], nrm_t[x]); }         \
            if (do_sn) { _Pragma("unroll") for (int x = 0; x < 4; ++x) nrm_s[x] = sumsq8(fsb[x], nrm_s[x]); }          \
        }                                                                                                              \
        SIMNN_MMA(fsa, ftb, false, false)                                                                              \
        if (NEXT_ && (dbg & 7) != 7) {                                                                                 \
            const _Float16* Bn = smem + n_slot * PSTAGE;                                                               \
            _Pragma("unroll") for (int x = 0; x < 4; ++x)                                                              \
                fsa[x] = *reinterpret_cast<const f16x8*>(Bn + sbase + x * 32 * PBK + foff0);                           \
            _Pragma("unroll") for (int x = 0; x < TB; ++x)                                                             \
                ftn[x] = *reinterpret_cast<const f16x8*>(Bn + tbase + x * 32 * PBK + foff0);                           \
        }                                                                                                              \
        if (DMA_) { SIMNN_DMA1(1) }                                                                                    \
        SIMNN_PIN()                                                                                                    \
        SIMNN_MMA(fsb, fta, false, false)                                                                              \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
        if (NEXT_) { _Pragma("unroll") for (int x = 0; x < TB; ++x) fta[x] = ftn[x]; }                                 \
        }                                                                                                              \
        r_slot = n_slot;                                                                                               \
        if (DMA_) SIMNN_DMA_NEXT()                                                                                     \
    }
    // Every stage of every tile but the last PD of the walk is a steady-state stage: no branch inside the body.  The
    // first k-step of a tile accumulates onto zero.  The first PD-1 stages of a tile that follows an epilogue let that
    // epilogue's stores stay in flight too (every wave issued at least EPI_ST of them; vmcnt counts loads + stores in
    // issue order, so allowing EPI_ST more keeps the wait to the LDS-DMA it is meant for instead of a store's round trip).
#define SIMNN_TILE_LOOP(NORMS)                                                                                         \
    {                                                                                                                  \
        const int nsteady = (n + 1 < ntile) ? ns : ns - PD;          /* >= 2 (host: ns >= NBUF + 1) */                 \
        SIMNN_STAGE(NORMS, 1, VM_STEADY, 1, true, true)                                                                \
        if (PD >= 3) SIMNN_STAGE(NORMS, 1, VM_STEADY, 1, false, true)                                                  \
        for (int s = (PD >= 3 ? 2 : 1); s < nsteady; ++s) SIMNN_STAGE(NORMS, 1, VM_STEADY, 1, false, false)            \
        if (n + 1 == ntile) {                                                                                          \
            if (EARLY) SIMNN_STAGE(NORMS, 0, NTI + NSI, 1, false, false)    /* (the barrier of stage g needs g + 2) */  \
            if (PD >= 3) SIMNN_STAGE(NORMS, 0, EARLY ? 0 : NTI + NSI, 1, false, false)                                 \
            SIMNN_STAGE(NORMS, 0, 0, 1, false, false)                                                                  \
            SIMNN_STAGE(NORMS, 0, 0, 0, false, false)                                                                  \
        }                                                                                                              \
    }

    constexpr bool DEFER_READ = DUAL != 0;
    SIMNN_READ(fsa, fta, 0, foff0)
    for (int n = 0; n < ntile; ++n) {
        int b, tt_, ts_;
        simnn_decode(p, base + slot + n * nslot, b, tt_, ts_);
        // The key-set kernels do not carry the next tile's first fragments across their epilogue (24 registers that pushed the
        // epilogue into scratch, and every scratch reload waits vmcnt(0), i.e. for the whole LDS-DMA queue): the copies fetched by
        // the last stage die there and are read again here.
        if (DEFER_READ && n > 0) { SIMNN_READ(fsa, fta, r_slot, foff0) }
        const int i0 = tt_ * TT, j0 = ts_ * ST;
        if constexpr (TRACE) tr_on = (blockIdx.x == 0 && n == 2);
        f32x16 acc[4][TB];
        // squared row norms for the exactness bound, accumulated from the MFMA fragments by the workgroups that own
        // the first tile of the other operand (every row of T / S is seen exactly once that way).  Those tiles run a
        // second copy of the loop, so the hot copy has no conditional inside a k-step.
        constexpr bool want_n = (dbg & 7) != 3 && (dbg & 7) != 7;
        const bool do_tn = (ts_ == 0) && (wsrc == 0) && want_n;
        const bool do_sn = (tt_ == 0) && (wtgt == 0) && want_n;
        float nrm_t[TB], nrm_s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int x = 0; x < TB; ++x) nrm_t[x] = 0.f;

        if ((ts_ == 0 || tt_ == 0) && want_n) {
            SIMNN_TILE_LOOP(true)
            simnn_norms<DUAL == 3, TB>(p, nrm_t, nrm_s, do_tn, do_sn, b, i0, j0, wsrc, wtgt);
        } else {
            SIMNN_TILE_LOOP(false)
        }

        if ((dbg & 7) == 1 || (dbg & 7) == 7) {
            float sacc = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < TB; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += acc[a][c][r];
            if (sacc == 1.2345f) p.pb[0] = sacc;
            continue;
        }
        // the ring slot of the stage computed last takes no LDS-DMA before the next tile's first stage
        float* const free_slot = reinterpret_cast<float*>(smem + ((r_slot + NBUF - 1) % NBUF) * PSTAGE);
        constexpr bool full = !EDGE;         // (EDGE launches: the masked variants of the reductions)
#ifdef DM_EXPERIMENTS
        if (DUAL && (p.dbg & 0x1000)) {                  // ablation (wrong results): no row-direction reduction
            if (acc[0][0][0] == 1.2345f) p.pb[0] = acc[1][1][1];
        } else
#endif
        simnn_tail<full, TT, ((dbg & 7) == 4 || (dbg & 7) == 6) ? 4 : 0, (DUAL == 3 ? 1 : DUAL), TB>(
            p, acc, nrm_t, nrm_s, do_tn, do_sn, b, i0, j0, ts_, lane, wsrc, wtgt, bias_lds + (n & 1) * BSLOT);
        if constexpr (TRACE) {
            if (tr_on) {
                unsigned long long te;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te));
                if (lane == 0) { tr_lds[tr_n < 256 ? tr_n : 255] = te; }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                for (int q = lane; q < 256; q += 64) p.trace[wave * 256 + q] = (q <= tr_n) ? tr_lds[q] : 0ull;
            }
        }
#ifdef DM_EXPERIMENTS
        // ablation (wrong results): no column-direction reduction.  (r04 - r05 this test sat in front of the TRACE block above and
        // its `else` bound to THAT: the flag did nothing, "no column reductions" timed the full kernel; found in r06.)
        if (DUAL == 3 && (p.dbg & 0x2000)) {
            if (acc[0][0][0] == 1.2345f) p.pb[0] = acc[1][1][1];
        } else
#endif
        if (DUAL == 3) {
            // transposes go through the free slot (8 waves: seven of them, the eighth has its own buffer)
            float* tb = (NW == 8 && wave == 7) ? tb_extra : free_slot + wave * (32 * 36);
            simnn_tail_cols<NWT, TB, full>(p, acc, nrm_t, nrm_s, do_tn, do_sn, b, i0, j0, tt_, tb, bias_lds + (n & 1) * BSLOT + 512, lane, wsrc, wtgt);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this wave is done with its buffer ...
            __builtin_amdgcn_s_barrier();                // ... and no wave starts the next tile's DMA into the slot before all are
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);                          // nothing of this workgroup may still be in flight
#undef SIMNN_TILE_LOOP
#undef SIMNN_STAGE
#undef SIMNN_TRACE_FLUSH
#undef SIMNN_STAMP
#undef SIMNN_PIN
#undef SIMNN_SYNC
#undef SIMNN_MMA
#undef SIMNN_READ
#undef SIMNN_DMA_NEXT
#undef SIMNN_DMA1
#undef SIMNN_DMA_TILE
}

#ifdef DM_EXPERIMENTS
static unsigned long long* g_simnn_trace = nullptr;
// experiments: the stage timeline of the last traced launch (8 waves x 256 stamps; tools/simnn_trace.py)
extern "C" int dm_debug_simnn_trace(dm_ctx* ctx, unsigned long long* out) {
    if (!ctx || !out || !g_simnn_trace) return DM_EINVAL;
    DM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    DM_CHECK_HIP(ctx, hipMemcpy(out, g_simnn_trace, 8 * 256 * 8, hipMemcpyDeviceToHost));
    return DM_OK;
}
#endif

// one key set of a tile pass: its partials in, the arg-max and the queue of ambiguous rows out.  Up to four key sets (the four
// maps of dm_fm_to_p2p: two per direction, with their own geometry) are merged by ONE launch: blockIdx.z selects the set.
struct simnn_merge_set {
    const float* pb; const int32_t* pj; const float* ps;   // (B, nparts, Npad)
    int nparts, N, Npad;
    const float* norm2;                              // (B, N) |row|^2 of the rows being reduced
    const unsigned int* max2;                        // (B) max |row|^2 of the other operand, float bits
    const float* tau_add; const float* tau_mul;      // two-key pass: max |bias| for the biased key, max scale for the scaled key
    const double* zero_if;                           // nullable (B, N): the answer is index 0 where this is 0 (an all-zero column)
    int32_t* nn; int32_t* flag_count; int32_t* flag_list; float* flag_thr;
    float* best; float* margin;                      // nullable (B, N)
};
struct simnn_merge_sets { simnn_merge_set s[4]; };
__global__ __launch_bounds__(256) void simnn_merge_kernel(simnn_merge_sets sets, float tau_scale, const int32_t* __restrict__ force_flag) {
    const simnn_merge_set& s = sets.s[blockIdx.z];
    const int b = blockIdx.y;
    const int i0 = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i0 < s.N;
    const int i = valid ? i0 : s.N - 1;              // (lanes past the end repeat the last row and store nothing: the queue append below is per wave)
    float bv = DM_NEG_INF_F32, sv = DM_NEG_INF_F32;
    int bj = DM_IDX_NONE;
    // (what the row's bound needs is requested before the partials, not behind them: one round trip less)
    const long long o = (long long)b * s.N + i;
    const bool zero = s.zero_if && s.zero_if[o] == 0.0;
    const float rn2 = s.norm2[o], mx2 = __uint_as_float(s.max2[b]);
    const float tmul = s.tau_mul ? s.tau_mul[b] : 1.0f, tadd = s.tau_add ? s.tau_add[b] : 0.0f;
    const bool forced = force_flag && force_flag[b] != 0;   // the caller could not bound the error for this pair: re-score everything
    // (loads of eight partials ahead of their merges: the loop is a chain of L2 round trips otherwise)
    int q = 0;
    for (; q + 8 <= s.nparts; q += 8) {
        float vb[8], vs[8];
        int vj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long op = ((long long)b * s.nparts + q + u) * s.Npad + i;
            vb[u] = s.pb[op]; vj[u] = s.pj[op]; vs[u] = s.ps[op];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) top2_merge(bv, bj, sv, vb[u], vj[u], vs[u]);
    }
    for (; q < s.nparts; ++q) {
        const long long op = ((long long)b * s.nparts + q) * s.Npad + i;
        top2_merge(bv, bj, sv, s.pb[op], s.pj[op], s.ps[op]);
    }
    const float m = bv - sv;
    const float tau = tau_scale * (sqrtf(rn2 * mx2) * tmul + tadd);
    if (valid) {
        s.nn[o] = (zero || bj == DM_IDX_NONE) ? 0 : bj;
        if (!zero && s.best) s.best[o] = bv;
        if (!zero && s.margin) s.margin[o] = m;
    }
    // queue of the rows to re-score: one atomic per wave that holds any (r04: one per flagged row -- thousands of returning atomics
    // on one address per launch, served one after the other)
    const bool flag = valid && !zero && (forced || !(m > tau));
    const unsigned long long fm = __ballot(flag);
    if (fm) {                                                     // uniform per wave
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0) base = atomicAdd(s.flag_count, __popcll(fm));
        base = __builtin_amdgcn_readfirstlane(base);
        if (flag) {
            const int pos = base + __popcll(fm & ((1ull << lane) - 1ull));
            s.flag_list[pos] = (int32_t)o;
            s.flag_thr[pos] = forced ? DM_NEG_INF_F32 : bv - tau;   // candidates scoring below this (in fp32) cannot be the float64 argmax
        }
    }
}

// float64 re-evaluation of the flagged rows: one workgroup per flagged row (grid-stride over the list).  Only the
// blocks of 32 source rows that can hold a candidate whose fp32 score reaches (best - tau) can contain the float64 argmax
// (every fp32 score is within tau/2 of the exact one): dm_simnn_keep, from the pass's own partials.  Such a block is re-scored
// exactly by the whole workgroup: 8 lanes per
// candidate, each wave instruction reads 8 x 128 contiguous bytes (fully used cache lines); fp16 products are exact
// in f64 and the summation order is fixed, so duplicated rows give identical scores and the lowest index wins.
__global__ __launch_bounds__(256) void simnn_fixup_kernel(const _Float16* __restrict__ Ftgt, const _Float16* __restrict__ Fsrc,
                                                          int N2, int N1, int D, const float* __restrict__ qpb,
                                                          const int32_t* __restrict__ qpj, const float* __restrict__ qps, int nparts, int pw,
                                                          int N2pad, const int32_t* __restrict__ flag_count,
                                                          const int32_t* __restrict__ flag_list,
                                                          const float* __restrict__ flag_thr, int32_t* __restrict__ nn) {
    extern __shared__ __attribute__((aligned(16))) double trow[];   // D doubles + 4 (value) + 4 ints
    __shared__ unsigned long long cmask[4];
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
        // A flagged row is a chain of dependent round trips, not arithmetic: the block filter of the first 256 blocks (dm_simnn_keep
        // with its three loads side by side) is requested together with the target row, and a kept block's loads are all in flight
        // at once (up to twelve 16-byte pieces per lane: D <= 768 in one round; r04: three dependent loads for the filter, four
        // pieces per round).
        const int nsub = nparts * (pw / 32);
        auto keep_of = [&](int sbt) -> bool {
            const int q = min((sbt * 32) / pw, nparts - 1);
            const long long oq = ((long long)b * nparts + q) * N2pad + i;
            const float vs = qps[oq], vb = qpb[oq];
            const int vj = qpj[oq];
            return sbt < nsub && (vs >= thr || (vb >= thr && (vj >> 5) == sbt));
        };
        bool keep = keep_of((int)threadIdx.x);
        __syncthreads();
        for (int k = threadIdx.x; k < D; k += 256) trow[k] = (double)tr[k];
        double bv = -DM_INF_F64;
        int bj = DM_IDX_NONE;
        // candidate blocks: only the blocks that can still hold the arg-max are visited, in ascending order
        for (int sb0 = 0; sb0 < nsub; sb0 += 256) {
            if (sb0) keep = keep_of(sb0 + (int)threadIdx.x);
            const unsigned long long km = __ballot(keep);
            if (lane == 0) cmask[wave] = km;
            __syncthreads();                                      // (also: trow is complete)
            for (int w = 0; w < 4; ++w) {
                unsigned long long mm = cmask[w];                 // uniform
                while (mm) {
                    const int sb = sb0 + w * 64 + __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    const int j = sb * 32 + cand;
                    const _Float16* sr = Fsrc + ((long long)b * N1 + min(j, N1 - 1)) * D;
                    double sacc = 0.0;
                    // pieces k = 8 part, + 64, ... of the candidate's row, NV at a time (D % 8 == 0 is guaranteed by the caller; a
                    // piece past the row re-reads the lane's first piece and is not added)
                    constexpr int NV = 12;
                    for (int k0 = part * 8; k0 < D; k0 += 64 * NV) {
                        f16x8 v[NV];
#pragma unroll
                        for (int w4 = 0; w4 < NV; ++w4) {
                            const int k = k0 + 64 * w4;
                            v[w4] = *reinterpret_cast<const f16x8*>(sr + (k < D ? k : part * 8));
                        }
#pragma unroll
                        for (int w4 = 0; w4 < NV; ++w4) {
                            const int k = k0 + 64 * w4;
                            if (k < D) {
#pragma unroll
                                for (int u = 0; u < 8; ++u) sacc = fma((double)v[w4][u], trow[k + u], sacc);
                            }
                        }
                    }
                    sacc += __shfl_xor(sacc, 1);
                    sacc += __shfl_xor(sacc, 2);
                    sacc += __shfl_xor(sacc, 4);
                    if (j < N1 && sacc > bv) { bv = sacc; bj = j; }   // blocks ascend: strict keeps the lowest index
                }
            }
            __syncthreads();
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

size_t dm_simnn_ctl_bytes(int B) { return 2 * dm_align_up((size_t)B * 4) + 1024; }
size_t dm_simnn_ws_bytes(int B, int N2, int N1, int dual) {
    const size_t N2pad = pad_to(N2, ST), tilesS = dm_cdiv(N1, ST);
    const size_t np = (size_t)B * 2 * tilesS * N2pad;          // row partials: two source halves per tile
    const size_t keyset = 3 * dm_align_up(np * 4) + dm_align_up((size_t)B * N2 * 4) * 2 + 512;
    size_t total = (dual ? 2 : 1) * keyset + dm_align_up((size_t)B * N2 * 4) + 2 * dm_align_up((size_t)B * 4) + 8192;
    if (dual == 3) {                                      // the two column-direction key sets
        const size_t N1pad = pad_to(N1, ST), cp = (size_t)B * (N2pad / 64) * N1pad;
        total += 2 * (3 * dm_align_up(cp * 4) + 2 * dm_align_up((size_t)B * N1 * 4)) + dm_align_up((size_t)B * N1 * 4) + 4096;
    }
    return total;
}

// can the two-key pass run on these sizes (interior 256 x 256 tiles, contraction a multiple of a stage and deep enough
// for the ring)?
bool dm_simnn_dual_ok(const dm_ctx* ctx, int N2, int N1, int D, bool padded) {
    return ctx->opt_simnn_pipe && (padded || (N2 % ST == 0 && N1 % ST == 0)) && D % PBK == 0 && D >= 5 * PBK;
}

// Tile kernel + merge: fp32 scores, top-2 per target row, the rows whose margin is inside the error bound queued for an
// exact re-evaluation by the caller.  Workspace comes from the context arena (the caller reserved dm_simnn_ws_bytes).
// rel_extra: additional relative error of a score (in units of |t_i| max_j |s_j|) on top of the fp32 accumulation bound.
// dual (nullable): a second reduction of the same products, see dm_simnn_dual; nn21 / q then belong to key A.
int dm_simnn_core(dm_ctx* ctx, int B, int N2, int N1, int D, const _Float16* Ftgt, int ldT, const _Float16* Fsrc, int ldS,
                  float rel_extra,
                  const int32_t* force_flag, int32_t* nn21, float* best, float* margin, dm_simnn_queue* q,
                  const dm_simnn_dual* dual, dm_simnn_ext* ext) {
    simnn_params p;
    memset(&p, 0, sizeof(p));
    p.Ftgt = Ftgt; p.Fsrc = Fsrc;
    p.N2 = N2; p.N1 = N1; p.D = D; p.N2pad = pad_to(N2, ST);
    p.ldT = ldT; p.ldS = ldS;
    // key-set passes may run on operands (and per-row term arrays) padded to whole 256-tiles: any N2, N1
    const bool padded = dual && dual->padded;
    p.rowsT = padded ? pad_to(N2, ST) : N2; p.rowsS = padded ? pad_to(N1, ST) : N1;
    p.tilesT = p.N2pad / ST; p.tilesS = dm_cdiv(N1, ST);
    p.total = B * p.tilesT * p.tilesS;
    p.band = p.tilesT;                       // (edge kernel: row-major)
    p.dbg = dm_knob("DM_SIMNN_DEBUG", 0);
    const int nparts = 2 * p.tilesS;                  // row partials: one per wave = per half (128 source rows) of a tile
    const size_t np = (size_t)B * nparts * p.N2pad;
    p.pb = (float*)dm_ws_take(ctx, np * 4);
    p.pj = (int32_t*)dm_ws_take(ctx, np * 4);
    p.ps = (float*)dm_ws_take(ctx, np * 4);
    p.tnorm2 = (float*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    int32_t* flag_list = (int32_t*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    float* flag_thr = (float*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    // per-pair norm maxima and the queue counters: one block, one memset
    const size_t ctl_bytes = dm_simnn_ctl_bytes(B);
    char* ctl = (ext && ext->ctl) ? (char*)ext->ctl : (char*)dm_ws_take(ctx, ctl_bytes);
    if (!p.pb || !p.pj || !p.ps || !p.tnorm2 || !flag_list || !flag_thr || !ctl)
        return dm_fail(ctx, DM_ENOMEM, "simnn: workspace not reserved");
    p.smax2 = (unsigned int*)ctl;
    p.tmax2 = (unsigned int*)(ctl + dm_align_up((size_t)B * 4));
    int32_t* flag_count = (int32_t*)(ctl + 2 * dm_align_up((size_t)B * 4));
    int32_t* flag_list2 = nullptr; float* flag_thr2 = nullptr; int32_t* flag_count2 = nullptr;
    const dm_simnn_cols* cols = dual ? dual->cols : nullptr;
    int32_t* cflag_list[2] = {nullptr, nullptr}; float* cflag_thr[2] = {nullptr, nullptr};
    int32_t* cflag_count[2] = {flag_count + 128, flag_count + 192};
    const bool single = dual && dual->single;         // key A alone on split rows
    if (single) {
        if (!dm_simnn_dual_ok(ctx, N2, N1, D, padded) || !dual->bias || dual->cols || dual->scale)
            return dm_fail(ctx, DM_EINVAL, "simnn: the biased-key pass needs interior tiles, D %% 32 == 0, D >= 160");
        p.bias = dual->bias;
    } else if (dual) {
        if (!dm_simnn_dual_ok(ctx, N2, N1, D, padded) || !dual->bias || !dual->nn_b || !dual->q_b)
            return dm_fail(ctx, DM_EINVAL, "simnn: the two-key pass needs interior tiles, D %% 32 == 0, D >= 160");
        p.bias = dual->bias; p.scale = dual->scale;
        p.pb_2 = (float*)dm_ws_take(ctx, np * 4);
        p.pj_2 = (int32_t*)dm_ws_take(ctx, np * 4);
        p.ps_2 = (float*)dm_ws_take(ctx, np * 4);
        flag_list2 = (int32_t*)dm_ws_take(ctx, (size_t)B * N2 * 4);
        flag_thr2 = (float*)dm_ws_take(ctx, (size_t)B * N2 * 4);
        flag_count2 = flag_count + 64;
        if (!p.pb_2 || !p.pj_2 || !p.ps_2 || !flag_list2 || !flag_thr2)
            return dm_fail(ctx, DM_ENOMEM, "simnn: workspace not reserved");
    }
    if (cols) {
        if (!dual->scale || !cols->biasT || !cols->nn_a || !cols->nn_b || !cols->q_a || !cols->q_b)
            return dm_fail(ctx, DM_EINVAL, "simnn: both-directions pass: missing operand");
        p.biasT = cols->biasT;
        p.N1pad = pad_to(N1, ST);
        const size_t cp = (size_t)B * (p.N2pad / 64) * p.N1pad;
        for (int kd = 0; kd < 2; ++kd) {
            p.cb[kd] = (float*)dm_ws_take(ctx, cp * 4);
            p.cj[kd] = (int32_t*)dm_ws_take(ctx, cp * 4);
            p.cs[kd] = (float*)dm_ws_take(ctx, cp * 4);
            cflag_list[kd] = (int32_t*)dm_ws_take(ctx, (size_t)B * N1 * 4);
            cflag_thr[kd] = (float*)dm_ws_take(ctx, (size_t)B * N1 * 4);
            if (!p.cb[kd] || !p.cj[kd] || !p.cs[kd] || !cflag_list[kd] || !cflag_thr[kd])
                return dm_fail(ctx, DM_ENOMEM, "simnn: workspace not reserved");
        }
        p.snorm2 = (float*)dm_ws_take(ctx, (size_t)B * N1 * 4);
        if (!p.snorm2) return dm_fail(ctx, DM_ENOMEM, "simnn: workspace not reserved");
    }

    if (!(ext && ext->ctl)) DM_CHECK_HIP(ctx, hipMemsetAsync(ctl, 0, ctl_bytes, ctx->stream));
    const size_t lds_edge = (size_t)4 * ST * SBK * sizeof(_Float16);
    const bool interior = padded || (N2 % ST == 0 && N1 % ST == 0);
    // DM_EXPERIMENTS: DM_SIMNN_DEBUG = variant bits XV (simnn_pipe_kernel) + 256 / 512 for the 8-wave / 4-wave shape
    // (both directions: the 8-wave shape; p2p_split = 3 selects 4 waves x 2 workgroups per CU, whose second workgroup covers
    // part of the epilogue but whose 1.5x operand traffic costs as much: config 2 1.737 vs 1.729 ms, config 5 20.0 vs 19.2 ms)
    // both directions: the 4-wave shape (two workgroups per CU: one's epilogue runs under the other's main loop, at 1.5x the operand
    // traffic) wins while a pair's operands stay in an XCD's L2 (-5 % at N = 2048, +2.5 % at N = 8192); p2p_split = 3 / 4 force
    // the 4-wave / 8-wave shape
    const bool ops_in_l2 = ((size_t)N2 * ldT + (size_t)N1 * ldS) * 2 <= ((size_t)3 << 20);
    const int WT = cols ? ((ctx->opt_p2p_split == 3 || (ctx->opt_p2p_split == 2 && ops_in_l2 && !ctx->opt_simnn_big)) ? 2 : 4)
                        : (single ? (ctx->opt_simnn1_wt == 2 ? 2 : 4) : (dual ? 4 : ((p.dbg & 256) ? 4 : ((p.dbg & 512) ? 2 : SIMNN_PRODUCT_WT))));
    // (the stage loop peels its first and last stages: the contraction must be at least ring depth + 1 stages deep)
    const bool aligned = (N2 % ST == 0 && N1 % ST == 0);
    const int TB = (WT == 4 && ctx->opt_simnn_big && aligned) ? 4 : 2;       // 32-row target blocks per wave
    if (interior && ctx->opt_simnn_pipe && D % PBK == 0 && D >= (WT == 4 ? 5 : 4) * PBK) {
        const int TT = 64 * WT;
        p.tilesT = p.N2pad / TT;
        const size_t lds_pipe = simnn_pipe_lds(WT, cols ? 3 : (dual ? 1 : 0));
        // workgroups that fit a CU at once walk the tiles (opt_simnn_persist: 0 = one workgroup per tile, 1 = as many
        // workgroups as are resident when there are more tiles than that, n > 1 = n workgroups (tests))
        const int ncu = ctx->n_cu > 0 ? ctx->n_cu : 256;
        const int resident = ncu * (WT == 4 ? 1 : 2);
        int rc = DM_OK, grid = 0;
#define SIMNN_LAUNCH_XV1(XV_, WT_, DUAL_, NAME_)                                                                       \
        {                                                                                                              \
            rc = dm_grant_lds(ctx, (const void*)simnn_pipe_kernel<XV_, WT_, DUAL_>, lds_pipe);                         \
            if (rc) return rc;                                                                                         \
            DM_LAUNCH(ctx, NAME_, (simnn_pipe_kernel<XV_, WT_, DUAL_>), dim3(grid), dim3(128 * WT_), lds_pipe, p);     \
        }
#ifdef DM_EXPERIMENTS   /* dm_set_option("simnn_prio", 1): the s_setprio variant (measured r06: no change on any workload, profiles/r06_simnn_setprio_ab.txt) */
#define SIMNN_LAUNCH_XV(XV_, WT_, DUAL_, NAME_)                                                                        \
        { if (ctx->opt_simnn_prio) SIMNN_LAUNCH_XV1((XV_) | 8192, WT_, DUAL_, NAME_) else SIMNN_LAUNCH_XV1(XV_, WT_, DUAL_, NAME_) }
#else
#define SIMNN_LAUNCH_XV(XV_, WT_, DUAL_, NAME_) SIMNN_LAUNCH_XV1(XV_, WT_, DUAL_, NAME_)
#endif
#define SIMNN_LAUNCH_EDGE(WT_, DUAL_, NAME_)                                                                           \
        {                                                                                                              \
            rc = dm_grant_lds(ctx, (const void*)simnn_pipe_kernel<SIMNN_PRODUCT_XV, WT_, DUAL_, 2, true>, lds_pipe);   \
            if (rc) return rc;                                                                                         \
            DM_LAUNCH(ctx, NAME_, (simnn_pipe_kernel<SIMNN_PRODUCT_XV, WT_, DUAL_, 2, true>), dim3(grid), dim3(128 * WT_), lds_pipe, p); \
        }
#define SIMNN_LAUNCH_BIG(DUAL_, NAME_)                                                                                 \
        {                                                                                                              \
            rc = dm_grant_lds(ctx, (const void*)simnn_pipe_kernel<SIMNN_PRODUCT_XV, 4, DUAL_, 4>, lds_pipe);           \
            if (rc) return rc;                                                                                         \
            DM_LAUNCH(ctx, NAME_, (simnn_pipe_kernel<SIMNN_PRODUCT_XV, 4, DUAL_, 4>), dim3(grid), dim3(256), lds_pipe, p); \
        }
        // One launch walks a rectangle of the tilesT x tilesS tile grid.  Aligned sizes: the whole grid.  Padded operands: the
        // whole tiles first, then the strips that reach into the padding with the masking (EDGE) instantiation.
        const int fullT = padded ? N2 / TT : p.tilesT, fullS = padded ? N1 / ST : p.tilesS;
        const int rects[3][5] = {{0, fullT, 0, fullS, 0}, {fullT, p.tilesT - fullT, 0, p.tilesS, 1}, {0, fullT, fullS, p.tilesS - fullS, 1}};
        for (int rq = 0; rq < 3; ++rq) {
            p.rt0 = rects[rq][0]; p.rnT = rects[rq][1]; p.rs0 = rects[rq][2]; p.rnS = rects[rq][3];
            const bool edge = rects[rq][4] != 0;
            if (p.rnT <= 0 || p.rnS <= 0) continue;
            p.total = B * p.rnT * p.rnS;
            // tile order: bands of opt_simnn_band tile rows, column-major inside (0: one tile row per band = row-major order)
            p.band = ctx->opt_simnn_band <= 0 ? 1 : (p.rnT < ctx->opt_simnn_band ? p.rnT : ctx->opt_simnn_band);
            const int want = ctx->opt_simnn_persist > 1 ? ctx->opt_simnn_persist : (ctx->opt_simnn_persist ? resident : p.total);
            grid = want < p.total ? want : p.total;
            if (edge) {
                if (cols) { if (WT == 4) SIMNN_LAUNCH_EDGE(4, 3, "simnn4_f16_mfma") else SIMNN_LAUNCH_EDGE(2, 3, "simnn4_f16_mfma") }
                else if (single) { if (WT == 4) SIMNN_LAUNCH_EDGE(4, 4, "simnn1_f16_mfma") else SIMNN_LAUNCH_EDGE(2, 4, "simnn1_f16_mfma") }
                else if (dual->scale) SIMNN_LAUNCH_EDGE(4, 1, "simnn2_f16_mfma")
                else SIMNN_LAUNCH_EDGE(4, 2, "simnn2_f16_mfma")
                continue;
            }
        if (TB == 4) {
            if (cols) SIMNN_LAUNCH_BIG(3, "simnn4_f16_mfma")
            else if (dual && dual->scale) SIMNN_LAUNCH_BIG(1, "simnn2_f16_mfma")
            else if (dual) SIMNN_LAUNCH_BIG(2, "simnn2_f16_mfma")
            else SIMNN_LAUNCH_BIG(0, "simnn_f16_mfma")
        }
#ifdef DM_EXPERIMENTS
        else if (dm_knob("DM_SIMNN_ILV", 0) && WT == 4 && (cols || single)) {    // the key-set kernels with interleaved issue
            if (cols) SIMNN_LAUNCH_XV(SIMNN_PRODUCT_XV + 4096, 4, 3, "simnn4_f16_mfma")
            else SIMNN_LAUNCH_XV(SIMNN_PRODUCT_XV + 4096, 4, 4, "simnn1_f16_mfma")
        }
#endif
        else if (cols) {
            if (WT == 4) SIMNN_LAUNCH_XV(SIMNN_PRODUCT_XV, 4, 3, "simnn4_f16_mfma")
            else SIMNN_LAUNCH_XV(SIMNN_PRODUCT_XV, 2, 3, "simnn4_f16_mfma")
        } else if (single) {
            if (WT == 4) SIMNN_LAUNCH_XV(SIMNN_PRODUCT_XV, 4, 4, "simnn1_f16_mfma")
            else SIMNN_LAUNCH_XV(SIMNN_PRODUCT_XV, 2, 4, "simnn1_f16_mfma")
        } else if (dual) {
            if (dual->scale) SIMNN_LAUNCH_XV(SIMNN_PRODUCT_XV, 4, 1, "simnn2_f16_mfma")
            else SIMNN_LAUNCH_XV(SIMNN_PRODUCT_XV, 4, 2, "simnn2_f16_mfma")
        } else {
#ifdef DM_EXPERIMENTS
        if (p.dbg == 0x20000) {         // five-slot ring, barrier at the end of the stage
            const size_t lds5 = simnn_pipe_lds(4, -2);
            rc = dm_grant_lds(ctx, (const void*)simnn_pipe_kernel<SIMNN_PRODUCT_XV + 2048, 4, 0>, lds5);
            if (rc) return rc;
            DM_LAUNCH(ctx, "simnn_f16_mfma", (simnn_pipe_kernel<SIMNN_PRODUCT_XV + 2048, 4, 0>), dim3(grid), dim3(512), lds5, p);
        } else
        if (p.dbg == 0x10000) {         // stage timeline (tools/simnn_trace.py): the product variant + stamps, 8 waves
            static unsigned long long* trace_dev = nullptr;
            if (!trace_dev) DM_CHECK_HIP(ctx, hipMalloc((void**)&trace_dev, 8 * 256 * 8));
            p.trace = trace_dev;
            g_simnn_trace = trace_dev;
            const size_t lds_tr = simnn_pipe_lds(4, -1);
            rc = dm_grant_lds(ctx, (const void*)simnn_pipe_kernel<SIMNN_PRODUCT_XV + 1024, 4, 0>, lds_tr);
            if (rc) return rc;
            DM_LAUNCH(ctx, "simnn_f16_mfma", (simnn_pipe_kernel<SIMNN_PRODUCT_XV + 1024, 4, 0>), dim3(grid), dim3(512), lds_tr, p);
        } else
        switch (p.dbg) {
#define SIMNN_CASE(XV_) case 512 + XV_: SIMNN_LAUNCH_XV(XV_, 2, 0, "simnn_f16_mfma") break; case 256 + XV_: SIMNN_LAUNCH_XV(XV_, 4, 0, "simnn_f16_mfma") break;
            SIMNN_CASE(0) SIMNN_CASE(1) SIMNN_CASE(7) SIMNN_CASE(9)
            SIMNN_CASE(16) SIMNN_CASE(32) SIMNN_CASE(64)
            SIMNN_CASE(64 + 1) SIMNN_CASE(64 + 5) SIMNN_CASE(64 + 7) SIMNN_CASE(64 + 9) SIMNN_CASE(64 + 16) SIMNN_CASE(64 + 32)
            SIMNN_CASE(64 + 32 + 1) SIMNN_CASE(64 + 32 + 7) SIMNN_CASE(64 + 32 + 9)
            SIMNN_CASE(64 + 48) SIMNN_CASE(64 + 48 + 1) SIMNN_CASE(64 + 48 + 7)
            SIMNN_CASE(64 + 2) SIMNN_CASE(64 + 4) SIMNN_CASE(64 + 6)
            SIMNN_CASE(64 + 128) SIMNN_CASE(64 + 128 + 1) SIMNN_CASE(64 + 128 + 9)
#undef SIMNN_CASE
            default: SIMNN_LAUNCH_XV(SIMNN_PRODUCT_XV + SIMNN_PRODUCT_ILV0, SIMNN_PRODUCT_WT, 0, "simnn_f16_mfma") break;
        }
#else
        SIMNN_LAUNCH_XV(SIMNN_PRODUCT_XV + SIMNN_PRODUCT_ILV0, SIMNN_PRODUCT_WT, 0, "simnn_f16_mfma")
#endif
        }
        }
#undef SIMNN_LAUNCH_BIG
#undef SIMNN_LAUNCH_EDGE
#undef SIMNN_LAUNCH_XV
#undef SIMNN_LAUNCH_XV1
    } else {
        int rc = dm_grant_lds(ctx, (const void*)simnn_edge_kernel, lds_edge);
        if (rc) return rc;
        DM_LAUNCH(ctx, "simnn_f16_mfma", simnn_edge_kernel, dim3(p.total), dim3(512), lds_edge, p);
    }
    // twice the error bound of a score, relative to |t_i| max_j |s_j|: fp32 accumulation (D exact products,
    // D (1 + 1/16) additions, unit roundoff 2^-23, safe for round-to-nearest and for truncating adders) + the caller's
    // own term, plus 4 * 2^-19 for the 4 mantissa bits the reduction keys give up (simnn_tail: the best key, the second key,
    // its truncation, and the truncated partials the fix-up filter compares with the threshold); 1 % slack for the
    // fp32 norms.  Two-key pass: one more rounding (2^-23 covers the fp32 bias / scale and the add / multiply); key A is
    // bounded relative to |t_i| max|s_j| + max|bias_j|, key B relative to |t_i| max|s_j| max scale_j.
    // (key-set passes read split rows: D halves per row stand for 3 D / 2 products, hx.hy + hx.ly + lx.hy per index)
    const float nprod = dual ? 1.5f * (float)D : (float)D;
    // DM_EXPERIMENTS builds: DM_TAU_EXTRA_LOG2 = e adds 2^-e to the relative error bound (results stay exact, more rows take the exact
    // path): what a first pass on the high halves alone (error 2^-10 |t||s|) would requeue -- tools/one_product_pass_experiment.py
    if (const int e_ = dm_knob("DM_TAU_EXTRA_LOG2", 0)) rel_extra += ldexpf(1.0f, -e_);
    const float tau_scale = 2.0f * 1.01f * (nprod * (1.0f + 1.0f / 16.0f) * 1.1920929e-7f + 2.0f * 1.9073486e-6f + rel_extra +
                                            (dual ? 1.1920929e-7f : 0.0f));
    if (ext) {
        ext->pb = p.pb; ext->pj = p.pj; ext->ps = p.ps; ext->nparts = nparts; ext->pw = 128; ext->N2pad = p.N2pad;
        ext->tnorm2 = p.tnorm2; ext->smax2 = p.smax2; ext->tau_scale = tau_scale;
        if (ext->skip_merge) return DM_OK;
    }
    simnn_merge_sets sets;
    memset(&sets, 0, sizeof(sets));
    int nsets = 0, maxN = N2;
    sets.s[nsets++] = simnn_merge_set{p.pb, p.pj, p.ps, nparts, N2, p.N2pad, p.tnorm2, p.smax2, dual ? dual->tau_add : nullptr, nullptr,
                                      nullptr, nn21, flag_count, flag_list, flag_thr, best, margin};
    if (dual && !single)
        sets.s[nsets++] = simnn_merge_set{p.pb_2, p.pj_2, p.ps_2, nparts, N2, p.N2pad, p.tnorm2, p.smax2, nullptr, dual->tau_mul,
                                          nullptr, dual->nn_b, flag_count2, flag_list2, flag_thr2, nullptr, nullptr};
    if (cols) {
        // the column direction: "targets" are the source rows, partials per (tile row, target quarter), bound from |s_j| max |t_i|
        const int cparts = p.N2pad / (32 * TB);          // one partial per wave along the targets
        sets.s[nsets++] = simnn_merge_set{p.cb[0], p.cj[0], p.cs[0], cparts, N1, p.N1pad, p.snorm2, p.tmax2, cols->tau_add, nullptr,
                                          nullptr, cols->nn_a, cflag_count[0], cflag_list[0], cflag_thr[0], nullptr, nullptr};
        sets.s[nsets++] = simnn_merge_set{p.cb[1], p.cj[1], p.cs[1], cparts, N1, p.N1pad, p.snorm2, p.tmax2, nullptr, nullptr,
                                          cols->zero_b, cols->nn_b, cflag_count[1], cflag_list[1], cflag_thr[1], nullptr, nullptr};
        maxN = N1 > N2 ? N1 : N2;
    }
    DM_LAUNCH(ctx, "simnn_merge", simnn_merge_kernel, dim3(dm_cdiv(maxN, 256), B, nsets), dim3(256), 0, sets, tau_scale, force_flag);
    ctx->last_flag_counts = flag_count; ctx->last_flag_sets = nsets;
    *q = dm_simnn_queue{p.pb, p.pj, p.ps, nparts, 128, p.N2pad, flag_count, flag_list, flag_thr};
    if (dual && !single) *dual->q_b = dm_simnn_queue{p.pb_2, p.pj_2, p.ps_2, nparts, 128, p.N2pad, flag_count2, flag_list2, flag_thr2};
    if (cols) {
        *cols->q_a = dm_simnn_queue{p.cb[0], p.cj[0], p.cs[0], p.N2pad / (32 * TB), 32 * TB, p.N1pad, cflag_count[0], cflag_list[0], cflag_thr[0]};
        *cols->q_b = dm_simnn_queue{p.cb[1], p.cj[1], p.cs[1], p.N2pad / (32 * TB), 32 * TB, p.N1pad, cflag_count[1], cflag_list[1], cflag_thr[1]};
    }
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
              (const _Float16*)Fsrc, N2, N1, D, q.pb, q.pj, q.ps, q.nparts, q.pw, q.Npad, q.flag_count, q.flag_list, q.flag_thr, nn21);
    return DM_OK;
}

extern "C" int dm_last_requeued_rows(dm_ctx* ctx, int out[4]) {
    if (!ctx || !out) return DM_EINVAL;
    for (int q = 0; q < 4; ++q) out[q] = -1;
    if (!ctx->last_flag_counts) return DM_OK;
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int32_t host[4 * 64];
    DM_CHECK_HIP(ctx, hipMemcpyAsync(host, ctx->last_flag_counts, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
    DM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int q = 0; q < ctx->last_flag_sets && q < 4; ++q) out[q] = host[64 * q];
    return DM_OK;
}

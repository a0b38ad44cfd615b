// Vertex map -> functional map (dm_p2p_to_fm) and ZoomOut refinement (dm_zoomout).
//
// Reference arithmetic reproduced (oracle/dm_oracle.py: p2p_to_fm, zoomout_refine):
//   C' = Phi2[:, :k2']^T (a2 * Phi1[p21, :k1'])                      pyFM/spectral/convert.py:39-48
//   repeat nit: p21 = NN(tree = Phi1[:, :k] C^T, query = Phi2[:, :k]) pyFM/refine/zoomout.py:40
//               C <- p2p_to_FM(p21) with k + step columns             pyFM/refine/zoomout.py:42
// The NN step is the fused G-tile kernel of dm_p2p.hip restricted to the row argmin; Phi2^T is
// staged K-major in float64 once for all iterations, the whole loop runs on the stream with no
// host synchronisation.
#include "dm_gemm_f64.h"
#include "dm_internal.h"
#include "dm_zoomfuse.h"

struct OutFM {
    double* direct; int ldc; long long strideC;     // nsplit == 1
    double* partial; int B, k2, k1;                 // (nsplit, B, k2, k1)
    __device__ __forceinline__ void store(int b, int split, int m, int c, double v) const {
        if (direct) direct[b * strideC + (long long)m * ldc + c] = v;
        else partial[(((long long)split * B + b) * k2 + m) * k1 + c] = v;
    }
};

__global__ __launch_bounds__(256) void splitk_reduce_fm_kernel(const double* __restrict__ partial, int nsplit, int B,
                                                               int k2, int k1, double* __restrict__ C, int ldc,
                                                               long long strideC) {
    const long long n = (long long)B * k2 * k1;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int q = 0; q < nsplit; ++q) s += partial[(long long)q * n + i];
    const int c = (int)(i % k1);
    const long long bm = i / k1;
    const int m = (int)(bm % k2);
    const int b = (int)(bm / k2);
    C[b * strideC + (long long)m * ldc + c] = s;
}

static inline int pad_to16(int x) { return (x + 15) / 16 * 16; }
// split-K by a fixed chunk of vertices: the summation order of a pair must not depend on the batch it is in
constexpr int FM_KCHUNK = 256;
static int fm_split(int B, int N2, int k1, int k2) {
    (void)B; (void)k1; (void)k2;
    return dm_cdiv(N2, FM_KCHUNK);
}

// ---------------------------------------------------------------------------------------------------------------------
// p2p_to_FM with register-resident output tiles and operands read straight from global memory (the default; the LDS-staged
// 64 x 64 tile kernel above with its split-K partials and reduce launch stays as p2pfm_direct = 0).
//
// The result is small (k2 x k1 <= 208 x 208) and the contraction long (N2 vertices), both operands are K-major as they
// lie in memory: row n of Phi2 is one contiguous run of the A operand of v_mfma_f64_16x16x4_f64 (lane l: A[m = l & 15][k = l >> 4]
// = Phi2[n0 + (l >> 4)][16 mb + (l & 15)]: four 128-byte runs per wave load), row p21[n] of Phi1 of the B operand.  So there is
// no LDS stage, no barrier and no operand conversion pass in the main loop: a wave owns RBW x (<= 7) blocks of 16 x 16
// (<= 112 accumulator registers), streams its own slice of the vertices with the loads of the next k-step in flight under
// the matrix instructions of the current one, and the S waves of a workgroup -- S slices of the same tile -- add their
// tiles up through LDS in a FIXED order at the end (no split-K partials in HBM, no reduce launch, no atomics: the sum of
// a pair is the same in every batch and every run).  All workgroups of a pair run on one XCD (block b -> XCD b % 8) and
// walk the vertices in step, so every row of Phi1 / Phi2 is fetched into that XCD's L2 once per slice position.
//   S = 8 slices while the map has at most 7 x 7 blocks (k <= 112: more, lighter waves), else 4; the tile of a wave (1 x c, 2 x c,
//   3 x 5, 5 x 3, 6 x 3 ... blocks) is chosen per launch by a cost model of the batch (launch_p2pfm_direct): it decides who
//   computes an entry, not in which order its terms are added.
template <typename TR>
struct p2pfm_args {
    const TR* Phi1; long long s1; int ld1; int N1;
    const double* Xs; long long sx; int ldx; int N2;        // mass2 * Phi2, float64, (B, N2 + 1, ldx): row N2 is zero, ldx % 16 == 0
    const int32_t* p21;
    int k1, k2;                       // columns (Phi1) / rows (Phi2) of the map
    double* C; int ldc; long long strideC;
    int B, TM, TC, rps;               // tiles = TM x TC per pair, vertices per slice (multiple of 4)
    int dbg;                          // DM_EXPERIMENTS builds only (0 in the product): 32 no reduction / stores, 64 three k-steps only, 128 every step reads the slice's first vertices
};

// Xs[b][n][m] = mass2[b][n] * Phi2[b][n][m] (one rounding, like the reference's A2 @ ... products), zero for m >= k2 and for the
// extra row n = N2 -- what the steps beyond a slice read.  Built once per ZoomOut call (the target basis does not change).
template <typename TR>
__global__ __launch_bounds__(256) void p2pfm_prescale_kernel(const TR* __restrict__ Phi2, int N2, int ld2, int k2, const double* __restrict__ mass2,
                                                            double* __restrict__ Xs, int ldx) {
    const int b = blockIdx.y;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)(N2 + 1) * ldx) return;
    const int n = (int)(e / ldx), m = (int)(e - (long long)n * ldx);
    double v = 0.0;
    if (n < N2 && m < k2) v = mass2[(long long)b * N2 + n] * (double)Phi2[((long long)b * N2 + n) * ld2 + m];
    Xs[(long long)b * (N2 + 1) * ldx + e] = v;
}
size_t dm_p2pfm_xs_bytes(int B, int N2, int k2) { return dm_align_up((size_t)B * (N2 + 1) * pad_to16(k2) * 8); }
template <typename TR>
int dm_p2pfm_prescale(dm_ctx* ctx, int B, int N2, int k2, const TR* Phi2, int ld2, const double* mass2, double* Xs) {
    const int ldx = pad_to16(k2);
    const long long n = (long long)(N2 + 1) * ldx;
    DM_LAUNCH(ctx, "p2pfm_prescale", p2pfm_prescale_kernel<TR>, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, Phi2, N2, ld2, k2, mass2, Xs, ldx);
    return DM_OK;
}
template int dm_p2pfm_prescale<float>(dm_ctx*, int, int, int, const float*, int, const double*, double*);
template int dm_p2pfm_prescale<double>(dm_ctx*, int, int, int, const double*, int, const double*, double*);

// raw buffer descriptor over `bytes` bytes at p: loads beyond the range return 0 (no clamping arithmetic), addresses are
// 32-bit offsets from a scalar base (one v_add per fragment instead of a 64-bit multiply-add: float64 matrix instructions do
// not overlap with vector ALU work on this part, every address instruction in the loop is paid in full)
typedef __amdgpu_buffer_rsrc_t p2pfm_rsrc_t;
__device__ __forceinline__ p2pfm_rsrc_t p2pfm_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
template <typename T> __device__ __forceinline__ T p2pfm_bload(p2pfm_rsrc_t rsrc, int voff);
template <> __device__ __forceinline__ double p2pfm_bload<double>(p2pfm_rsrc_t rsrc, int voff) {
    typedef int i32x2_t __attribute__((ext_vector_type(2)));
    const i32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 0, 0);
    return __hiloint2double(v[1], v[0]);
}
template <> __device__ __forceinline__ float p2pfm_bload<float>(p2pfm_rsrc_t rsrc, int voff) {
    return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0));
}

// RBW x CBW blocks per wave, exactly (no guard between the matrix instructions).  Where the blocks do not divide evenly the LAST
// group of row / column blocks starts early and overlaps its neighbour: the shared blocks are computed twice, by the same
// arithmetic in the same order, and stored twice with identical bits (a 16 x 16 block per 13 costs less than a branch per MFMA).
template <typename TR, int RBW, int CBW, int S>
__global__ __launch_bounds__(64 * S, 2) void p2pfm_direct_kernel(p2pfm_args<TR> a) {
    // dynamic LDS: first every wave's slice of the gather indices (S x rps ints), then, for the reduction, one 16 x 16 block of
    // every wave, double-buffered (2 x S x 256 doubles)
    extern __shared__ __attribute__((aligned(16))) double p2pfm_sm[];
    const int T = a.TM * a.TC;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = xcd + 8 * (slot / T);
    if (b >= a.B) return;
    const int tile = slot % T;
    const int tm = tile / a.TC, tc = tile - tm * a.TC;
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrb = dm_cdiv(a.k2, 16), ncb = dm_cdiv(a.k1, 16);
    const int mb0 = max(0, min(tm * RBW, nrb - RBW)), cb0 = max(0, min(tc * CBW, ncb - CBW));

    const p2pfm_rsrc_t rx = p2pfm_rsrc(a.Xs + (long long)b * a.sx, (unsigned)((long long)(a.N2 + 1) * a.ldx * 8));
    const p2pfm_rsrc_t ry = p2pfm_rsrc(a.Phi1 + (long long)b * a.s1, (unsigned)((long long)a.N1 * a.ld1 * sizeof(TR)));
    const int32_t* __restrict__ idx = a.p21 + (long long)b * a.N2;
    int* ip = reinterpret_cast<int*>(p2pfm_sm) + wave * a.rps;      // this wave's gather indices, clamped (an index load inside the
                                                                    // loop shares the counter of the operand loads: the compiler drained
                                                                    // the whole queue for it at the top of every iteration)
    // byte offsets of this lane's first column inside a row; the other blocks sit at compile-time distances (16 columns).
    // Rows of Xs are ldx >= 16 nrb wide; a column of Phi1 beyond its row feeds only result entries that are not stored, a read
    // beyond the array returns 0.
    const int cax = (mb0 * 16 + l15) * 8, cby = (cb0 * 16 + l15) * (int)sizeof(TR);
    const int xstep = a.ldx * 8, ystep = a.ld1 * (int)sizeof(TR);

    f64x4 acc[RBW][CBW];
#pragma unroll
    for (int r = 0; r < RBW; ++r)
#pragma unroll
        for (int c = 0; c < CBW; ++c) acc[r][c] = f64x4{0.0, 0.0, 0.0, 0.0};

    const int nb = wave * a.rps, ne = min(a.N2, nb + a.rps);
    int nks = ne > nb ? (ne - nb + 3) >> 2 : 0;
    if (a.dbg & 64) nks = min(nks, 3);
    const int tstep = (a.dbg & 128) ? 0 : 4;
    // k-step t covers vertices n = nb + 4 t + g.  Per step and lane: the gather index of its vertex (requested three steps
    // ahead), one entry of each A block and of each B block (two steps ahead).  A vertex beyond the slice reads the zero row
    // of Xs and the gather index of the last vertex: it adds exact zeros.
    double fa0[RBW], fa1[RBW], fa2[RBW];
    TR fb0[CBW], fb1[CBW], fb2[CBW];
    for (int q = lane; q < a.rps; q += 64) ip[q] = min(max(idx[min(nb + q, a.N2 - 1)], 0), a.N1 - 1);
    __builtin_amdgcn_wave_barrier();
    int pr0 = 0, pr1 = 0, pr2 = 0;                // gather indices of the steps = 0 / 1 / 2 mod 3
#define PF_IDX(t_, p_) { p_ = ip[min(tstep * (t_) + g, a.rps - 1)]; }
#define PF_LOAD(t_, p_, fa_, fb_)                                                  \
    {                                                                              \
        const int n_ = nb + tstep * (t_) + g;                                      \
        const int xo_ = ((n_ < ne) ? n_ : a.N2) * xstep + cax;                     \
        const int yo_ = p_ * ystep + cby;                                          \
        _Pragma("unroll") for (int r = 0; r < RBW; ++r) fa_[r] = p2pfm_bload<double>(rx, xo_ + r * 128); \
        _Pragma("unroll") for (int c = 0; c < CBW; ++c) fb_[c] = p2pfm_bload<TR>(ry, yo_ + c * 16 * (int)sizeof(TR)); \
    }
#define PF_MMA(fa_, fb_)                                                           \
    {                                                                              \
        _Pragma("unroll") for (int c = 0; c < CBW; ++c)                            \
            _Pragma("unroll") for (int r = 0; r < RBW; ++r)                        \
                acc[r][c] = mfma_f64_16x16x4(fa_[r], (double)fb_[c], acc[r][c]);   \
    }
    // The loop body has no branch, so the compiler can count the loads in flight (with a conditional fetch it waited vmcnt(0)
    // -- for the operands it had only just requested -- in front of every group of matrix instructions).  Per third of an
    // iteration: fetch the operands two steps ahead (their gather index was requested three steps before them), then the matrix
    // instructions of the current step.  A compiler-level memory barrier (an empty asm that clobbers memory) + a scheduling
    // barrier follow every fetch: without the first the loads of read-only memory were sunk across the loop's back edge to their
    // use, without the second the matrix instructions were hoisted in front of the fetch.
#define PF_FENCE() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
    if (nks > 0) {
        PF_IDX(0, pr0)
        PF_IDX(1, pr1)
        PF_IDX(2, pr2)
        PF_LOAD(0, pr0, fa0, fb0)
        PF_IDX(3, pr0)
        PF_FENCE()                                // (the order of the loads in flight at the loop's entry = the order inside the loop:
        PF_LOAD(1, pr1, fa1, fb1)                 //  the counted waits of the loop then allow two sets in flight)
        PF_IDX(4, pr1)
        PF_FENCE()
        const int nks3 = (nks + 2) / 3 * 3;
        for (int t = 0; t < nks3; t += 3) {
            PF_LOAD(t + 2, pr2, fa2, fb2)
            PF_IDX(t + 5, pr2)
            PF_FENCE()
            PF_MMA(fa0, fb0)
            PF_LOAD(t + 3, pr0, fa0, fb0)
            PF_IDX(t + 6, pr0)
            PF_FENCE()
            PF_MMA(fa1, fb1)
            PF_LOAD(t + 4, pr1, fa1, fb1)
            PF_IDX(t + 7, pr1)
            PF_FENCE()
            PF_MMA(fa2, fb2)
        }
    }
#undef PF_FENCE
#undef PF_MMA
#undef PF_LOAD
#undef PF_IDX
    if (a.dbg & 32) { double z = 0.0; _Pragma("unroll") for (int r = 0; r < RBW; ++r) _Pragma("unroll") for (int c = 0; c < CBW; ++c) z += acc[r][c][0] + acc[r][c][3]; if (z == 1.2345) a.C[0] = z; return; }
    // the S slices of every block, added in ascending slice order by wave (block number % S)
    __syncthreads();                                          // (the reduction buffers overlay the index slices)
    double* Cb = a.C + (long long)b * a.strideC;
#pragma unroll
    for (int r = 0; r < RBW; ++r)
#pragma unroll
        for (int c = 0; c < CBW; ++c) {
            const int blk = r * CBW + c;
            double* buf = p2pfm_sm + (blk & 1) * (S * 256);
            *reinterpret_cast<f64x4*>(buf + (wave * 64 + lane) * 4) = acc[r][c];
            __syncthreads();
            if (wave == blk % S) {
                f64x4 s = *reinterpret_cast<const f64x4*>(buf + lane * 4);
#pragma unroll
                for (int w = 1; w < S; ++w) {
                    const f64x4 o = *reinterpret_cast<const f64x4*>(buf + (w * 64 + lane) * 4);
                    s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
                }
                const int cc = (cb0 + c) * 16 + l15;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = (mb0 + r) * 16 + g + 4 * q;
                    if (m < a.k2 && cc < a.k1) Cb[(long long)m * a.ldc + cc] = s[q];
                }
            }
        }
}

size_t dm_p2pfm_ws_bytes(int B, int N2, int k1, int k2) {
    const int nsplit = fm_split(B, N2, k1, k2);
    return (nsplit > 1 ? dm_align_up((size_t)nsplit * B * k2 * k1 * 8) + 4096 : 4096) + dm_p2pfm_xs_bytes(B, N2, k2) + 4096;
}

template <typename TR, int RBW, int CBW, int S>
static int p2pfm_launch(dm_ctx* ctx, const p2pfm_args<TR>& a, int grid) {
    const size_t red = (size_t)2 * S * 256 * 8, ind = (size_t)S * a.rps * 4;
    DM_LAUNCH(ctx, "p2pfm_tn_f64", (p2pfm_direct_kernel<TR, RBW, CBW, S>), dim3(grid), dim3(64 * S), red > ind ? red : ind, a);
    return DM_OK;
}
// Xs: mass2 * Phi2 as dm_p2pfm_prescale builds it for ldx / 16 blocks of columns (>= ceil(k2 / 16)); null: built here
template <typename TR>
static int launch_p2pfm_direct(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21, const TR* Phi1,
                               int ld1, const TR* Phi2, int ld2, const double* mass2, double* C, int ldc, long long strideC,
                               const double* Xs, int ldx) {
    if (!Xs) {
        double* buf = (double*)dm_ws_take(ctx, dm_p2pfm_xs_bytes(B, N2, k2));
        if (!buf) return dm_fail(ctx, DM_ENOMEM, "p2p_to_fm: workspace not reserved");
        int rc = dm_p2pfm_prescale<TR>(ctx, B, N2, k2, Phi2, ld2, mass2, buf);
        if (rc) return rc;
        Xs = buf; ldx = pad_to16(k2);
    }
    const int nrb = dm_cdiv(k2, 16), ncb = dm_cdiv(k1, 16);
    // The K-slicing fixes the summation order of every entry: it depends on the sizes only, never on the batch.
    // Small maps (at most 7 x 7 blocks, k <= 112): 8 slices (more, lighter waves), else 4.
    const bool small = nrb <= 7 && ncb <= 7;
    const int knob = dm_knob("DM_P2PFM_SHAPE", 0);         // experiments: 1 = 8 slices for every size, 8 = 4 slices for every size, 16 = the r04 tile choice
    const int S = ((small && !(knob & 8)) || (knob & 1)) ? 8 : 4;
    // The tile of a workgroup (R x C blocks per wave) only decides who computes an entry, not how: it is chosen for the batch.
    // A workgroup is one quantum of N R C / 16 matrix instructions per SIMD, the pairs of an XCD (b % 8) share its 32 CUs, a CU
    // that holds two workgroups takes twice as long: cost = ceil(tiles x pairs on the XCD / 32) (R C + 0.3 (R + C) + 0.75).  (r04 took 2 x 7-wide tiles for every large map: 10 / 12 / 14 tiles per pair,
    // i.e. 1.25 - 1.75 workgroups per CU at B = 32 -- two rounds where 3 x 5, 5 x 3 or 6 x 3 tiles need one.)
    static const int shapes4[][2] = {{1, 1}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6}, {1, 7}, {2, 1}, {2, 2}, {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7},
                                     {3, 3}, {3, 4}, {3, 5}, {4, 3}, {4, 4}, {5, 3}, {6, 3}};
    static const int shapes8[][2] = {{1, 1}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6}, {1, 7}};
    const int (*shapes)[2] = S == 8 ? shapes8 : shapes4;
    const int nshapes = S == 8 ? (int)(sizeof(shapes8) / sizeof(shapes8[0])) : (int)(sizeof(shapes4) / sizeof(shapes4[0]));
    int R = 0, Cw = 0;
    if (knob >> 8) {                                       // experiments: forced shape, (R << 4 | C) << 8
        R = (knob >> 12) & 15; Cw = (knob >> 8) & 15;
        if (R > nrb) R = nrb;
        if (Cw > ncb) Cw = ncb;
    } else if (knob & 16) {
        const bool two = !small;
        R = two ? 2 : 1;
        Cw = dm_cdiv(ncb, dm_cdiv(ncb, 7));
        if (S == 8) R = 1;
    } else {
        long long best = -1;
        const int ppx = dm_cdiv(B, 8);                     // pairs on the fullest XCD
        for (int q = 0; q < nshapes; ++q) {
            const int r = shapes[q][0], c = shapes[q][1];
            if (r > nrb || c > ncb) continue;
            const int T = dm_cdiv(nrb, r) * dm_cdiv(ncb, c);
            // per round and workgroup: R C matrix instructions per k-step, R + C operand loads (0.3 of a matrix instruction
            // each, measured on the 1 x 7 shape), a fixed part (index slices, reduction)
            const long long cost = (long long)dm_cdiv(T * ppx, 32) * (100 * r * c + 30 * (r + c) + 75);
            const long long key = (cost << 12) + T;
            if (best < 0 || key < best) { best = key; R = r; Cw = c; }
        }
    }
    p2pfm_args<TR> a;
    a.Phi1 = Phi1; a.s1 = (long long)N1 * ld1; a.ld1 = ld1; a.N1 = N1;
    a.Xs = Xs; a.sx = (long long)(N2 + 1) * ldx; a.ldx = ldx; a.N2 = N2;
    a.p21 = p21; a.k1 = k1; a.k2 = k2; a.C = C; a.ldc = ldc; a.strideC = strideC; a.B = B;
    a.dbg = dm_knob("DM_ZO_DEBUG", 0);
    a.TM = dm_cdiv(nrb, R);
    a.TC = dm_cdiv(ncb, Cw);
    a.rps = dm_cdiv(dm_cdiv(N2, S), 4) * 4;
    const int grid = a.TM * a.TC * dm_cdiv(B, 8) * 8;
#define P2PFM_CASE(R_, C_, S_) if (R == R_ && Cw == C_ && S == S_) return p2pfm_launch<TR, R_, C_, S_>(ctx, a, grid);
    P2PFM_CASE(1, 1, 4) P2PFM_CASE(1, 2, 4) P2PFM_CASE(1, 3, 4) P2PFM_CASE(1, 4, 4) P2PFM_CASE(1, 5, 4) P2PFM_CASE(1, 6, 4) P2PFM_CASE(1, 7, 4)
    P2PFM_CASE(2, 1, 4) P2PFM_CASE(2, 2, 4) P2PFM_CASE(2, 3, 4) P2PFM_CASE(2, 4, 4) P2PFM_CASE(2, 5, 4) P2PFM_CASE(2, 6, 4) P2PFM_CASE(2, 7, 4)
    P2PFM_CASE(3, 3, 4) P2PFM_CASE(3, 4, 4) P2PFM_CASE(3, 5, 4) P2PFM_CASE(4, 3, 4) P2PFM_CASE(4, 4, 4) P2PFM_CASE(5, 3, 4) P2PFM_CASE(6, 3, 4)
    P2PFM_CASE(1, 1, 8) P2PFM_CASE(1, 2, 8) P2PFM_CASE(1, 3, 8) P2PFM_CASE(1, 4, 8) P2PFM_CASE(1, 5, 8) P2PFM_CASE(1, 6, 8) P2PFM_CASE(1, 7, 8)
#undef P2PFM_CASE
    return dm_fail(ctx, DM_EINVAL, "p2p_to_fm: no tile shape %d x %d x %d", R, Cw, S);
}

template <typename TR>
int dm_launch_p2p_to_fm(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21, const TR* Phi1,
                        int ld1, const TR* Phi2, int ld2, const double* mass2, double* C, int ldc,
                        long long strideC, const double* Xs, int ldx) {
    // (arrays of 4 GiB or more per pair would not fit the 32-bit offsets of the direct kernel)
    if (ctx->opt_p2pfm_direct && (long long)N1 * ld1 * 8 < (1LL << 31) && (long long)(N2 + 1) * pad_to16(k2) * 8 < (1LL << 31) && N2 <= 15000)
        return launch_p2pfm_direct<TR>(ctx, B, N1, N2, k1, k2, p21, Phi1, ld1, Phi2, ld2, mass2, C, ldc, strideC, Xs, ldx);
    const int nsplit = fm_split(B, N2, k1, k2);
    const int kchunk = FM_KCHUNK;
    double* partial = nullptr;
    if (nsplit > 1) {
        partial = (double*)dm_ws_take(ctx, (size_t)nsplit * B * k2 * k1 * 8);
        if (!partial) return dm_fail(ctx, DM_ENOMEM, "p2p_to_fm: workspace not reserved");
    }
    RowsScaled<TR, double> opx{Phi2, (long long)N2 * ld2, ld2, k2, nullptr, 0};
    RowsGatherScaled<TR, double> opy{Phi1, (long long)N1 * ld1, ld1, k1, p21, (long long)N2, N1, mass2, (long long)N2};
    OutFM out{nsplit > 1 ? nullptr : C, ldc, strideC, partial, B, k2, k1};
    dim3 grid(dm_cdiv(k2, TN_T) * dm_cdiv(k1, TN_T), nsplit, B);
    DM_LAUNCH(ctx, "p2pfm_tn_f64", (gemm_tn_f64<RowsScaled<TR, double>, RowsGatherScaled<TR, double>, OutFM>), grid, dim3(256), 0, opx,
              opy, out, k2, k1, N2, kchunk);
    if (nsplit > 1) {
        const long long n = (long long)B * k2 * k1;
        DM_LAUNCH(ctx, "splitk_reduce", splitk_reduce_fm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, partial,
                  nsplit, B, k2, k1, C, ldc, strideC);
    }
    return DM_OK;
}
template int dm_launch_p2p_to_fm<float>(dm_ctx*, int, int, int, int, int, const int32_t*, const float*, int, const float*, int,
                                        const double*, double*, int, long long, const double*, int);
template int dm_launch_p2p_to_fm<double>(dm_ctx*, int, int, int, int, int, const int32_t*, const double*, int, const double*, int,
                                         const double*, double*, int, long long, const double*, int);

template <typename TR>
static int p2p_to_fm_impl(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21, const TR* Phi1,
                          int ld1, const TR* Phi2, int ld2, const TR* mass2_in, double* C) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N1 > 0 && N2 > 0 && k1 > 0 && k2 > 0, "sizes must be positive");
    DM_REQUIRE(ctx, p21 && Phi1 && Phi2 && mass2_in && C, "null pointer");
    DM_REQUIRE(ctx, ld1 >= k1 && ld2 >= k2, "eigenvector row stride smaller than the map size");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = dm_ws_reserve(ctx, dm_p2pfm_ws_bytes(B, N2, k1, k2) + dm_align_up((size_t)B * N2 * 8));
    if (rc) return rc;
    const double* mass2 = nullptr;
    rc = dm_widen_mass(ctx, (long long)B * N2, mass2_in, (double*)dm_ws_take(ctx, (size_t)B * N2 * 8), &mass2);
    if (rc) return rc;
    return dm_launch_p2p_to_fm<TR>(ctx, B, N1, N2, k1, k2, p21, Phi1, ld1, Phi2, ld2, mass2, C, k1, (long long)k2 * k1);
}
extern "C" int dm_p2p_to_fm(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21, const float* Phi1,
                            int ld1, const float* Phi2, int ld2, const float* mass2, double* C) {
    return p2p_to_fm_impl<float>(ctx, B, N1, N2, k1, k2, p21, Phi1, ld1, Phi2, ld2, mass2, C);
}
extern "C" int dm_p2p_to_fm_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21, const double* Phi1,
                                int ld1, const double* Phi2, int ld2, const double* mass2, double* C) {
    return p2p_to_fm_impl<double>(ctx, B, N1, N2, k1, k2, p21, Phi1, ld1, Phi2, ld2, mass2, C);
}

static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

// ---- the fused iteration (dm_zoomfuse.hip): 5 launches per iteration, no memset, no K-major copies -------------------------
// Eligible: both meshes have at least one 256-row tile and the final map fits the embedding kernel's accumulators (k <= 208);
// anything else runs the six-launch loop below (also dm_set_option "zoomout_fused" = 0: the tests compare the two).
static bool zoomout_fused_ok(const dm_ctx* ctx, int N1, int N2, int kf) {
    return ctx->opt_zoomout_fused && ctx->opt_knn_split && ctx->opt_simnn_pipe && N1 >= 256 && N2 >= 256 && kf <= 208;
}
static inline int zo_depth(int k) { const int d = 32 * ((k + 15) / 16); return d < 160 ? 160 : d; }   // halves per split row (tile kernel: >= 5 stages)
constexpr int ZO_NCH = 32;

template <typename TR>
static int zoomout_fused(dm_ctx* ctx, int B, int N1, int N2, int k0, int nit, int step, const TR* Phi1, int ld1,
                         const TR* Phi2, int ld2, const TR* mass2_in, const double* C0, double* Cout, int32_t* p21_out) {
    const int kf = k0 + nit * step;
    const int Kpad = pad_to(kf, 16), R1 = pad_to(N1, 256), R2 = pad_to(N2, 256), N1pad = pad_to(N1, 128);
    const int ldT = zo_depth(kf), ldS = ldT;
    const size_t bytes_C = (size_t)B * Kpad * Kpad * 8;
    const size_t bytes_Fx = (size_t)B * R2 * ldT * 2, bytes_Fy = (size_t)B * R1 * ldS * 2;
    const size_t ctl1 = dm_simnn_ctl_bytes(B), amax1 = dm_align_up((size_t)B * 8), bmax1 = dm_align_up((size_t)B * 4);
    const size_t bytes_zero = (size_t)(nit + 3) * amax1 + (size_t)(nit + 2) * (bmax1 + dm_align_up(ctl1)) + dm_align_up((size_t)(nit + 2) * 4);
    const size_t need = dm_align_up((size_t)B * N2 * 8) + dm_align_up(bytes_Fx) + dm_align_up(bytes_Fy) + 2 * dm_align_up(bytes_C) +
                        dm_align_up((size_t)B * R1 * 4) + dm_align_up((size_t)B * N1pad * 8) + dm_align_up((size_t)B * N1 * Kpad * 8) +
                        3 * dm_align_up((size_t)B * N2 * 4) + dm_align_up((size_t)B * ZO_NCH * 8) + dm_align_up(bytes_zero) +
                        dm_simnn_ws_bytes(B, N2, N1, 0) + dm_p2pfm_ws_bytes(B, N2, kf, kf) + 65536;
    // (dm_p2pfm_ws_bytes = the prescaled basis Xs AND the split-K partials of the staged p2p_to_FM, which dm_launch_p2p_to_fm falls back
    //  to beyond 15000 target vertices or with p2pfm_direct = 0: ADVICE r04 -- with the Xs share alone that fallback ran out of workspace)
    int rc = dm_ws_reserve(ctx, need);
    if (rc) return rc;
    const double* mass2 = nullptr;
    rc = dm_widen_mass(ctx, (long long)B * N2, mass2_in, (double*)dm_ws_take(ctx, (size_t)B * N2 * 8), &mass2);
    if (rc) return rc;
    _Float16* Fx = (_Float16*)dm_ws_take(ctx, bytes_Fx);
    _Float16* Fy = (_Float16*)dm_ws_take(ctx, bytes_Fy);
    double* Ca = (double*)dm_ws_take(ctx, bytes_C);
    double* Cb = (double*)dm_ws_take(ctx, bytes_C);
    float* bias = (float*)dm_ws_take(ctx, (size_t)B * R1 * 4);
    double* n1 = (double*)dm_ws_take(ctx, (size_t)B * N1pad * 8);
    double* embr = (double*)dm_ws_take(ctx, (size_t)B * N1 * Kpad * 8);
    int32_t* p21 = (int32_t*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    int* qrow = (int*)dm_ws_take(ctx, (size_t)B * N2 * 4);                // queue of the rows an iteration re-evaluates exactly
    float* qthr = (float*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    double* amaxT = (double*)dm_ws_take(ctx, (size_t)B * ZO_NCH * 8);
    char* zero = (char*)dm_ws_take(ctx, bytes_zero);
    if (!Fx || !Fy || !Ca || !Cb || !bias || !n1 || !embr || !p21 || !qrow || !qthr || !amaxT || !zero)
        return dm_fail(ctx, DM_ENOMEM, "zoomout: workspace not reserved");
    char* amax_slots = zero;                                             // (nit + 3) x B maxima of |emb1| (float64 bits)
    char* bmax_slots = zero + (size_t)(nit + 3) * amax1;                 // (nit + 2) x B maxima of |bias| (fp32 bits)
    char* ctl_slots = bmax_slots + (size_t)(nit + 2) * bmax1;            // (nit + 2) control blocks of the tile pass
    unsigned int* qcount_slots = (unsigned int*)(ctl_slots + (size_t)(nit + 2) * dm_align_up(ctl1));   // (nit + 2) queue lengths
    // everything an iteration accumulates into with atomicMax, for all iterations: ONE memset per call
    DM_CHECK_HIP(ctx, hipMemsetAsync(zero, 0, bytes_zero, ctx->stream));
    DM_CHECK_HIP(ctx, hipMemsetAsync(Ca, 0, bytes_C, ctx->stream));      // maps grow inside zeroed Kpad x Kpad frames
    DM_CHECK_HIP(ctx, hipMemsetAsync(Cb, 0, bytes_C, ctx->stream));
    if (R2 != N2) DM_CHECK_HIP(ctx, hipMemsetAsync(Fx, 0, bytes_Fx, ctx->stream));
    if (R1 != N1) {
        DM_CHECK_HIP(ctx, hipMemsetAsync(Fy, 0, bytes_Fy, ctx->stream));
        DM_CHECK_HIP(ctx, hipMemsetAsync(bias, 0, (size_t)B * R1 * 4, ctx->stream));
    }
    // target side, once: scale and split rows of Phi2 for the full depth (a search at depth k reads the first 32 ceil(k / 16)
    // halves; the source rows are zero beyond k)
    rc = dm_zo_absmax_rows<TR>(ctx, B, N2, kf, Phi2, ld2, ZO_NCH, amaxT);
    if (rc) return rc;
    rc = dm_fm_split_build_rows<TR>(ctx, B, N2, kf, Phi2, ld2, amaxT, ZO_NCH, ldT, Fx, R2);
    if (rc) return rc;
    rc = dm_zo_copy_mat(ctx, B, k0, k0, C0, k0, (long long)k0 * k0, Ca, Kpad, (long long)Kpad * Kpad);
    if (rc) return rc;
    // mass2 * Phi2 for p2p_to_FM, once for the full width (a smaller map reads its first blocks of columns)
    double* Xs = (double*)dm_ws_take(ctx, dm_p2pfm_xs_bytes(B, N2, kf));
    if (!Xs) return dm_fail(ctx, DM_ENOMEM, "zoomout: workspace not reserved");
    rc = dm_p2pfm_prescale<TR>(ctx, B, N2, kf, Phi2, ld2, mass2, Xs);
    if (rc) return rc;
    const size_t ws_mark = ctx->ws_off;

    double* cur = Ca;
    double* nxt = Cb;
    int k = k0;
    for (int it = 0; it <= nit; ++it) {
        const bool last = (it == nit);
        if (last && !p21_out) break;
        ctx->ws_off = ws_mark;
        zo_embed_args<TR> ea;
        ea.Phi1 = Phi1; ea.s1 = (long long)N1 * ld1; ea.ld1 = ld1; ea.N1 = N1;
        ea.C = cur; ea.strideC = (long long)Kpad * Kpad; ea.ldc = Kpad;
        ea.k = k; ea.nrb = dm_cdiv(k, 16); ea.D = zo_depth(k); ea.ldS = ldS; ea.R1 = R1;
        ea.Fy = Fy; ea.bias = bias; ea.n1 = n1; ea.N1pad = N1pad; ea.embr = embr; ea.Kpad = Kpad;
        ea.amaxT = amaxT; ea.nT = ZO_NCH; ea.dbg = dm_knob("DM_ZO_DEBUG", 0);
        ea.bmax = (unsigned int*)(bmax_slots + (size_t)it * bmax1);
        if (it == 0) {                                                 // the first scale: a pass that only takes the maximum
            ea.amax_prev = (const unsigned long long*)amax_slots;
            ea.amax_cur = (unsigned long long*)amax_slots;
            ea.only_max = 1;
            rc = dm_zo_embed_split<TR>(ctx, B, ea);
            if (rc) return rc;
        }
        ea.amax_prev = (const unsigned long long*)(amax_slots + (size_t)it * amax1);
        ea.amax_cur = (unsigned long long*)(amax_slots + (size_t)(it + 1) * amax1);
        ea.only_max = 0;
        rc = dm_zo_embed_split<TR>(ctx, B, ea);
        if (rc) return rc;
        // the search: one biased key on split rows; dropped <xl, yl> and the two residuals 3 * 2^-22, the fp16 subnormal
        // floor 2 sqrt(k) 2^-25 budgeted four times (the scale may be up to 4x below the ideal one), 25 % slack
        const float rel_extra = 1.25f * (3.0f * 2.3841858e-7f + 4.0f * 2.0f * sqrtf((float)k) * 2.9802322e-8f);
        dm_simnn_dual dual{};
        dual.bias = bias; dual.tau_add = (const float*)ea.bmax; dual.padded = true; dual.single = true;
        dm_simnn_ext ext;
        ext.ctl = ctl_slots + (size_t)it * dm_align_up(ctl1); ext.skip_merge = true;
        dm_simnn_queue qd;
        rc = dm_simnn_core(ctx, B, N2, N1, ea.D, Fx, ldT, Fy, ldS, rel_extra, nullptr, nullptr, nullptr, nullptr, &qd, &dual, &ext);
        if (rc) return rc;
        zo_mx_args<TR> ma;
        ma.q = dm_simnn_queue{ext.pb, ext.pj, ext.ps, ext.nparts, ext.pw, ext.N2pad, nullptr, nullptr, nullptr};
        ma.tnorm2 = ext.tnorm2; ma.smax2 = ext.smax2; ma.bmax = ea.bmax; ma.tau_scale = ext.tau_scale;
        ma.amax_prev = ea.amax_prev; ma.amax_cur = ea.amax_cur;
        ma.Phi2 = Phi2; ma.ld2 = ld2; ma.embr = embr; ma.Kpad = Kpad; ma.n1 = n1; ma.N1pad = N1pad;
        ma.K = k; ma.N2 = N2; ma.N1 = N1; ma.nn = last ? p21_out : p21;
        ma.qrow = qrow; ma.qthr = qthr; ma.qcount = qcount_slots + it; ma.qcap = B * N2;
        ma.dbg = ((ea.dbg & 16) ? 1 : 0) | ((ea.dbg & 256) ? 2 : 0) | ((ea.dbg & 512) ? 4 : 0);
        rc = dm_zo_merge_exact<TR>(ctx, B, ma);
        if (rc) return rc;
        if (last) break;
        const int kn = k + step;
        rc = dm_launch_p2p_to_fm<TR>(ctx, B, N1, N2, kn, kn, p21, Phi1, ld1, Phi2, ld2, mass2, nxt, Kpad, (long long)Kpad * Kpad, Xs, Kpad);
        if (rc) return rc;
        double* tmp = cur; cur = nxt; nxt = tmp;
        k = kn;
    }
    return dm_zo_copy_mat(ctx, B, kf, kf, cur, Kpad, (long long)Kpad * Kpad, Cout, kf, (long long)kf * kf);
}

template <typename TR>
static int zoomout_impl(dm_ctx* ctx, int B, int N1, int N2, int k0, int nit, int step, const TR* Phi1, int ld1,
                        const TR* Phi2, int ld2, const TR* mass2_in, const double* C0, double* Cout,
                        int32_t* p21_out) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N1 > 0 && N2 > 0 && k0 > 0 && nit >= 0 && step > 0, "sizes must be positive");
    DM_REQUIRE(ctx, Phi1 && Phi2 && mass2_in && C0 && Cout, "null pointer");
    const int kf = k0 + nit * step;
    DM_REQUIRE(ctx, ld1 >= kf && ld2 >= kf, "not enough eigenvectors for k0 + nit*step (zoomout.py:87-92)");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (zoomout_fused_ok(ctx, N1, N2, kf))
        return zoomout_fused<TR>(ctx, B, N1, N2, k0, nit, step, Phi1, ld1, Phi2, ld2, mass2_in, C0, Cout, p21_out);

    const int N1pad = pad_to(N1, 128), N2pad = pad_to(N2, 128), Kpad = pad_to(kf, 16);
    const size_t bytes_AT = (size_t)B * Kpad * N2pad * 8, bytes_BT = (size_t)B * Kpad * N1pad * 8;
    const size_t bytes_C = (size_t)B * kf * kf * 8;
    const size_t need = dm_align_up(bytes_AT) + dm_align_up(bytes_BT) + 2 * dm_align_up(bytes_C) +
                        dm_align_up((size_t)B * N1pad * 8) + dm_align_up((size_t)B * N2 * 4) +
                        dm_gred_ws_bytes(B, N2, N1) + dm_knn_split_prep_bytes(B, N2, kf) + dm_knn_split_ws_bytes(B, N2, N1, kf) +
                        dm_align_up((size_t)B * (N1pad / DM_EMB_COLS + 1) * 8) + dm_p2pfm_ws_bytes(B, N2, kf, kf) +
                        dm_align_up((size_t)B * N2 * 8);
    int rc = dm_ws_reserve(ctx, need);
    if (rc) return rc;
    const double* mass2 = nullptr;
    rc = dm_widen_mass(ctx, (long long)B * N2, mass2_in, (double*)dm_ws_take(ctx, (size_t)B * N2 * 8), &mass2);
    if (rc) return rc;
    double* AT = (double*)dm_ws_take(ctx, bytes_AT);
    double* BT = (double*)dm_ws_take(ctx, bytes_BT);
    double* Ca = (double*)dm_ws_take(ctx, bytes_C);
    double* Cb = (double*)dm_ws_take(ctx, bytes_C);
    double* n1 = (double*)dm_ws_take(ctx, (size_t)B * N1pad * 8);
    int32_t* p21 = (int32_t*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    double* amaxS = (double*)dm_ws_take(ctx, (size_t)B * (N1pad / DM_EMB_COLS + 1) * 8);   // max |emb1| per block of columns (dm_launch_embed)
    if (!AT || !BT || !Ca || !Cb || !n1 || !p21 || !amaxS) return dm_fail(ctx, DM_ENOMEM, "zoomout: workspace not reserved");

    // Phi2^T for all kf columns, once.  Row c of AT only enters G when c < current k because the
    // matching row of BT (emb1^T) is zero beyond the current map size.
    rc = dm_launch_phiT<TR>(ctx, B, N2, kf, Phi2, ld2, AT, Kpad, N2pad);
    if (rc) return rc;
    // target side of the nearest-neighbour search (fp16 split of Phi2, all kf columns), once for the whole call
    dm_knn_split_state knn;
    rc = dm_knn_split_prepare(ctx, B, N2, N2pad, Kpad, kf, AT, &knn);
    if (rc) return rc;
    const size_t ws_mark = ctx->ws_off;
    DM_CHECK_HIP(ctx, hipMemcpyAsync(Ca, C0, (size_t)B * k0 * k0 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    // emb1^T buffer: rows >= current k and columns >= N1 must read as zero; rows only ever grow, so one clear suffices
    DM_CHECK_HIP(ctx, hipMemsetAsync(BT, 0, bytes_BT, ctx->stream));

    double* cur = Ca;
    double* nxt = Cb;
    int k = k0;
    for (int it = 0; it <= nit; ++it) {
        const bool last = (it == nit);
        if (last && !p21_out) break;
        ctx->ws_off = ws_mark;
        // emb1 = Phi1[:, :k] C^T  ->  BT (rows >= k zero), n1
        rc = dm_launch_embed<TR>(ctx, B, N1, k, k, Phi1, ld1, cur, k, (long long)k * k, 0, BT, Kpad, N1pad, n1, 0, amaxS);
        if (rc) return rc;
        dm_gred_args a;
        a.B = B; a.N2 = N2; a.N1 = N1; a.Kloop = pad_to(k, 16); a.Ktrue = k;
        a.AT = AT; a.N2pad = N2pad; a.BT = BT; a.N1pad = N1pad; a.Kpad = Kpad;
        a.n1 = n1; a.n2 = nullptr; a.mass1 = nullptr;
        a.knn21 = last ? p21_out : p21; a.knn12 = nullptr; a.ind21 = nullptr; a.ind12 = nullptr;
        rc = dm_launch_knn21(ctx, a, knn, amaxS, dm_cdiv(N1pad, DM_EMB_COLS));
        if (rc) return rc;
        if (last) break;
        const int kn = k + step;
        rc = dm_launch_p2p_to_fm<TR>(ctx, B, N1, N2, kn, kn, p21, Phi1, ld1, Phi2, ld2, mass2, nxt, kn, (long long)kn * kn);
        if (rc) return rc;
        double* tmp = cur; cur = nxt; nxt = tmp;
        k = kn;
    }
    DM_CHECK_HIP(ctx, hipMemcpyAsync(Cout, cur, (size_t)B * kf * kf * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return DM_OK;
}
extern "C" int dm_zoomout(dm_ctx* ctx, int B, int N1, int N2, int k0, int nit, int step, const float* Phi1, int ld1,
                          const float* Phi2, int ld2, const float* mass2, const double* C0, double* Cout,
                          int32_t* p21_out) {
    return zoomout_impl<float>(ctx, B, N1, N2, k0, nit, step, Phi1, ld1, Phi2, ld2, mass2, C0, Cout, p21_out);
}
extern "C" int dm_zoomout_f64(dm_ctx* ctx, int B, int N1, int N2, int k0, int nit, int step, const double* Phi1, int ld1,
                              const double* Phi2, int ld2, const double* mass2, const double* C0, double* Cout,
                              int32_t* p21_out) {
    return zoomout_impl<double>(ctx, B, N1, N2, k0, nit, step, Phi1, ld1, Phi2, ld2, mass2, C0, Cout, p21_out);
}

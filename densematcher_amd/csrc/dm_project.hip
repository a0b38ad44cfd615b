// Spectral projection on the fp16 matrix cores (dm_project fast path for fp16 descriptors).
//
//   Ared[b] = Phi[b][:, :k]^T (mass[b] * F[b])        pyFM/optimize/base_functions.py:526-532, trimesh.py:533-556
//
// F is fp16 already (example.ipynb cells 2/5/11: the network runs under autocast), so it enters the MFMA
// exactly.  The other operand X[n][m] = mass[n] * Phi[n][m] * 2^e (fp32) is split on the fly into two fp16
// pieces hi + lo (22 significant bits, e chosen so that max |X| ~ 2^14), and the contraction over the N
// vertices runs as two v_mfma_f32_32x32x16_f16 per tile step into one fp32 accumulator:
//   error per term <= 2^-22 (split) + 2^-24 (mass product), fp32 accumulation over N / nsplit terms;
// the split-K partial sums are combined in float64.  Net: relative error ~1e-6 worst case (3e-7 typical) of the
// projected descriptor, i.e. the same class as the reference's own fp32 torch projection.
// The float64 MFMA path (dm_fmap.hip) stays available through the DM_PROJECT_F64 flag.
//
// Both operands are K-major in memory (the contraction index n is the row index), which is the natural layout
// for ds_read_b64_tr_b16: tiles are staged row-major [n][m] / [n][d] and the MFMA fragments (8 consecutive n
// per lane) come out of the hardware transpose read.
#include "dm_device.h"
#include "dm_internal.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr int PT = 128;      // output tile: PT basis functions x PTD descriptor channels.  The fp32 -> hi/lo split of the
constexpr int PTD = 192;     // basis slab is VALU work repeated by every d-tile of a (pair, chunk): wide d-tiles amortise it;
                             // 192 (not 256): D = 768 gives 4 d-tiles x 2 chunks x 64 pairs = 512 workgroups = every slot of
                             // the chip (two per CU) in ONE round; 256-wide tiles filled 384 of the 512 slots
constexpr int PNT = PTD / 64;   // 32-channel accumulator blocks per wave (the two waves along d own PTD / 2 channels each)
constexpr int PFV = PTD / 64;   // 16-byte vectors of descriptor channels a staging thread moves (PTD / 8 channels)
constexpr int PBK = 32;      // vertices per stage
constexpr int PLD = 160;     // LDS row stride in halves (320 B): rows land 16 banks apart -> conflict-free tr reads
constexpr int PLDF = 224;    // same property for the 192-wide descriptor rows (448 B = 112 dwords = 48 mod 64)
constexpr int PSTAGE = 2 * PBK * PLD + PBK * PLDF;   // halves per stage buffer: Xhi | Xlo | F

template <int LD>
__device__ __forceinline__ f16x8 tr_frag(const _Float16* base, int row0, int col, int lane) {
    // 16-lane group: lane t supplies the address of row (row0 + (t >> 2)), columns col + 4 (t & 3) .. +3 and
    // receives column (col + t), rows row0 .. row0 + 3.  Two reads give 8 consecutive k.
    const int t = lane & 15;
    const _Float16* p = base + (row0 + (t >> 2)) * LD + col + 4 * (t & 3);
    const h4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)p);
    const h4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)(p + 4 * LD));
    f16x8 r;
    r[0] = (_Float16)a[0]; r[1] = (_Float16)a[1]; r[2] = (_Float16)a[2]; r[3] = (_Float16)a[3];
    r[4] = (_Float16)b[0]; r[5] = (_Float16)b[1]; r[6] = (_Float16)b[2]; r[7] = (_Float16)b[3];
    return r;
}

// max |mass[n] * Phi[n][c]| per pair -> power-of-two scale.  The basis is streamed as one flat float4 array (all ld
// columns: entries beyond k can only make the scale more conservative), four loads in flight per thread.
// (TR = float | double: a float64 basis / mass is rounded to fp32 as it is loaded -- what the reference's fit does before it
//  projects, pyFM/functional.py:410-414 -- so both forms compute the same numbers)
// A float64 basis is also WRITTEN BACK as fp32 by this pass (phi32 / mass32, nullable): the tile kernel re-reads every slab
// of the basis once per descriptor tile from L2 and is bound by that traffic (87 us on fp32, 127 us straight from float64).
template <typename TR>
__global__ __launch_bounds__(256) void proj_absmax_kernel(const TR* __restrict__ Phi, const TR* __restrict__ mass, int N,
                                                          int ld, float* __restrict__ amax_part, float* __restrict__ phi32,
                                                          float* __restrict__ mass32, int n_part, dm_c00_args<TR> cz) {
    const int b = blockIdx.y;
    if ((int)blockIdx.x >= n_part) {                          // (uniform) the extra workgroup of dm_fmap_fit: the pair's c00
        __shared__ double red[2][4];
        dm_c00_body<TR>(cz, b, threadIdx.x, red);
        return;
    }
    const TR* P = Phi + (long long)b * N * ld;
    const TR* a = mass + (long long)b * N;
    const unsigned total = (unsigned)N * (unsigned)ld;           // < 2^31 per pair (checked by the caller)
    float m = 0.f;
    if (sizeof(TR) == 4 && ((ld & 3) == 0) && ((((uintptr_t)Phi) & 15) == 0)) {
        const unsigned nvec = total >> 2, ld4 = (unsigned)ld >> 2;
        const float4* P4 = reinterpret_cast<const float4*>(P);
        // each workgroup streams contiguous 16 KiB chunks (4 consecutive float4 per lane would be 64 B per lane; the
        // four loads of a lane are 4 KiB apart inside the chunk so that every wave instruction is one contiguous KiB):
        // large power-of-two strides between the loads in flight camp on a few HBM channels
        for (unsigned c0 = blockIdx.x * 1024; c0 < nvec; c0 += n_part * 1024) {
            float4 v[4];
            float an[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned e = c0 + u * 256 + threadIdx.x;
                const bool ok = e < nvec;
                v[u] = ok ? P4[e] : float4{0.f, 0.f, 0.f, 0.f};
                an[u] = ok ? fabsf((float)a[e / ld4]) : 0.f;       // 32-bit division (a 64-bit one dominated this kernel)
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                m = fmaxf(m, an[u] * fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
        }
    } else if (sizeof(TR) == 8 && ((ld & 1) == 0) && ((((uintptr_t)Phi) & 15) == 0)) {
        const unsigned nvec = total >> 1, ld2 = (unsigned)ld >> 1;
        const f64x2* P2 = reinterpret_cast<const f64x2*>(P);
        for (unsigned c0 = blockIdx.x * 1024; c0 < nvec; c0 += n_part * 1024) {
            f64x2 v[4];
            float an[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned e = c0 + u * 256 + threadIdx.x;
                const bool ok = e < nvec;
                v[u] = ok ? P2[e] : f64x2{0.0, 0.0};
                an[u] = ok ? fabsf((float)a[e / ld2]) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float x0 = (float)v[u][0], x1 = (float)v[u][1];
                m = fmaxf(m, an[u] * fmaxf(fabsf(x0), fabsf(x1)));
                const unsigned e = c0 + u * 256 + threadIdx.x;
                if (phi32 && e < nvec) reinterpret_cast<float2*>(phi32 + (long long)b * N * ld)[e] = float2{x0, x1};
            }
        }
    } else {
        for (unsigned e = blockIdx.x * 256 + threadIdx.x; e < total; e += n_part * 256) {
            m = fmaxf(m, fabsf((float)a[e / (unsigned)ld]) * fabsf((float)P[e]));
            if (sizeof(TR) == 8 && phi32) phi32[(long long)b * N * ld + e] = (float)P[e];
        }
    }
    if (sizeof(TR) == 8 && mass32) {
        for (unsigned e = blockIdx.x * 256 + threadIdx.x; e < (unsigned)N; e += n_part * 256) mass32[(long long)b * N + e] = (float)a[e];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    // one partial per workgroup, no atomics: 16 k same-line atomics serialise (~7 ns each) and took 100 us
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) amax_part[b * n_part + blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

template <typename TR>
struct proj_params {
    const TR* Phi; const TR* mass; const _Float16* F; const float* amax_part; int n_part;
    float* partial;          // (nsplit, B, k, D) fp32
    int B, N, D, k, ld, nsplit, kchunk, tiles_m, tiles_d;
};

template <typename TR>
__global__ __launch_bounds__(256, 2) void proj_f16split_kernel(proj_params<TR> p) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];               // [2 buffers][Xhi | Xlo | F], 68 KiB
    // 1-D grid, XCD-aware: the d-tiles that share one (pair, vertex chunk) slab of the basis are neighbours in the
    // logical order and therefore meet in the same XCD's L2 (otherwise every tile re-fetches the slab from HBM)
    const int ntile = p.tiles_m * p.tiles_d;
    const int id = xcd_remap(blockIdx.x, ntile * p.nsplit * p.B);
    const int tile = id % ntile;
    const int split = (id / ntile) % p.nsplit;
    const int b = id / (ntile * p.nsplit);
    const int tm = tile / p.tiles_d, td = tile - tm * p.tiles_d;
    const int m0 = tm * PT, d0 = td * PTD;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nbeg = split * p.kchunk, nend = min(p.N, nbeg + p.kchunk);

    // scale = 2^e with max |X| * scale in [2^13, 2^14)
    float amax = 0.f;
    for (int q = 0; q < p.n_part; ++q) amax = fmaxf(amax, p.amax_part[b * p.n_part + q]);
    int ex = 0;
    if (amax > 0.f) (void)frexpf(amax, &ex);                 // amax = f * 2^ex, f in [0.5, 1)
    const float scale = ldexpf(1.0f, 14 - ex);

    const TR* Phi = p.Phi + (long long)b * p.N * p.ld;
    const TR* mass = p.mass + (long long)b * p.N;
    const _Float16* F = p.F + (long long)b * p.N * p.D;

    f32x16 acc[2][PNT];                                      // wave tile 64 (m) x PTD / 2 (d)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < PNT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    const int srow = t >> 3, scol = (t & 7) * 16;            // staging: row of the stage, 16 consecutive basis columns
    const int fcol = (t & 7) * (PTD / 8);                    //          and PTD / 8 consecutive descriptor channels
    const bool xvec = ((p.ld & (sizeof(TR) == 4 ? 3 : 1)) == 0) && ((((uintptr_t)p.Phi) & 15) == 0);
    const bool fvec = ((p.D & 7) == 0) && ((((uintptr_t)p.F) & 15) == 0);
    // staged exactly as loaded: the mass scaling and the hi / lo split happen in PROJ_STASH, one stage later (arithmetic on
    // the loaded values inside PROJ_FETCH would make every fetch wait for its own data)
    TR xr[16], an_raw = (TR)0;
    u32x4 fr[PFV];
    typedef __attribute__((address_space(1))) const f32x4 gf32x4;
    typedef __attribute__((address_space(1))) const u32x4 gu32x4;
    typedef __attribute__((address_space(1))) const TR gfloat;
    typedef __attribute__((address_space(1))) const f64x2 gf64x2;
#define PROJ_FETCH(s_)                                                                                        \
    {                                                                                                         \
        const int n_ = nbeg + (s_) * PBK + srow;                                                              \
        const bool rv = n_ < nend;                                                                            \
        an_raw = rv ? ((gfloat*)mass)[n_] : (TR)0;                                                            \
        const TR* xrow = Phi + (long long)n_ * p.ld + m0 + scol;                                              \
        if (rv && xvec && m0 + scol + 15 < p.k) {                                                             \
            if constexpr (sizeof(TR) == 4) {                                                                  \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                               \
                    const f32x4 v = *(gf32x4*)(xrow + 4 * q);                                                 \
                    xr[4 * q] = v[0]; xr[4 * q + 1] = v[1]; xr[4 * q + 2] = v[2]; xr[4 * q + 3] = v[3];       \
                }                                                                                             \
            } else {                                                                                          \
                _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                               \
                    const f64x2 v = *(gf64x2*)(xrow + 2 * q);                                                 \
                    xr[2 * q] = v[0]; xr[2 * q + 1] = v[1];                                                   \
                }                                                                                             \
            }                                                                                                 \
        } else {                                                                                              \
            _Pragma("unroll") for (int q = 0; q < 16; ++q)                                                    \
                xr[q] = (rv && m0 + scol + q < p.k) ? xrow[q] : (TR)0;                                        \
        }                                                                                                     \
        const _Float16* frow = F + (long long)n_ * p.D + d0 + fcol;                                           \
        if (rv && fvec && d0 + fcol + PTD / 8 - 1 < p.D) {                                                    \
            _Pragma("unroll") for (int q = 0; q < PFV; ++q) fr[q] = *(gu32x4*)(frow + 8 * q);                 \
        } else {                                                                                              \
            _Pragma("unroll") for (int q = 0; q < PFV; ++q) {                                                 \
                f16x8 tmp;                                                                                    \
                _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                 \
                    tmp[e] = (rv && d0 + fcol + 8 * q + e < p.D) ? frow[8 * q + e] : (_Float16)0.f;           \
                fr[q] = *reinterpret_cast<const u32x4*>(&tmp);                                                \
            }                                                                                                 \
        }                                                                                                     \
    }
#define PROJ_STASH(buf_)                                                                                      \
    {                                                                                                         \
        _Float16* Xh = smem + (buf_) * PSTAGE + srow * PLD + scol;                                            \
        _Float16* Xl = Xh + PBK * PLD;                                                                        \
        _Float16* Fs = smem + (buf_) * PSTAGE + 2 * PBK * PLD + srow * PLDF + fcol;                           \
        f16x8 h[2], l[2];                                                                                     \
        const float an = (float)an_raw * scale;                                                               \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                      \
            float xs_ = (float)xr[q] * an;                                                                         \
            /* the product must be ONE rounded value for both pieces: left visible, the compiler contracts it into the */ \
            /* conversion of hi (v_fma_mix, single rounding) but not into the one it stores, and hi + lo is off by an */ \
            /* ulp of hi on ties */                                                                           \
            asm volatile("" : "+v"(xs_));                                                                     \
            const _Float16 hi = (_Float16)xs_;                                                                \
            h[q >> 3][q & 7] = hi;                                                                            \
            l[q >> 3][q & 7] = (_Float16)(xs_ - (float)hi);                                                   \
        }                                                                                                     \
        *reinterpret_cast<f16x8*>(Xh) = h[0]; *reinterpret_cast<f16x8*>(Xh + 8) = h[1];                       \
        *reinterpret_cast<f16x8*>(Xl) = l[0]; *reinterpret_cast<f16x8*>(Xl + 8) = l[1];                       \
        _Pragma("unroll") for (int q = 0; q < PFV; ++q) *reinterpret_cast<u32x4*>(Fs + 8 * q) = fr[q];        \
    }

    const int ns = (nend - nbeg + PBK - 1) / PBK;
    if (ns > 0) {
        PROJ_FETCH(0)
        PROJ_STASH(0)
    }
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        const int buf = s & 1;
        if (s + 1 < ns) PROJ_FETCH(s + 1)
        const _Float16* Xh = smem + buf * PSTAGE;
        const _Float16* Xl = Xh + PBK * PLD;
        const _Float16* Fs = Xl + PBK * PLD;
#pragma unroll
        for (int ks = 0; ks < PBK / 16; ++ks) {
            const int row0 = ks * 16 + 8 * (lane >> 5);
            const int sub = 16 * ((lane >> 4) & 1) + (lane & 15) - (lane & 15);   // 16-column half of the 32-wide tile
            f16x8 ah[2], al[2], bf[PNT];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                ah[x] = tr_frag<PLD>(Xh, row0, wm * 64 + x * 32 + sub, lane);
                al[x] = tr_frag<PLD>(Xl, row0, wm * 64 + x * 32 + sub, lane);
            }
#pragma unroll
            for (int x = 0; x < PNT; ++x) bf[x] = tr_frag<PLDF>(Fs, row0, wn * (PTD / 2) + x * 32 + sub, lane);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < PNT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bf[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bf[nt], acc[mt][nt], 0, 0, 0);
                }
        }
        if (s + 1 < ns) { PROJ_STASH(buf ^ 1) }
        __syncthreads();
    }
#undef PROJ_FETCH
#undef PROJ_STASH

    // acc[mt][nt][r] = O[m = m0 + wm*64 + mt*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)][d = d0 + wn*128 + nt*32 + (lane&31)]
    const float inv_scale = 1.0f / scale;
    float* out = p.partial + ((long long)split * p.B + b) * p.k * p.D;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < PNT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int d = d0 + wn * (PTD / 2) + nt * 32 + (lane & 31);
                if (m < p.k && d < p.D) out[(long long)m * p.D + d] = acc[mt][nt][r] * inv_scale;
            }
}

// ---- one pass over the basis (r04) ---------------------------------------------------------------------------------------
// The kernel above needs max |mass Phi| of the pair before it can split a value: a pass of its own over the basis (project_absmax:
// 2 x 48 us and 410 MB of the config-2 step for a maximum), plus the fp32 copy that pass leaves behind because four d-tiles
// re-reading a float64 slab from L2 bound the tile kernel.  This kernel reads the basis ONCE, as it is:
//  * the scale is a RUNNING one per workgroup: the split-K partial of a workgroup is written back unscaled, so nothing outside
//    the workgroup ever sees its scale.  Every stage (32 vertices x 128 basis columns) publishes its max |mass Phi| through an
//    LDS max one stage before it is split; when a stage raises the running maximum past the current power of two, the
//    accumulators are multiplied by the (exact) ratio of the two scales and the loop goes on.  The scale only ever shrinks:
//    no value can overflow, and a value far below the running maximum loses low bits of its low half only (fp16 subnormals
//    degrade gradually: the absolute error stays below 2^-25 of the scaled maximum).
//  * 8 waves per workgroup, tile 128 basis columns x 384 descriptor channels: two d-tiles (not four) re-read a slab, which from
//    float64 is the L2 traffic the fp32 copy had; half the split arithmetic per output.
//  * a wave whose 32-column blocks lie beyond k skips their matrix instructions (k = 200: one block in eight).
constexpr int P2TD = 384;                 // descriptor channels per tile
constexpr int P2NT = 3;                   // 32-channel blocks per wave (4 waves along d)
constexpr int P2LDF = 416;                // LDS row stride of the descriptor rows in halves (832 B = 208 dwords = 16 mod 64)
constexpr int P2STAGE = 2 * PBK * PLD + PBK * P2LDF;   // halves per stage buffer: Xhi | Xlo | F
constexpr size_t P2_LDS = (size_t)2 * P2STAGE * sizeof(_Float16) + 16;   // two buffers + three max slots

template <typename TR>
struct proj2_params {
    const TR* Phi; const TR* mass; const _Float16* F;
    float* partial;          // (nsplit, B, k, D) fp32
    int B, N, D, k, ld, nsplit, kchunk, tiles_m, tiles_d, ntiles;
    dm_c00_args<TR> cz; int with_c00;
    int flip;                // 1: the two waves of a SIMD run the halves of an iteration in opposite order (0: experiments)
};

template <typename TR, int NMT>
__device__ __forceinline__ void proj2_body(const proj2_params<TR>& p, _Float16* smem, unsigned* slot, int b, int split, int m0, int d0) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nbeg = split * p.kchunk, nend = min(p.N, nbeg + p.kchunk);
    const TR* Phi = p.Phi + (long long)b * p.N * p.ld;
    const TR* mass = p.mass + (long long)b * p.N;
    const _Float16* F = p.F + (long long)b * p.N * p.D;

    f32x16 acc[NMT > 0 ? NMT : 1][P2NT];
#pragma unroll
    for (int a = 0; a < (NMT > 0 ? NMT : 1); ++a)
#pragma unroll
        for (int c = 0; c < P2NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    const int srow = t >> 4, scol = (t & 15) * 8;            // staging: row of the stage, 8 consecutive basis columns
    const int fcol = (t & 15) * (P2TD / 16);                 //          and 24 consecutive descriptor channels
    // Loads run a full iteration ahead of their first use (one workgroup per CU: nothing else covers their latency): the basis
    // slab of stage s + 3 and the descriptor rows of stage s + 2 are requested at the top of iteration s; register sets A / B
    // alternate by stage parity.  Every load is an UNCONDITIONAL buffer load (rows past the pair read as 0; rows past the chunk
    // and columns past k are zeroed where the value is used): with loads under conditions the compiler cannot count the ones in
    // flight and waits for all of them -- also the ones it issued a moment ago.
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    constexpr int NQ = (int)sizeof(TR) / 2;                  // 16-byte loads per thread and stage: 8 values
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<TR*>(Phi), (short)0, (int)((unsigned)p.N * p.ld * sizeof(TR)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<TR*>(mass), (short)0, (int)((unsigned)p.N * sizeof(TR)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(F), (short)0, (int)((unsigned)p.N * p.D * 2), 0x00020000);
    unsigned colmask = 0;                                    // bit q: basis column m0 + scol + q exists
#pragma unroll
    for (int q = 0; q < 8; ++q) colmask |= (m0 + scol + q < p.k) ? (1u << q) : 0u;
    i32x4_t xrA[NQ], xrB[NQ], anA, anB;                      // stages s + 2 / s + 3 as loaded (an: [0 .. 1] hold the mass)
    float xf[8];                                             // stage s + 1: fl(fl32(Phi) * fl32(mass)), not yet scaled
    i32x4_t frA[3], frB[3];
#define P2_FENCE() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define P2_FETCH_X(s_, xr, an_raw)                                                                            \
    {                                                                                                         \
        const int n_ = nbeg + (s_) * PBK + srow;                                                              \
        const int vo = colmask ? (n_ * p.ld + m0 + scol) * (int)sizeof(TR) : 0x7fffff00;   /* no column: reads 0 */ \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) xr[q] = __builtin_amdgcn_raw_buffer_load_b128(rX, vo + 16 * q, 0, 0);   \
        if constexpr (sizeof(TR) == 8) {                                                                      \
            const auto m2 = __builtin_amdgcn_raw_buffer_load_b64(rM, n_ * 8, 0, 0);                           \
            an_raw[0] = m2[0]; an_raw[1] = m2[1];                                                             \
        } else {                                                                                              \
            an_raw[0] = __builtin_amdgcn_raw_buffer_load_b32(rM, n_ * 4, 0, 0);                               \
        }                                                                                                     \
        P2_FENCE()                                                                                            \
    }
#define P2_FETCH_F(s_, fr)                                                                                    \
    {                                                                                                         \
        const int n_ = nbeg + (s_) * PBK + srow;                                                              \
        const int vo = (n_ * p.D + d0 + fcol) * 2;                                                            \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) fr[q] = __builtin_amdgcn_raw_buffer_load_b128(rF, vo + 16 * q, 0, 0);   \
        P2_FENCE()                                                                                            \
    }
    // the loaded stage becomes fp32 products (the rounding the reference's fit applies: basis and mass rounded to fp32, one
    // rounded product); their maximum goes to the stage's slot.  Non-negative floats order like their bit patterns.
#define P2_PUBLISH(s_, xr, an_raw)                                                                            \
    {                                                                                                         \
        const bool rv = nbeg + (s_) * PBK + srow < nend;                                                      \
        float an;                                                                                             \
        if constexpr (sizeof(TR) == 8) an = (float)__hiloint2double(an_raw[1], an_raw[0]);                    \
        else an = __int_as_float(an_raw[0]);                                                                  \
        an = rv ? an : 0.f;              /* a row of the next chunk: finite data times 0 */                   \
        float mx = 0.f;                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                       \
            float x_;                                                                                         \
            if constexpr (sizeof(TR) == 8) x_ = (float)__hiloint2double(xr[q >> 1][2 * (q & 1) + 1], xr[q >> 1][2 * (q & 1)]);   \
            else x_ = __int_as_float(xr[q >> 2][q & 3]);                                                      \
            float v_ = x_ * an;                                                                               \
            asm volatile("" : "+v"(v_));                                                                      \
            xf[q] = v_;                                                                                       \
        }                                                                                                     \
        if (colmask != 0xffu && colmask != 0u) {     /* the one thread per row whose columns straddle k: whatever lies */ \
            _Pragma("unroll") for (int q = 0; q < 8; ++q) xf[q] = ((colmask >> q) & 1u) ? xf[q] : 0.f;   /* behind k must not reach the scale */ \
        }                                                                                                     \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) mx = fmaxf(mx, fabsf(xf[q]));                           \
        /* wave maximum: four DPP steps inside the rows of 16, the four row results through scalar registers, ONE LDS   \
           atomic per wave (left to the compiler, same-address atomics become a scalar loop over the active lanes) */     \
        mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0xB1, 0xF, 0xF, true)));   \
        mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x4E, 0xF, 0xF, true)));   \
        mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x141, 0xF, 0xF, true)));  \
        mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x140, 0xF, 0xF, true)));  \
        const unsigned mu = __builtin_bit_cast(unsigned, mx);                                                 \
        const unsigned m01 = max((unsigned)__builtin_amdgcn_readlane((int)mu, 0), (unsigned)__builtin_amdgcn_readlane((int)mu, 16));   \
        const unsigned m23 = max((unsigned)__builtin_amdgcn_readlane((int)mu, 32), (unsigned)__builtin_amdgcn_readlane((int)mu, 48));  \
        const unsigned mw = max(m01, m23);                                                                    \
        if (lane == 0 && mw != 0u) __hip_atomic_fetch_max(&slot[(s_) % 3], mw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);        \
    }
    // split stage s_ with the running scale (max |X| * scale in [2^13, 2^14) at the moment the scale was set)
#define P2_STASH(s_, buf_, fr)                                                                                   \
    {                                                                                                         \
        const float smx = __builtin_bit_cast(float, slot[(s_) % 3]);                                          \
        if (smx > run_max) {                                                                                  \
            run_max = smx;                                                                                    \
            int ex = 0;                                                                                       \
            (void)frexpf(smx, &ex);                                                                           \
            const int e2 = min(14 - ex, 100);                                                                 \
            if (e2 < cur_e) {           /* the accumulators follow once the current stage's products are in (P2_RESCALE) */ \
                ratio = ldexpf(1.0f, e2 - cur_e);                                                             \
                cur_e = e2;                                                                                   \
                scale = ldexpf(1.0f, e2);                                                                     \
            }                                                                                                 \
        }                                                                                                     \
        _Float16* Xh = smem + (buf_) * P2STAGE + srow * PLD + scol;                                           \
        _Float16* Xl = Xh + PBK * PLD;                                                                        \
        _Float16* Fs = smem + (buf_) * P2STAGE + 2 * PBK * PLD + srow * P2LDF + fcol;                         \
        f16x8 h, l;                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                       \
            const float xs_ = xf[q] * scale;                                                                  \
            const _Float16 hi = (_Float16)xs_;                                                                \
            h[q] = hi;                                                                                        \
            l[q] = (_Float16)(xs_ - (float)hi);                                                               \
        }                                                                                                     \
        *reinterpret_cast<f16x8*>(Xh) = h;                                                                    \
        *reinterpret_cast<f16x8*>(Xl) = l;                                                                    \
        const bool rvf = nbeg + (s_) * PBK + srow < nend;                                                     \
        _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                         \
            *reinterpret_cast<i32x4_t*>(Fs + 8 * q) = rvf ? fr[q] : i32x4_t{0, 0, 0, 0};                      \
    }

#define P2_RESCALE()                                                                                          \
    if (ratio != 1.0f) {                                                                                      \
        _Pragma("unroll") for (int a = 0; a < (NMT > 0 ? NMT : 1); ++a)                                       \
            _Pragma("unroll") for (int c = 0; c < P2NT; ++c)                                                  \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[a][c][r] *= ratio;                         \
        ratio = 1.0f;                                                                                         \
    }
    float run_max = 0.f, scale = ldexpf(1.0f, 100), ratio = 1.0f;
    int cur_e = 100;
    const int ns = (nend - nbeg + PBK - 1) / PBK;
    if (t < 3) slot[t] = 0u;
    __syncthreads();
    if (ns > 0) {
        P2_FETCH_X(0, xrA, anA)
        P2_FETCH_F(0, frA)
        P2_FETCH_X(1, xrB, anB)
        P2_FETCH_F(1, frB)
        P2_PUBLISH(0, xrA, anA)
        P2_FETCH_X(2, xrA, anA)
    }
    __syncthreads();
    if (ns > 0) {
        P2_STASH(0, 0, frA)
        ratio = 1.0f;                                        // (nothing accumulated yet)
        if (ns > 1) P2_PUBLISH(1, xrB, anB)
    }
    __syncthreads();
    // iteration s: XL / ANL receive stage s + 3, XU / ANU hold stage s + 2; FL receives stage s + 2, FU holds stage s + 1
#define P2_MMA(buf_)                                                                                          \
        if constexpr (NMT > 0) {                                                                              \
            const _Float16* Xh = smem + (buf_) * P2STAGE;                                                     \
            const _Float16* Xl = Xh + PBK * PLD;                                                              \
            const _Float16* Fs = Xl + PBK * PLD;                                                              \
            _Pragma("unroll") for (int ks = 0; ks < PBK / 16; ++ks) {                                         \
                const int row0 = ks * 16 + 8 * (lane >> 5);                                                   \
                const int sub = 16 * ((lane >> 4) & 1);          /* 16-column half of the 32-wide tile */      \
                f16x8 ah[NMT > 0 ? NMT : 1], al[NMT > 0 ? NMT : 1], bf[P2NT];                                 \
                _Pragma("unroll") for (int x = 0; x < NMT; ++x) {                                             \
                    ah[x] = tr_frag<PLD>(Xh, row0, wm * 64 + x * 32 + sub, lane);                             \
                    al[x] = tr_frag<PLD>(Xl, row0, wm * 64 + x * 32 + sub, lane);                             \
                }                                                                                             \
                _Pragma("unroll") for (int x = 0; x < P2NT; ++x)                                              \
                    bf[x] = tr_frag<P2LDF>(Fs, row0, wn * (P2TD / 4) + x * 32 + sub, lane);                   \
                _Pragma("unroll") for (int mt = 0; mt < NMT; ++mt)                                            \
                    _Pragma("unroll") for (int nt = 0; nt < P2NT; ++nt) {                                     \
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bf[nt], acc[mt][nt], 0, 0, 0);   \
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bf[nt], acc[mt][nt], 0, 0, 0);   \
                    }                                                                                         \
            }                                                                                                 \
        }
    // The two waves of a SIMD (w and w + 4) run the halves of an iteration in opposite order: one splits and publishes (VALU,
    // LDS writes) while the other multiplies (matrix pipe).  In lockstep -- all eight waves in the same phase, which is what one
    // barrier per stage gives -- the two pipes take turns idling.
#define P2_ITER(s_, XL, ANL, XU, ANU, FL, FU)                                                                 \
    {                                                                                                         \
        const int buf = (s_) & 1;                                                                             \
        P2_FETCH_X((s_) + 3, XL, ANL)                        /* (past the chunk: zeroed at use) */                 \
        P2_FETCH_F((s_) + 2, FL)                                                                              \
        if (p.flip && wave >= 4) {                                                                            \
            if ((s_) + 1 < ns) { P2_STASH((s_) + 1, buf ^ 1, FU) }                                            \
            if ((s_) + 2 < ns) { P2_PUBLISH((s_) + 2, XU, ANU) }                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            P2_MMA(buf)                                                                                       \
        } else {                                                                                              \
            P2_MMA(buf)                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            if ((s_) + 1 < ns) { P2_STASH((s_) + 1, buf ^ 1, FU) }                                            \
            if ((s_) + 2 < ns) { P2_PUBLISH((s_) + 2, XU, ANU) }                                              \
        }                                                                                                     \
        P2_RESCALE()                                                                                          \
        if (t == 0) slot[(s_) % 3] = 0u;                     /* the slot of stage s + 3 (read for stage s one barrier ago) */ \
        __syncthreads();                                                                                      \
    }
    for (int s = 0; s < ns; s += 2) {
        P2_ITER(s, xrB, anB, xrA, anA, frA, frB)
        if (s + 1 < ns) P2_ITER(s + 1, xrA, anA, xrB, anB, frB, frA)
    }
#undef P2_ITER
#undef P2_FENCE
#undef P2_MMA
#undef P2_RESCALE
#undef P2_FETCH_X
#undef P2_FETCH_F
#undef P2_PUBLISH
#undef P2_STASH

    if constexpr (NMT > 0) {
        const float inv_scale = 1.0f / scale;
        float* out = p.partial + ((long long)split * p.B + b) * p.k * p.D;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int nt = 0; nt < P2NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int d = d0 + wn * (P2TD / 4) + nt * 32 + (lane & 31);
                    if (m < p.k && d < p.D) out[(long long)m * p.D + d] = acc[mt][nt][r] * inv_scale;
                }
    }
}

template <typename TR>
__global__ __launch_bounds__(512, 2) void proj_onepass_kernel(proj2_params<TR> p) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];               // [2 buffers][Xhi | Xlo | F] | max slots
    const int t = threadIdx.x;
    if ((int)blockIdx.x >= p.ntiles) {                        // (uniform) the extra workgroups of dm_fmap_fit: the pairs' c00
        __shared__ double red[2][4];
        dm_c00_body<TR>(p.cz, (int)blockIdx.x - p.ntiles, t, red);
        return;
    }
    unsigned* slot = reinterpret_cast<unsigned*>(smem + 2 * P2STAGE);
    // XCD-aware: the tiles that share one (pair, vertex chunk) slab of the basis / of the descriptors are neighbours in the
    // logical order and meet in the same XCD's L2
    const int ntile = p.tiles_m * p.tiles_d;
    const int id = xcd_remap(blockIdx.x, p.ntiles);
    const int tile = id % ntile;
    const int split = (id / ntile) % p.nsplit;
    const int b = id / (ntile * p.nsplit);
    const int tm = tile / p.tiles_d, td = tile - tm * p.tiles_d;
    const int m0 = tm * PT, d0 = td * P2TD;
    const int wm = __builtin_amdgcn_readfirstlane(t >> 8);
    const int left = p.k - (m0 + wm * 64);                    // basis columns this wave's two blocks still cover
    if (left > 32) proj2_body<TR, 2>(p, smem, slot, b, split, m0, d0);
    else if (left > 0) proj2_body<TR, 1>(p, smem, slot, b, split, m0, d0);
    else proj2_body<TR, 0>(p, smem, slot, b, split, m0, d0);
}

__global__ __launch_bounds__(256) void proj_reduce_kernel(const float* __restrict__ partial, int nsplit, long long n,
                                                          float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int q = 0; q < nsplit; ++q) s += (double)partial[(long long)q * n + i];
    out[i] = (float)s;
}

// workspace of one projection (dm_project_f16split_launch)
size_t dm_project_f16split_ws(int B, int N, int D, int k, int ld, int real_bytes) {
    const int nsplit = dm_cdiv(N, 1024);
    return dm_align_up((size_t)nsplit * B * k * D * 4) + dm_align_up((size_t)B * 64 * 4) + 4096 +
           (real_bytes == 8 ? dm_align_up((size_t)B * N * ld * 4) + dm_align_up((size_t)B * N * 4) : 0);
}

// The launches of one projection, workspace taken from what the caller reserved.  Ared != null: the split-K partials are
// reduced into it.  Ared == null: they are left as they are, *partial_out (nsplit, B, k, D) / *nsplit_out tell where (dm_fmap_fit
// hands them to the Gram kernel, which adds them up as it reads).
template <typename TR>
int dm_project_f16split_launch(dm_ctx* ctx, int B, int N, int D, int k, const TR* Phi, int ld, const TR* mass,
                               const void* F, float* Ared, const float** partial_out, int* nsplit_out, const dm_c00_args<TR>* cz) {
    proj_params<TR> p;
    p.Phi = Phi; p.mass = mass; p.F = (const _Float16*)F;
    p.B = B; p.N = N; p.D = D; p.k = k; p.ld = ld;
    p.tiles_m = dm_cdiv(k, PT); p.tiles_d = dm_cdiv(D, PTD);
    // split-K by a fixed chunk of vertices: the summation order of a pair must not depend on the batch it is in
    p.kchunk = 1024;
    const int nsplit = dm_cdiv(N, p.kchunk);
    p.nsplit = nsplit;
    if ((long long)N * ld >= (1ll << 31)) return dm_fail(ctx, DM_EINVAL, "dm_project: N * ld must be below 2^31");
    const size_t pbytes = (size_t)nsplit * B * k * D * 4;
    int rc = DM_OK;
    p.partial = (float*)dm_ws_take(ctx, pbytes);
    const int n_part = max(1, min(64, (int)(((long long)N * ld) / 4096)));
    float* amax_part = (float*)dm_ws_take(ctx, (size_t)B * n_part * 4);
    if (!p.partial || !amax_part) return dm_fail(ctx, DM_ENOMEM, "dm_project: workspace not reserved");
    p.amax_part = amax_part; p.n_part = n_part;
    const size_t lds = (size_t)2 * PSTAGE * sizeof(_Float16);
    // (its buffer loads take 32-bit byte offsets into a pair's arrays and dword-aligned descriptor rows)
    const bool onepass_ok = (long long)N * ld * (long long)sizeof(TR) < (1ll << 31) - 4096 && (long long)N * D * 2 < (1ll << 31) && (D & 1) == 0;
    if (ctx->opt_proj_onepass && onepass_ok) {
        // one pass over the basis as it is (running scale per workgroup): no maxima pass, no fp32 copy
        proj2_params<TR> q;
        q.Phi = Phi; q.mass = mass; q.F = p.F; q.partial = p.partial;
        q.B = B; q.N = N; q.D = D; q.k = k; q.ld = ld; q.nsplit = nsplit; q.kchunk = p.kchunk;
        q.tiles_m = p.tiles_m; q.tiles_d = dm_cdiv(D, P2TD); q.ntiles = q.tiles_m * q.tiles_d * nsplit * B;
        q.with_c00 = cz ? 1 : 0;
        q.flip = 1;
#ifdef DM_EXPERIMENTS
        q.flip = dm_knob("DM_PROJ_FLIP", 1);
#endif
        q.cz = cz ? *cz : dm_c00_args<TR>{};
        rc = dm_grant_lds(ctx, (const void*)proj_onepass_kernel<TR>, P2_LDS);
        if (rc) return rc;
        DM_LAUNCH(ctx, "project_f16split_mfma", proj_onepass_kernel<TR>, dim3(q.ntiles + (cz ? B : 0)), dim3(512), P2_LDS, q);
    } else
    if constexpr (sizeof(TR) == 8) {
        // float64 basis: the scale pass also leaves an fp32 copy (the rounding the reference's fit applies), the tile kernel runs on it
        float* phi32 = (float*)dm_ws_take(ctx, (size_t)B * N * ld * 4);
        float* mass32 = (float*)dm_ws_take(ctx, (size_t)B * N * 4);
        if (!phi32 || !mass32) return dm_fail(ctx, DM_ENOMEM, "dm_project: workspace not reserved");
        DM_LAUNCH(ctx, "project_absmax", proj_absmax_kernel<TR>, dim3(n_part + (cz ? 1 : 0), B), dim3(256), 0, Phi, mass, N, ld, amax_part, phi32,
                  mass32, n_part, cz ? *cz : dm_c00_args<TR>{});
        proj_params<float> pf;
        pf.Phi = phi32; pf.mass = mass32; pf.F = p.F; pf.amax_part = p.amax_part; pf.n_part = p.n_part; pf.partial = p.partial;
        pf.B = B; pf.N = N; pf.D = D; pf.k = k; pf.ld = ld; pf.nsplit = p.nsplit; pf.kchunk = p.kchunk; pf.tiles_m = p.tiles_m; pf.tiles_d = p.tiles_d;
        rc = dm_grant_lds(ctx, (const void*)proj_f16split_kernel<float>, lds);
        if (rc) return rc;
        DM_LAUNCH(ctx, "project_f16split_mfma", proj_f16split_kernel<float>, dim3(p.tiles_m * p.tiles_d * nsplit * B), dim3(256), lds, pf);
    } else {
        DM_LAUNCH(ctx, "project_absmax", proj_absmax_kernel<TR>, dim3(n_part + (cz ? 1 : 0), B), dim3(256), 0, Phi, mass, N, ld, amax_part,
                  (float*)nullptr, (float*)nullptr, n_part, cz ? *cz : dm_c00_args<TR>{});
        rc = dm_grant_lds(ctx, (const void*)proj_f16split_kernel<TR>, lds);
        if (rc) return rc;
        DM_LAUNCH(ctx, "project_f16split_mfma", proj_f16split_kernel<TR>, dim3(p.tiles_m * p.tiles_d * nsplit * B), dim3(256), lds, p);
    }
    if (partial_out) *partial_out = p.partial;
    if (nsplit_out) *nsplit_out = nsplit;
    if (Ared) {
        const long long n = (long long)B * k * D;
        DM_LAUNCH(ctx, "project_reduce", proj_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.partial, nsplit, n,
                  Ared);
    }
    return DM_OK;
}
template int dm_project_f16split_launch<float>(dm_ctx*, int, int, int, int, const float*, int, const float*, const void*, float*, const float**, int*,
                                               const dm_c00_args<float>*);
template int dm_project_f16split_launch<double>(dm_ctx*, int, int, int, int, const double*, int, const double*, const void*, float*, const float**, int*,
                                                const dm_c00_args<double>*);

template <typename TR>
int dm_project_f16split(dm_ctx* ctx, int B, int N, int D, int k, const TR* Phi, int ld, const TR* mass,
                        const void* F, float* Ared) {
    int rc = dm_ws_reserve(ctx, dm_project_f16split_ws(B, N, D, k, ld, (int)sizeof(TR)));
    if (rc) return rc;
    return dm_project_f16split_launch<TR>(ctx, B, N, D, k, Phi, ld, mass, F, Ared, nullptr, nullptr, nullptr);
}
template int dm_project_f16split<float>(dm_ctx*, int, int, int, int, const float*, int, const float*, const void*, float*);
template int dm_project_f16split<double>(dm_ctx*, int, int, int, int, const double*, int, const double*, const void*, float*);

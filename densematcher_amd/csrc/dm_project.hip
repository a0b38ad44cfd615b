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

typedef __fp16 h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr int PT = 128;      // output tile (m x d)
constexpr int PBK = 32;      // vertices per stage
constexpr int PLD = 160;     // LDS row stride in halves (320 B): rows land 16 banks apart -> conflict-free tr reads

__device__ __forceinline__ f16x8 tr_frag(const _Float16* base, int row0, int col, int lane) {
    // 16-lane group: lane t supplies the address of row (row0 + (t >> 2)), columns col + 4 (t & 3) .. +3 and
    // receives column (col + t), rows row0 .. row0 + 3.  Two reads give 8 consecutive k.
    const int t = lane & 15;
    const _Float16* p = base + (row0 + (t >> 2)) * PLD + col + 4 * (t & 3);
    const h4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)p);
    const h4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)(p + 4 * PLD));
    f16x8 r;
    r[0] = (_Float16)a[0]; r[1] = (_Float16)a[1]; r[2] = (_Float16)a[2]; r[3] = (_Float16)a[3];
    r[4] = (_Float16)b[0]; r[5] = (_Float16)b[1]; r[6] = (_Float16)b[2]; r[7] = (_Float16)b[3];
    return r;
}

// max |mass[n] * Phi[n][m]| per pair -> power-of-two scale.  One 16-lane group per row, float4 per lane when the
// layout allows it; many workgroups per pair so the 4 MB basis streams at HBM speed.
__global__ __launch_bounds__(256) void proj_absmax_kernel(const float* __restrict__ Phi, const float* __restrict__ mass, int N,
                                                          int k, int ld, unsigned int* __restrict__ amax) {
    const int b = blockIdx.y;
    const float* P = Phi + (long long)b * N * ld;
    const float* a = mass + (long long)b * N;
    const int sub = threadIdx.x & 15, rowl = threadIdx.x >> 4;          // 16 rows per pass per workgroup
    const bool vec = ((ld & 3) == 0) && ((((uintptr_t)Phi) & 15) == 0);
    float m = 0.f;
    for (int n = blockIdx.x * 16 + rowl; n < N; n += gridDim.x * 16) {
        const float an = fabsf(a[n]);
        const float* row = P + (long long)n * ld;
        float rm = 0.f;
        if (vec) {
            for (int c = sub * 4; c < k; c += 64) {
                if (c + 3 < k) {
                    const float4 v = *reinterpret_cast<const float4*>(row + c);
                    rm = fmaxf(fmaxf(rm, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
                } else {
                    for (int e = c; e < k; ++e) rm = fmaxf(rm, fabsf(row[e]));
                }
            }
        } else {
            for (int c = sub; c < k; c += 16) rm = fmaxf(rm, fabsf(row[c]));
        }
        m = fmaxf(m, an * rm);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(amax + b, __float_as_uint(m));
}

struct proj_params {
    const float* Phi; const float* mass; const _Float16* F; const unsigned int* amax;
    float* partial;          // (nsplit, B, k, D) fp32
    int B, N, D, k, ld, nsplit, kchunk, tiles_m, tiles_d;
};

__global__ __launch_bounds__(256, 2) void proj_f16split_kernel(proj_params p) {
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * 3 * PBK * PLD];     // [2 buffers][Xhi | Xlo | F], 60 KiB
    const int tile = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int tm = tile / p.tiles_d, td = tile - tm * p.tiles_d;
    const int m0 = tm * PT, d0 = td * PT;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nbeg = split * p.kchunk, nend = min(p.N, nbeg + p.kchunk);

    // scale = 2^e with max |X| * scale in [2^13, 2^14)
    const float amax = __uint_as_float(p.amax[b]);
    int ex = 0;
    if (amax > 0.f) (void)frexpf(amax, &ex);                 // amax = f * 2^ex, f in [0.5, 1)
    const float scale = ldexpf(1.0f, 14 - ex);

    const float* Phi = p.Phi + (long long)b * p.N * p.ld;
    const float* mass = p.mass + (long long)b * p.N;
    const _Float16* F = p.F + (long long)b * p.N * p.D;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    const int srow = t >> 3, scol = (t & 7) * 16;            // staging: row of the stage, 16 consecutive columns
    const bool xvec = ((p.ld & 3) == 0) && ((((uintptr_t)p.Phi) & 15) == 0);
    const bool fvec = ((p.D & 7) == 0) && ((((uintptr_t)p.F) & 15) == 0);
    float xr[16];
    uint4 fr[2];
#define PROJ_FETCH(s_)                                                                                        \
    {                                                                                                         \
        const int n_ = nbeg + (s_) * PBK + srow;                                                              \
        const bool rv = n_ < nend;                                                                            \
        const float an = rv ? mass[n_] * scale : 0.f;                                                         \
        const float* xrow = Phi + (long long)n_ * p.ld + m0 + scol;                                           \
        if (rv && xvec && m0 + scol + 15 < p.k) {                                                             \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                   \
                const float4 v = *reinterpret_cast<const float4*>(xrow + 4 * q);                              \
                xr[4 * q] = v.x * an; xr[4 * q + 1] = v.y * an; xr[4 * q + 2] = v.z * an; xr[4 * q + 3] = v.w * an; \
            }                                                                                                 \
        } else {                                                                                              \
            _Pragma("unroll") for (int q = 0; q < 16; ++q)                                                    \
                xr[q] = (rv && m0 + scol + q < p.k) ? xrow[q] * an : 0.f;                                     \
        }                                                                                                     \
        const _Float16* frow = F + (long long)n_ * p.D + d0 + scol;                                           \
        if (rv && fvec && d0 + scol + 15 < p.D) {                                                             \
            fr[0] = *reinterpret_cast<const uint4*>(frow);                                                    \
            fr[1] = *reinterpret_cast<const uint4*>(frow + 8);                                                \
        } else {                                                                                              \
            _Float16 tmp[16];                                                                                 \
            _Pragma("unroll") for (int q = 0; q < 16; ++q)                                                    \
                tmp[q] = (rv && d0 + scol + q < p.D) ? frow[q] : (_Float16)0.f;                               \
            fr[0] = *reinterpret_cast<const uint4*>(tmp);                                                     \
            fr[1] = *reinterpret_cast<const uint4*>(tmp + 8);                                                 \
        }                                                                                                     \
    }
#define PROJ_STASH(buf_)                                                                                      \
    {                                                                                                         \
        _Float16* Xh = smem + (buf_) * 3 * PBK * PLD + srow * PLD + scol;                                     \
        _Float16* Xl = Xh + PBK * PLD;                                                                        \
        _Float16* Fs = Xl + PBK * PLD;                                                                        \
        f16x8 h[2], l[2];                                                                                     \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                      \
            const _Float16 hi = (_Float16)xr[q];                                                              \
            h[q >> 3][q & 7] = hi;                                                                            \
            l[q >> 3][q & 7] = (_Float16)(xr[q] - (float)hi);                                                 \
        }                                                                                                     \
        *reinterpret_cast<f16x8*>(Xh) = h[0]; *reinterpret_cast<f16x8*>(Xh + 8) = h[1];                       \
        *reinterpret_cast<f16x8*>(Xl) = l[0]; *reinterpret_cast<f16x8*>(Xl + 8) = l[1];                       \
        *reinterpret_cast<uint4*>(Fs) = fr[0]; *reinterpret_cast<uint4*>(Fs + 8) = fr[1];                     \
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
        const _Float16* Xh = smem + buf * 3 * PBK * PLD;
        const _Float16* Xl = Xh + PBK * PLD;
        const _Float16* Fs = Xl + PBK * PLD;
#pragma unroll
        for (int ks = 0; ks < PBK / 16; ++ks) {
            const int row0 = ks * 16 + 8 * (lane >> 5);
            const int sub = 16 * ((lane >> 4) & 1) + (lane & 15) - (lane & 15);   // 16-column half of the 32-wide tile
            f16x8 ah[2], al[2], bf[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                ah[x] = tr_frag(Xh, row0, wm * 64 + x * 32 + sub, lane);
                al[x] = tr_frag(Xl, row0, wm * 64 + x * 32 + sub, lane);
                bf[x] = tr_frag(Fs, row0, wn * 64 + x * 32 + sub, lane);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bf[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bf[nt], acc[mt][nt], 0, 0, 0);
                }
        }
        if (s + 1 < ns) { PROJ_STASH(buf ^ 1) }
        __syncthreads();
    }
#undef PROJ_FETCH
#undef PROJ_STASH

    // acc[mt][nt][r] = O[m = m0 + wm*64 + mt*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)][d = d0 + wn*64 + nt*32 + (lane&31)]
    const float inv_scale = 1.0f / scale;
    float* out = p.partial + ((long long)split * p.B + b) * p.k * p.D;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int d = d0 + wn * 64 + nt * 32 + (lane & 31);
                if (m < p.k && d < p.D) out[(long long)m * p.D + d] = acc[mt][nt][r] * inv_scale;
            }
}

__global__ __launch_bounds__(256) void proj_reduce_kernel(const float* __restrict__ partial, int nsplit, long long n,
                                                          float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int q = 0; q < nsplit; ++q) s += (double)partial[(long long)q * n + i];
    out[i] = (float)s;
}

int dm_project_f16split(dm_ctx* ctx, int B, int N, int D, int k, const float* Phi, int ld, const float* mass,
                        const void* F, float* Ared) {
    proj_params p;
    p.Phi = Phi; p.mass = mass; p.F = (const _Float16*)F;
    p.B = B; p.N = N; p.D = D; p.k = k; p.ld = ld;
    p.tiles_m = dm_cdiv(k, PT); p.tiles_d = dm_cdiv(D, PT);
    // split-K by a fixed chunk of vertices: the summation order of a pair must not depend on the batch it is in
    p.kchunk = 512;
    const int nsplit = dm_cdiv(N, p.kchunk);
    p.nsplit = nsplit;
    const size_t pbytes = (size_t)nsplit * B * k * D * 4;
    int rc = dm_ws_reserve(ctx, dm_align_up(pbytes) + 4096);
    if (rc) return rc;
    p.partial = (float*)dm_ws_take(ctx, pbytes);
    unsigned int* amax = (unsigned int*)dm_ws_take(ctx, (size_t)B * 4);
    p.amax = amax;
    DM_CHECK_HIP(ctx, hipMemsetAsync(amax, 0, (size_t)B * 4, ctx->stream));
    DM_LAUNCH(ctx, "project_absmax", proj_absmax_kernel, dim3(min(64, dm_cdiv(N, 16)), B), dim3(256), 0, Phi, mass, N, k, ld, amax);
    DM_LAUNCH(ctx, "project_f16split_mfma", proj_f16split_kernel, dim3(p.tiles_m * p.tiles_d, nsplit, B), dim3(256), 0, p);
    const long long n = (long long)B * k * D;
    DM_LAUNCH(ctx, "project_reduce", proj_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p.partial, nsplit, n,
              Ared);
    return DM_OK;
}

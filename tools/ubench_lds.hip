// What does one CU's LDS deliver, and what does the tile kernels' inner loop cost WITHOUT anything else around it?
//  (a) ds_read_b128 alone: W waves per CU, 12 reads per iteration (the fragment reads of one stage of simnn_pipe_kernel), conflict-free
//      addresses -> bytes per clock per CU;
//  (b) the stage of the one-key tile kernel reduced to its matrix instructions and fragment reads: 16 v_mfma_f32_32x32x16_f16 + 12
//      ds_read_b128 per wave and iteration, 8 waves per CU, no LDS-DMA, no barrier, no epilogue; operands zero (clock stays at 2.4 GHz)
//      or random (power-limited clock) -> matrix-pipe utilisation of the bare loop;
//  (c) (b) plus the LDS-DMA writes of a stage (4 x global_load_lds of 16 B per lane per wave) from an L2-resident buffer.
// Each line prints wall time per iteration (HIP events over 20 000 iterations on every CU), the same in 2.4 GHz clocks, bytes per clock
// and CU, and the matrix pipe's busy fraction against the 2.4 GHz issue rate (16 MFMA x 32 clocks x 2 waves per SIMD = 1024 clocks).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o tools/ubench_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: reads only; 1: reads + MFMA (reads pinned in front); 2: reads + MFMA + LDS-DMA; 3: MFMA only; 4: LDS-DMA only;
// 5: reads + MFMA + the same 4 KiB per wave as global_load_dwordx4 into registers, written with ds_write_b128 an iteration later;
// 8: as 2, but every wave issues its four LDS-DMA instructions between its matrix instructions, at positions that differ from
//    wave to wave (the eight waves' requests reach the address unit spread over the iteration instead of in two bursts);
// 6: those global loads alone (values kept alive, no LDS write); 7: global loads + ds_write, nothing else
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_lds(const _Float16* __restrict__ g, float* out, int iters, unsigned long long* clk, int fill) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];      // 128 KiB: four 32 KiB slots
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // (scalar: the LDS-DMA's M0 then never waits for the vector ALU)
    for (int i = t; i < 65536; i += blockDim.x) smem[i] = fill ? g[i] : (_Float16)0.f;
    __syncthreads();
    f32x16 acc[4][2];
    for (int a = 0; a < 4; ++a) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    f16x8 fs[4], ft[2], fs2[4], ft2[2];
    for (int x = 0; x < 4; ++x) { fs[x] = *reinterpret_cast<const f16x8*>(smem + (x * 64 + lane) * 8); fs2[x] = fs[x]; }
    for (int x = 0; x < 2; ++x) { ft[x] = *reinterpret_cast<const f16x8*>(smem + (256 + x * 64 + lane) * 8); ft2[x] = ft[x]; }
    const char* gb = reinterpret_cast<const char*>(g) + (size_t)blockIdx.x * 65536 + lane * 16;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) const u32x4 gu32x4;
    u32x4 st4[4] = {};
    constexpr bool RD = MODE == 0 || MODE == 1 || MODE == 2 || MODE == 5 || MODE == 8 || MODE == 11;
    constexpr bool MM = MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 8 || MODE == 10 || MODE == 12 || MODE == 13 || MODE == 14;
    constexpr bool DMA = MODE == 2 || MODE == 4 || MODE == 10 || MODE == 11;
    constexpr bool SPREAD = MODE == 8;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    constexpr bool GL = MODE == 5 || MODE == 6 || MODE == 7 || MODE == 12 || MODE == 13 || MODE == 14;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        const int slot = (i & 3) * 16384;                    // halves
        const _Float16* B0 = smem + slot + wave * 512 + lane * 8;
        if (GL) {
            if (MODE != 6 && MODE != 12) {
#pragma unroll
                for (int u = 0; u < 4; ++u) *reinterpret_cast<u32x4*>(smem + ((i + 2) & 3) * 16384 + wave * 2048 + u * 512 + lane * 8) = st4[u];
            } else {
                asm volatile("" :: "v"(st4[0]), "v"(st4[1]), "v"(st4[2]), "v"(st4[3]));     // (the loads are used: not dead code)
            }
            if (MODE != 13) {
#pragma unroll
                for (int u = 0; u < 4; ++u) st4[u] = *(gu32x4*)(gb + ((i * 4 + u) & 15) * 1024 + wave * 16384 % 49152);
            }
            asm volatile("" ::: "memory");
        }
        if (RD) {
#pragma unroll
            for (int x = 0; x < 4; ++x) fs2[x] = *reinterpret_cast<const f16x8*>(B0 + x * 1024);
#pragma unroll
            for (int x = 0; x < 2; ++x) ft2[x] = *reinterpret_cast<const f16x8*>(B0 + (4 + x) * 1024);
        }
        if (DMA) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                __builtin_amdgcn_global_load_lds((gptr_t)(gb + ((i * 4 + u) & 15) * 1024 + wave * 16384 % 49152), (lptr_t)(smem + ((i + 2) & 3) * 16384 + wave * 2048 + u * 512), 16, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MM) {
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    acc[st][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs[st], ft[tt], acc[st][tt], 0, 0, 0);
                    if (SPREAD) {
                        const int mi = st * 2 + tt;
                        if (mi == (wv & 7)) {
                            __builtin_amdgcn_global_load_lds((gptr_t)(gb + ((i * 4 + 0) & 15) * 1024 + wave * 16384 % 49152), (lptr_t)(smem + ((i + 2) & 3) * 16384 + wave * 2048 + 0 * 512), 16, 0, 0);
                        }
                        if (mi == ((wv + 4) & 7)) {
                            __builtin_amdgcn_global_load_lds((gptr_t)(gb + ((i * 4 + 1) & 15) * 1024 + wave * 16384 % 49152), (lptr_t)(smem + ((i + 2) & 3) * 16384 + wave * 2048 + 1 * 512), 16, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
        if (RD) {
#pragma unroll
            for (int x = 0; x < 4; ++x) fs[x] = *reinterpret_cast<const f16x8*>(B0 + (6 + x) * 1024);
#pragma unroll
            for (int x = 0; x < 2; ++x) ft[x] = *reinterpret_cast<const f16x8*>(B0 + (10 + x) * 1024);
        }
        if (DMA) {
#pragma unroll
            for (int u = 2; u < 4; ++u)
                __builtin_amdgcn_global_load_lds((gptr_t)(gb + ((i * 4 + u) & 15) * 1024 + wave * 16384 % 49152), (lptr_t)(smem + ((i + 2) & 3) * 16384 + wave * 2048 + u * 512), 16, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MM) {
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    acc[st][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs2[st], ft2[tt], acc[st][tt], 0, 0, 0);
                    if (SPREAD) {
                        const int mi = st * 2 + tt;
                        if (mi == (wv & 7)) {
                            __builtin_amdgcn_global_load_lds((gptr_t)(gb + ((i * 4 + 2) & 15) * 1024 + wave * 16384 % 49152), (lptr_t)(smem + ((i + 2) & 3) * 16384 + wave * 2048 + 2 * 512), 16, 0, 0);
                        }
                        if (mi == ((wv + 4) & 7)) {
                            __builtin_amdgcn_global_load_lds((gptr_t)(gb + ((i * 4 + 3) & 15) * 1024 + wave * 16384 % 49152), (lptr_t)(smem + ((i + 2) & 3) * 16384 + wave * 2048 + 3 * 512), 16, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        } else if (RD) {
            // keep the loaded values alive without VALU work worth mentioning
            asm volatile("" :: "v"(fs[0]), "v"(fs[1]), "v"(fs[2]), "v"(fs[3]), "v"(ft[0]), "v"(ft[1]));
            asm volatile("" :: "v"(fs2[0]), "v"(fs2[1]), "v"(fs2[2]), "v"(fs2[3]), "v"(ft2[0]), "v"(ft2[1]));
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int c = 0; c < 2; ++c) s += acc[a][c][(a + c) & 15];
    for (int x = 0; x < 4; ++x) s += (float)fs[x][0] + (float)fs2[x][1] + (float)st4[x][0];
    out[blockIdx.x * blockDim.x + t] = s + (float)ft[0][0] + (float)ft2[1][1];
    if (blockIdx.x == 0 && t == 0) clk[0] = t1 - t0;
}


// (h) wave specialisation: waves 0..7 run (b) -- 16 MFMA + 12 fragment reads per iteration, never a memory instruction -- and waves
// 8..11 (one more per SIMD) issue the LDS-DMA of all of them, 8 instructions each per iteration.  Free running, no barrier: the
// time is that of the slower role.  (Needs <= 168 VGPRs: three waves per SIMD.)
__global__ __launch_bounds__(768) void k_spec(const _Float16* __restrict__ g, float* out, int iters, unsigned long long* clk, int fill, int dma_per_iter) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int i = t; i < 65536; i += blockDim.x) smem[i] = fill ? g[i] : (_Float16)0.f;
    __syncthreads();
    if (wave >= 8) {
        const char* gb = reinterpret_cast<const char*>(g) + (size_t)blockIdx.x * 65536 + lane * 16;
        const int pw = wave - 8;
        for (int i = 0; i < iters; ++i) {
            for (int u = 0; u < dma_per_iter; ++u)
                __builtin_amdgcn_global_load_lds((gptr_t)(gb + ((i * 8 + u) & 15) * 1024 + pw * 16384 % 49152),
                                                 (lptr_t)(smem + ((i + 2) & 3) * 16384 + pw * 4096 + (u & 7) * 512), 16, 0, 0);
            if ((i & 7) == 7) __builtin_amdgcn_s_waitcnt(0x0070 | 8 | (0 << 14));     // vmcnt(8): a bounded number in flight, like a ring
        }
        __builtin_amdgcn_s_waitcnt(0);
        return;
    }
    // single set of fragments (24 registers): read, wait, eight matrix instructions; the SIMD's other consumer wave fills the
    // matrix pipe while this one waits for its reads
    f32x16 acc[4][2];
    for (int a = 0; a < 4; ++a) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    f16x8 fs[4], ft[2];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        const _Float16* B0 = smem + (i & 3) * 16384 + wave * 512 + lane * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int x = 0; x < 4; ++x) fs[x] = *reinterpret_cast<const f16x8*>(B0 + (6 * h + x) * 1024);
#pragma unroll
            for (int x = 0; x < 2; ++x) ft[x] = *reinterpret_cast<const f16x8*>(B0 + (6 * h + 4 + x) * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) acc[st][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs[st], ft[tt], acc[st][tt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int c = 0; c < 2; ++c) s += acc[a][c][(a + c) & 15];
    out[blockIdx.x * blockDim.x + t] = s;
    if (blockIdx.x == 0 && t == 0) clk[0] = t1 - t0;
}

static int run_spec(const _Float16* g, float* out, unsigned long long* clk, int fill, int ncu, int dma_per_iter) {
    const int iters = 20000;
    CK(hipFuncSetAttribute((const void*)k_spec, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_spec<<<ncu, 768, 131072>>>(g, out, 2000, clk, fill, dma_per_iter);
    CK(hipEventRecord(e0));
    k_spec<<<ncu, 768, 131072>>>(g, out, iters, clk, fill, dma_per_iter);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = 1e6 * ms / iters;
    printf("(h) 8 waves 16 MFMA + 12 ds_read, 4 waves %d LDS-DMA each  %-6s %8.1f ns/iter  = %7.1f clk @2.4GHz   delivered %5.1f B/clk/CU   matrix pipe busy %5.1f %% (of 2.4 GHz)\n",
           dma_per_iter, fill ? "random" : "zeros", ns, ns * 2.4, 4.0 * dma_per_iter * 1024 / (ns * 2.4), 100.0 * 16.0 * 32.0 * 2.0 / (ns * 2.4));
    return 0;
}

// (n) register-staged delivery with the fragment reads and the global loads kept APART in time: per iteration the wave writes the
// 4 KiB it loaded an iteration ago to the LDS, reads all 12 fragments, waits for them, requests the next 4 KiB, and only then issues
// its 16 matrix instructions (the SIMD's other wave fills the pipe meanwhile).  ORDER 1: the loads are requested BEFORE the reads.
template <int ORDER>
__global__ __launch_bounds__(512, 2) void k_apart(const _Float16* __restrict__ g, float* out, int iters, int fill) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int i = t; i < 65536; i += blockDim.x) smem[i] = fill ? g[i] : (_Float16)0.f;
    __syncthreads();
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) const u32x4 gu32x4;
    f32x16 acc[4][2];
    for (int a = 0; a < 4; ++a) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    f16x8 fs[4], ft[2], fs2[4], ft2[2];
    u32x4 st4[4] = {};
    const char* gb = reinterpret_cast<const char*>(g) + (size_t)blockIdx.x * 65536 + lane * 16;
    for (int i = 0; i < iters; ++i) {
        const _Float16* B0 = smem + (i & 3) * 16384 + wave * 512 + lane * 8;
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<u32x4*>(smem + ((i + 2) & 3) * 16384 + wave * 2048 + u * 512 + lane * 8) = st4[u];
        if (ORDER == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) st4[u] = *(gu32x4*)(gb + ((i * 4 + u) & 15) * 1024 + wave * 16384 % 49152);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) { fs[x] = *reinterpret_cast<const f16x8*>(B0 + x * 1024); fs2[x] = *reinterpret_cast<const f16x8*>(B0 + (6 + x) * 1024); }
#pragma unroll
        for (int x = 0; x < 2; ++x) { ft[x] = *reinterpret_cast<const f16x8*>(B0 + (4 + x) * 1024); ft2[x] = *reinterpret_cast<const f16x8*>(B0 + (10 + x) * 1024); }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): no fragment read in flight beyond this point
        __builtin_amdgcn_sched_barrier(0);
        if (ORDER == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) st4[u] = *(gu32x4*)(gb + ((i * 4 + u) & 15) * 1024 + wave * 16384 % 49152);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) acc[st][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs[st], ft[tt], acc[st][tt], 0, 0, 0);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) acc[st][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs2[st], ft2[tt], acc[st][tt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int c = 0; c < 2; ++c) s += acc[a][c][(a + c) & 15];
    out[blockIdx.x * blockDim.x + t] = s + (float)st4[0][0];
}
template <int ORDER>
static int run_apart(const _Float16* g, float* out, int fill, int ncu) {
    const int iters = 20000;
    CK(hipFuncSetAttribute((const void*)k_apart<ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_apart<ORDER><<<ncu, 512, 131072>>>(g, out, 2000, fill);
    CK(hipEventRecord(e0));
    k_apart<ORDER><<<ncu, 512, 131072>>>(g, out, iters, fill);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = 1e6 * ms / iters;
    printf("(n) write, %s, 16 MFMA   %-6s %8.1f ns/iter  = %7.1f clk @2.4GHz   delivered %5.1f B/clk/CU   matrix pipe busy %5.1f %% (of 2.4 GHz)\n",
           ORDER ? "4 loads, 12 reads + wait" : "12 reads + wait, 4 loads", fill ? "random" : "zeros", ns, ns * 2.4, 32768.0 / (ns * 2.4), 100.0 * 1024.0 / (ns * 2.4));
    return 0;
}

// (p), (q): VERDICT r04 #6 -- the TARGET fragments come straight from global memory into the registers the matrix instructions read (a
// lane's 16 bytes are contiguous in a row-major fp16 row: global_load_dwordx4, requested four iterations ahead into a ring of register
// sets), only the SOURCE operand travels through the LDS (half the LDS-DMA bytes of a stage).
//   LAYOUT 0: eight waves x (32 targets x 256 sources): per k-step 8 source fragments (ds_read) x 1 target fragment (global): the target
//             operand is read once per workgroup, the source fragments by all eight waves;
//   LAYOUT 1: the product kernel's 2 x 4 arrangement (4 source x 2 target fragments per k-step): every target fragment is requested by
//             four waves (32 KiB of global requests per stage for 16 KiB of data).
// Per wave and iteration (two k-steps): 16 matrix instructions, NRD = 16 / 8 ds_read_b128, 2 LDS-DMA, NGL = 2 / 4 global_load_dwordx4.
template <int LAYOUT, bool WITH_DMA, bool WITH_GL>
__global__ __launch_bounds__(512, 2) void k_direct(const _Float16* __restrict__ g, float* out, int iters, int fill) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int i = t; i < 65536; i += blockDim.x) smem[i] = fill ? g[i] : (_Float16)0.f;
    __syncthreads();
    constexpr int NS = LAYOUT == 0 ? 8 : 4, NT = LAYOUT == 0 ? 1 : 2;      // source / target fragments per k-step
    f32x16 acc[NS][NT];
    for (int a = 0; a < NS; ++a) for (int c = 0; c < NT; ++c) for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    typedef __attribute__((address_space(1))) const f16x8 gf16x8;
    const char* gb = reinterpret_cast<const char*>(g) + (size_t)blockIdx.x * 65536 + lane * 16;
    f16x8 ring[4][2][NT];                                       // [iteration mod 4][k-step][target fragment]
    for (int u = 0; u < 4; ++u) for (int h = 0; h < 2; ++h) for (int c = 0; c < NT; ++c)
        ring[u][h][c] = *(gf16x8*)(gb + ((u * 4 + h * 2 + c) & 15) * 1024 + wave * 16384 % 49152);
    f16x8 fs[NS];
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const _Float16* B0 = smem + ((i + u) & 3) * 16384 + wave * 512 + lane * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int x = 0; x < NS; ++x) fs[x] = *reinterpret_cast<const f16x8*>(B0 + ((h * NS + x) & 15) * 1024);
                if (WITH_DMA)
                    __builtin_amdgcn_global_load_lds((gptr_t)(gb + (((i + u) * 2 + h) & 15) * 1024 + wave * 16384 % 49152),
                                                     (lptr_t)(smem + ((i + u + 2) & 3) * 16384 + wave * 1024 + h * 512), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int st = 0; st < NS; ++st)
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) acc[st][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fs[st], ring[u][h][tt], acc[st][tt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (WITH_GL) {                                    // the same slot's fragments for four iterations from now
#pragma unroll
                    for (int c = 0; c < NT; ++c)
                        ring[u][h][c] = *(gf16x8*)(gb + ((((i + u + 4) * 4) + h * 2 + c) & 15) * 1024 + wave * 16384 % 49152);
                    asm volatile("" ::: "memory");
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    float s = 0.f;
    for (int a = 0; a < NS; ++a) for (int c = 0; c < NT; ++c) s += acc[a][c][(a + c) & 15];
    out[blockIdx.x * blockDim.x + t] = s + (float)fs[0][0] + (float)ring[0][0][0][0];
}
template <int LAYOUT, bool WITH_DMA, bool WITH_GL>
static int run_direct(const char* name, const _Float16* g, float* out, int fill, int ncu) {
    const int iters = 20000;
    CK(hipFuncSetAttribute((const void*)k_direct<LAYOUT, WITH_DMA, WITH_GL>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_direct<LAYOUT, WITH_DMA, WITH_GL><<<ncu, 512, 131072>>>(g, out, 2000, fill);
    CK(hipEventRecord(e0));
    k_direct<LAYOUT, WITH_DMA, WITH_GL><<<ncu, 512, 131072>>>(g, out, iters, fill);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = 1e6 * ms / iters;
    const int nrd = LAYOUT == 0 ? 16 : 8, ngl = WITH_GL ? (LAYOUT == 0 ? 2 : 4) : 0, ndma = WITH_DMA ? 2 : 0;
    printf("%-58s %-6s %8.1f ns/iter  = %7.1f clk @2.4GHz   LDS read %6.1f B/clk/CU  (LDS-DMA %5.1f, to registers %5.1f)   matrix pipe busy %5.1f %% (of 2.4 GHz)\n",
           name, fill ? "random" : "zeros", ns, ns * 2.4, 8.0 * nrd * 1024 / (ns * 2.4), 8.0 * ndma * 1024 / (ns * 2.4), 8.0 * ngl * 1024 / (ns * 2.4),
           100.0 * 1024.0 / (ns * 2.4));
    return 0;
}

template <int MODE>
static int run(const char* name, const _Float16* g, float* out, unsigned long long* clk, int waves, int fill, int ncu) {
    const int iters = 20000;
    CK(hipFuncSetAttribute((const void*)k_lds<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_lds<MODE><<<ncu, waves * 64, 131072>>>(g, out, 2000, clk, fill);
    CK(hipEventRecord(e0));
    k_lds<MODE><<<ncu, waves * 64, 131072>>>(g, out, iters, clk, fill);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2];
    CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double cyc = (double)h[0] / iters;                 // s_memtime ticks (100 MHz reference scaled? see clock column)
    const double ns = 1e6 * ms / iters;
    const bool rd = MODE == 0 || MODE == 1 || MODE == 2 || MODE == 5 || MODE == 8 || MODE == 11, mm = MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 8 || MODE == 10 || MODE == 12 || MODE == 13 || MODE == 14;
    const double rd_bytes = rd ? (double)waves * 12 * 1024 : 0.0, dma_bytes = (MODE == 2 || MODE >= 4) ? (double)waves * 4 * 1024 : 0.0;
    const double mfma_cyc = mm ? 16.0 * 32.0 * waves / 4.0 : 0.0;            // matrix-pipe cycles per SIMD and iteration
    printf("%-44s waves %2d  %-6s %8.1f ns/iter  = %7.1f clk @2.4GHz   LDS read %6.1f B/clk/CU  (delivered %5.1f)   matrix pipe busy %5.1f %% (of 2.4 GHz)\n",
           name, waves, fill ? "random" : "zeros", ns, ns * 2.4, rd_bytes / (ns * 2.4), dma_bytes / (ns * 2.4), 100.0 * mfma_cyc / (ns * 2.4));
    (void)cyc;
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    _Float16* g; float* out; unsigned long long* clk;
    const size_t gbytes = (size_t)ncu * 65536 + (1 << 20);
    CK(hipMalloc(&g, gbytes)); CK(hipMalloc(&out, (size_t)ncu * 1024 * 4)); CK(hipMalloc(&clk, 16));
    _Float16* h = (_Float16*)malloc(gbytes);
    srand(1);
    for (size_t i = 0; i < gbytes / 2; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.0f);
    CK(hipMemcpy(g, h, gbytes, hipMemcpyHostToDevice));
    for (int waves : {4, 8}) {
        if (run<0>("(a) 12 ds_read_b128 per wave and iteration", g, out, clk, waves, 1, ncu)) return 1;
    }
    for (int fill : {0, 1}) {
        if (run<3>("    16 MFMA per wave and iteration, no reads", g, out, clk, 8, fill, ncu)) return 1;
        if (run<1>("(b) 16 MFMA + 12 ds_read_b128", g, out, clk, 8, fill, ncu)) return 1;
        if (run<2>("(c) 16 MFMA + 12 ds_read_b128 + 4 LDS-DMA", g, out, clk, 8, fill, ncu)) return 1;
        if (run<8>("(c') as (c), DMA spread between the MFMAs per wave", g, out, clk, 8, fill, ncu)) return 1;
        if (run<5>("(d) 16 MFMA + 12 ds_read + 4 global_load x4 + 4 ds_write", g, out, clk, 8, fill, ncu)) return 1;
    }
    for (int fill : {0, 1}) { if (run_apart<0>(g, out, fill, ncu)) return 1; if (run_apart<1>(g, out, fill, ncu)) return 1; }
    if (run<10>("(i) 16 MFMA + 4 LDS-DMA (no fragment reads)", g, out, clk, 8, 0, ncu)) return 1;
    if (run<12>("(k) 16 MFMA + 4 global_load_dwordx4 to registers", g, out, clk, 8, 0, ncu)) return 1;
    if (run<14>("(m) 16 MFMA + 4 global_load_dwordx4 + 4 ds_write_b128", g, out, clk, 8, 0, ncu)) return 1;
    if (run<13>("(l) 16 MFMA + 4 ds_write_b128 (no memory instr.)", g, out, clk, 8, 0, ncu)) return 1;
    if (run<11>("(j) 12 ds_read_b128 + 4 LDS-DMA (no MFMA)", g, out, clk, 8, 0, ncu)) return 1;
    for (int fill : {0, 1}) for (int d : {0, 8, 12}) if (run_spec(g, out, clk, fill, ncu, d)) return 1;
    for (int fill : {0, 1}) {
        if (run_direct<0, false, false>("(p0) 8x1 waves: 16 MFMA + 16 ds_read (no delivery)", g, out, fill, ncu)) return 1;
        if (run_direct<0, false, true>("(p1) 8x1: 16 MFMA + 16 ds_read + 2 global->VGPR fragments", g, out, fill, ncu)) return 1;
        if (run_direct<0, true, false>("(p2) 8x1: 16 MFMA + 16 ds_read + 2 LDS-DMA", g, out, fill, ncu)) return 1;
        if (run_direct<0, true, true>("(p)  8x1: 16 MFMA + 16 ds_read + 2 LDS-DMA + 2 global->VGPR", g, out, fill, ncu)) return 1;
        if (run_direct<1, false, true>("(q1) 2x4: 16 MFMA + 8 ds_read + 4 global->VGPR fragments", g, out, fill, ncu)) return 1;
        if (run_direct<1, true, true>("(q)  2x4: 16 MFMA + 8 ds_read + 2 LDS-DMA + 4 global->VGPR", g, out, fill, ncu)) return 1;
    }
    if (run<4>("(e) 4 LDS-DMA per wave and iteration alone", g, out, clk, 8, 1, ncu)) return 1;
    if (run<4>("(e) 4 LDS-DMA per wave and iteration alone", g, out, clk, 4, 1, ncu)) return 1;
    if (run<6>("(f) 4 global_load_dwordx4 alone (to registers)", g, out, clk, 8, 1, ncu)) return 1;
    if (run<7>("(g) 4 global_load_dwordx4 + 4 ds_write_b128", g, out, clk, 8, 1, ncu)) return 1;
    return 0;
}

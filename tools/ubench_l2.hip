// L2 -> CU delivery rate probes on the GPU box: one 512-thread workgroup per CU streams 64 KiB pieces of an
// L2-resident 2 MiB buffer, either into VGPRs (global_load_dwordx4) or into LDS by LDS-DMA (global_load_lds, 16 B/lane).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_l2.hip -o tools/ubench_l2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(512) void k_vgpr(const u32x4* __restrict__ buf, unsigned int* out, int iters, int nchunk) {
    extern __shared__ char smem[];
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const u32x4* src = buf + (size_t)((blockIdx.x * 5 + it) % nchunk) * 4096;      // 64 KiB pieces (4096 x 16 B)
        u32x4 r[DEPTH];
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) r[q] = src[(q % 8) * 512 + threadIdx.x];
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) acc ^= r[q];
    }
    if (acc[0] == 0x12345678u) out[0] = acc[1] + acc[2] + acc[3] + (unsigned)(size_t)smem;
}
// DEPTH stages of 64 KiB in flight (ring of DEPTH+1 slots of 64 KiB would not fit: slots are 32 KiB x 4 when DEPTH > 1)
template <int PIECES, int INFLIGHT>
__global__ __launch_bounds__(512) void k_dma(const u32x4* __restrict__ buf, unsigned int* out, int iters, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        const u32x4* src = buf + (size_t)((blockIdx.x * 5 + it) % nchunk) * 4096;
        char* dst = smem + (it & 1) * 65536;
#pragma unroll
        for (int q = 0; q < PIECES; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (wave * PIECES + q) * 64 + lane), (lptr_t)(dst + (wave * PIECES + q) * 1024), 16, 0, 0);
        if (INFLIGHT == 0) __builtin_amdgcn_s_waitcnt(0x0F70);
        else __builtin_amdgcn_s_waitcnt(0x0F70 | PIECES);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (smem[threadIdx.x] == 0x7f && smem[threadIdx.x + 1] == 0x3e && smem[threadIdx.x + 2] == 0x11) out[0] = 1;
}
// The similarity kernel's pattern: per stage 2 x 256 rows x 128 B at row stride RS bytes (RS = 1536 for D = 768), column
// chunk (it % ncol) of panel ((blockIdx + it / ncol) % npanel); 8 DMA pieces per wave, each 8 rows x 128 B.
template <int MODE>   // bit 0: drain + barrier per stage; bit 1: XOR-swizzled chunk order inside each 128-B line; bit 2: per-XCD panel sets
__global__ __launch_bounds__(512) void k_dma_rows(const char* __restrict__ buf, unsigned int* out, int iters, int RS, int ncol, int npanel) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        const int col = it % ncol;
        char* dst = smem + (it & 1) * 65536;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int piece = wave * 8 + q;                     // 0..63: 32 pieces per operand
            int panel = (blockIdx.x + it / ncol + (piece >> 5)) % npanel;
            if (MODE & 4) panel = (blockIdx.x & 7) * npanel + ((blockIdx.x >> 3) + it / ncol + (piece >> 5) * 5) % npanel;
            const int row = (piece & 31) * 8 + (lane >> 3);
            const int chunk = (MODE & 2) ? ((lane & 7) ^ ((row >> 1) & 7)) : (lane & 7);
            const char* src = buf + (size_t)panel * 256 * RS + (size_t)row * RS + col * 128 + chunk * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + piece * 1024), 16, 0, 0);
        }
        if (MODE & 1) { __builtin_amdgcn_s_waitcnt(0x0F70); __builtin_amdgcn_s_barrier(); }
        else __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (smem[threadIdx.x] == 0x7f && smem[threadIdx.x + 1] == 0x3e && smem[threadIdx.x + 2] == 0x11) out[0] = 1;
}
int main() {
    const int nchunk = 32;                       // 2 MiB
    void* buf; CK(hipMalloc(&buf, (size_t)nchunk * 65536)); CK(hipMemset(buf, 1, (size_t)nchunk * 65536));
    unsigned int* out; CK(hipMalloc(&out, 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    float ms;
    CK(hipFuncSetAttribute((const void*)k_vgpr<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)k_vgpr<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)k_dma<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)k_dma<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
#define RUN(name, kern, G, lds, bytes_per_it)                                                                        \
    for (int rep = 0; rep < 2; ++rep) {                                                                              \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(G), dim3(512), lds, 0, (const u32x4*)buf, out, iters, nchunk); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));                   \
        if (rep) printf("%-44s: %.2f TB/s aggregate, %.1f B/clk/CU at 2.4 GHz\n", name,                              \
                        (double)(G) * iters * (bytes_per_it) / (ms * 1e-3) / 1e12,                                    \
                        (double)(G) * iters * (bytes_per_it) / (ms * 1e-3) / 256 / 2.4e9);                            \
    }
    RUN("vgpr dwordx4, 8 loads in flight, 1 WG/CU", (k_vgpr<8>), 256, 131072, 65536.0)
    RUN("vgpr dwordx4, 16 loads in flight, 1 WG/CU", (k_vgpr<16>), 256, 131072, 131072.0)
    RUN("vgpr dwordx4, 8 loads in flight, 2 WG/CU", (k_vgpr<8>), 512, 65536, 65536.0)
    RUN("vgpr dwordx4, 8 loads in flight, 4 WG/CU", (k_vgpr<8>), 1024, 32768, 65536.0)
    RUN("lds-dma 16 B, drain per 64 KiB, 1 WG/CU", (k_dma<8, 0>), 256, 131072, 65536.0)
    RUN("lds-dma 16 B, one 64 KiB stage in flight, 1 WG/CU", (k_dma<8, 1>), 256, 131072, 65536.0)
    void* big; CK(hipMalloc(&big, (size_t)256 << 20)); CK(hipMemset(big, 1, (size_t)256 << 20));
#define ROWS(MODE, npanel, label)                                                                                     \
    CK(hipFuncSetAttribute((const void*)k_dma_rows<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));       \
    for (int rep = 0; rep < 2; ++rep) {                                                                               \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_dma_rows<MODE>, dim3(256), dim3(512), 131072, 0, (const char*)big, out, iters, 1536, 12, npanel); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));                    \
        if (rep) printf("lds-dma rows stride 1536, %-58s: %.2f TB/s aggregate\n", label, 256.0 * iters * 65536.0 / (ms * 1e-3) / 1e12); \
    }
    ROWS(0, 4, "4 shared panels, 1 stage in flight")
    ROWS(1, 4, "4 shared panels, drain+barrier per stage")
    ROWS(2, 4, "4 shared panels, swizzled, 1 stage in flight")
    ROWS(3, 4, "4 shared panels, swizzled, drain+barrier")
    ROWS(4, 12, "12 panels per XCD (4.7 MB), 1 stage in flight")
    ROWS(5, 12, "12 panels per XCD (4.7 MB), drain+barrier")
    ROWS(7, 12, "12 panels per XCD (4.7 MB), swizzled, drain+barrier")
    ROWS(7, 8, "8 panels per XCD (3.1 MB), swizzled, drain+barrier")
    ROWS(7, 16, "16 panels per XCD (6.3 MB), swizzled, drain+barrier")
    return 0;
}

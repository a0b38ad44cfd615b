// What does an LDS operand read per float64 matrix instruction cost?  Every wave runs a loop of NM v_mfma_f64_16x16x4_f64 with
// NR ds_read_b64 feeding their B operands (the shape of the LDS-staged float64 products: gram_nt_f64, embed_nt_f64,
// zo_embed_split), WPS waves per SIMD, no barriers, no global traffic.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_f64_lds.hip -o tools/ubench_f64_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// NR reads per group of 4 matrix instructions (NR = 0: operands stay in registers)
template <int NR, int B128>
__global__ __launch_bounds__(1024) void k_lds(double* out, int iters) {
    __shared__ double sm[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = 1.0 + 1e-6 * i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3;
    double b0 = 1.0, b1 = 1.0, b2 = 1.0, b3 = 1.0;
    const double* base = sm + (lane & 15) * 18 + (lane >> 4);
    for (int i = 0; i < iters; ++i) {
        const double* p = base + ((i & 7) * 288);
        if (B128) {
            if (NR >= 2) { const double2 v = *reinterpret_cast<const double2*>(p - ((lane >> 4) & 1)); b0 = v.x; b1 = v.y; }
            if (NR >= 4) { const double2 v = *reinterpret_cast<const double2*>(p + 16 - ((lane >> 4) & 1)); b2 = v.x; b3 = v.y; }
        } else {
            if (NR >= 1) b0 = p[0];
            if (NR >= 2) b1 = p[4];
            if (NR >= 3) b2 = p[8];
            if (NR >= 4) b3 = p[12];
        }
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, b0, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, b1, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, b2, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, b3, a3, 0, 0, 0);
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main() {
    double* out; CK(hipMalloc(&out, 512 * 1024 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    float ms;
#define RUN(NR_, B128_, THREADS, NAME)                                                                              \
    for (int rep = 0; rep < 2; ++rep) {                                                                             \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_lds<NR_, B128_>), dim3(256), dim3(THREADS), 0, 0, out, iters); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));                  \
        if (rep) printf("%-46s %d waves per SIMD: %.3f ms, %.1f TFLOP/s, %.1f ns per matrix instruction per SIMD\n", NAME, THREADS / 256, ms, \
                        256.0 * (THREADS / 64) * iters * 4 * 2048.0 / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * 4.0 * (THREADS / 256))); \
    }
    RUN(0, 0, 256, "no LDS reads") RUN(0, 0, 512, "no LDS reads") RUN(0, 0, 1024, "no LDS reads")
    RUN(2, 0, 256, "2 ds_read_b64 per 4 (0.5 per instruction)") RUN(2, 0, 512, "2 ds_read_b64 per 4") RUN(2, 0, 1024, "2 ds_read_b64 per 4")
    RUN(4, 0, 256, "4 ds_read_b64 per 4 (1 per instruction)") RUN(4, 0, 512, "4 ds_read_b64 per 4") RUN(4, 0, 1024, "4 ds_read_b64 per 4")
    RUN(4, 1, 256, "2 ds_read_b128 per 4 (same operands)") RUN(4, 1, 512, "2 ds_read_b128 per 4") RUN(4, 1, 1024, "2 ds_read_b128 per 4")
    return 0;
}

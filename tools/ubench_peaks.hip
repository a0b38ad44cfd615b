// Peak probes on the GPU box: MFMA issue rates (f64 16x16x4, f32 32x32x2, f16 32x32x16) and a
// float4 stream copy.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_peaks.hip -o tools/ubench_peaks
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ITER> __global__ __launch_bounds__(256) void k_f64(double* out) {
    f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < ITER; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
template <int ITER> __global__ __launch_bounds__(256) void k_f32(float* out) {
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) a0[r] = a1[r] = a2[r] = a3[r] = 0.f;
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < ITER; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
template <int ITER> __global__ __launch_bounds__(256) void k_f16(float* out) {
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) a0[r] = a1[r] = a2[r] = a3[r] = 0.f;
    f16x8 x, y;
    for (int r = 0; r < 8; ++r) { x[r] = (_Float16)(threadIdx.x * 1e-3f + r); y[r] = (_Float16)(1.f + r * 0.01f); }
    for (int i = 0; i < ITER; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
int main() {
    void* buf; CK(hipMalloc(&buf, 64 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ITER = 4096, G = 256 * 8;
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_f64<ITER>, dim3(G), dim3(256), 0, 0, (double*)buf); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("mfma_f64_16x16x4 : %.1f TFLOP/s\n", (double)G * 4 * ITER * 4 * 2048.0 / (ms * 1e-3) / 1e12);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_f32<ITER>, dim3(G), dim3(256), 0, 0, (float*)buf); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("mfma_f32_32x32x2  : %.1f TFLOP/s\n", (double)G * 4 * ITER * 4 * 4096.0 / (ms * 1e-3) / 1e12);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_f16<ITER>, dim3(G), dim3(256), 0, 0, (float*)buf); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("mfma_f16_32x32x16 : %.1f TFLOP/s\n", (double)G * 4 * ITER * 4 * 32768.0 / (ms * 1e-3) / 1e12);
    }
    // one wave per SIMD (256 blocks x 256 threads), 4 independent accumulator chains: what a latency-bound kernel sees
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_f64<ITER>, dim3(256), dim3(256), 0, 0, (double*)buf); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("mfma_f64_16x16x4, 1 wave/SIMD x 4 chains: %.1f TFLOP/s  (%.1f cycles/MFMA at 2.4 GHz)\n", 256.0 * 4 * ITER * 4 * 2048.0 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (ITER * 4.0));
    }
    size_t nbytes = (size_t)2 << 30; void *a, *b; CK(hipMalloc(&a, nbytes)); CK(hipMalloc(&b, nbytes));
    CK(hipMemset(a, 1, nbytes));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, (const float4*)a, (float4*)b, nbytes / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("float4 copy       : %.0f GB/s (read+write)\n", 2.0 * nbytes / (ms * 1e-3) / 1e9);
    }
    return 0;
}

// Dependent-issue latency of v_fma_f64 / v_add_f64 / v_fma_f32: one wave per SIMD, NCH independent chains per wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu_f64_lat.hip -o tools/ubench_valu_f64_lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int NCH, int KIND>
__global__ void k_lat(double* out, int iters) {
    double d[NCH]; float f[NCH];
    for (int q = 0; q < NCH; ++q) { d[q] = 1.0 + 1e-3 * (threadIdx.x + 64 * q); f[q] = (float)d[q]; }
    const double c = 1.0000001;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32 / NCH; ++u)
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[q]) : "v"(c));
                if (KIND == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[q]) : "v"(c));
                if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[q]));
            }
    }
    double t = 0;
    for (int q = 0; q < NCH; ++q) t += d[q] + f[q];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = t;
}
int main() {
    double* out; CK(hipMalloc(&out, 256 * 1024 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000; float ms;
    const char* names[] = {"v_fma_f64", "v_add_f64", "v_fma_f32"};
#define RUN(N, K, WPS)                                                                                              \
    for (int rep = 0; rep < 2; ++rep) {                                                                             \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_lat<N, K>), dim3(256), dim3(256 * WPS), 0, 0, out, iters);     \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));                  \
        if (rep) printf("%-10s %d chain(s), %d wave(s)/SIMD: %.3f ms -> %.1f clocks per instruction per wave (2.4 GHz)\n", names[K], N, WPS, ms, ms * 1e6 / (iters * 32.0) * 2.4); \
    }
    RUN(1, 0, 1) RUN(2, 0, 1) RUN(4, 0, 1) RUN(8, 0, 1) RUN(1, 0, 2) RUN(2, 0, 2) RUN(1, 0, 4) RUN(2, 0, 4) RUN(2, 0, 3)
    RUN(1, 1, 1) RUN(2, 1, 1) RUN(4, 1, 1) RUN(1, 2, 1) RUN(2, 2, 1) RUN(4, 2, 1)
    return 0;
}

// Why does the fp16 MFMA probe of tools/ubench_peaks.hip read 1.74 PF when MI355X_MICROARCH.md records 2.5 PF for the same
// instruction?  Same loop (v_mfma_f32_32x32x16_f16, independent accumulator chains), varied along the axes that could
// explain it: operand DATA (zeros / constant / random: switching activity -> power -> clock), number of independent
// chains per wave, waves per SIMD, and burst length.  Each line prints the rate and the shader clock during the kernel
// (s_memtime ticks / s_memrealtime ticks x 100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_clock.hip -o tools/ubench_mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int CH>
__global__ __launch_bounds__(256) void k_f16(const f16x8* __restrict__ in, float* out, int iters, unsigned long long* clk) {
    f32x16 a[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) a[c][r] = 0.f;
    const f16x8 x = in[threadIdx.x], y = in[256 + threadIdx.x];
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) a[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += a[c][c & 15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int CH>
__global__ __launch_bounds__(256) void k_f64(const double* __restrict__ in, double* out, int iters, unsigned long long* clk) {
    f64x4 a[CH];
    for (int c = 0; c < CH; ++c) a[c] = f64x4{0, 0, 0, 0};
    const double x = in[threadIdx.x], y = in[256 + threadIdx.x];
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) a[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0.0;
    for (int c = 0; c < CH; ++c) s += a[c][c & 3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    void *in, *out; unsigned long long* clk;
    CK(hipMalloc(&in, 1 << 16)); CK(hipMalloc(&out, 64 << 20)); CK(hipMalloc((void**)&clk, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    _Float16 h[512 * 8]; double hd[512];
    const char* dname[3] = {"zeros", "constant 1.0", "random N(0,1)"};
    for (int data = 0; data < 3; ++data) {
        srand(1);
        for (int i = 0; i < 512 * 8; ++i) {
            float u = 0.f;
            if (data == 1) u = 1.f;
            if (data == 2) { float s = 0; for (int q = 0; q < 12; ++q) s += rand() / (float)RAND_MAX; u = s - 6.f; }
            h[i] = (_Float16)u;
            if (i < 512) hd[i] = u;
        }
        CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
        for (int wps = 1; wps <= 8; wps *= 2) {                 // waves per SIMD (blocks of 4 waves, one per SIMD)
            for (int iters : {2048, 65536}) {
#define RUN16(CH)                                                                                                        \
                {                                                                                                        \
                    float ms = 0; unsigned long long hc[2];                                                              \
                    for (int rep = 0; rep < 2; ++rep) {                                                                  \
                        CK(hipEventRecord(e0));                                                                          \
                        hipLaunchKernelGGL(k_f16<CH>, dim3(256 * wps), dim3(256), 0, 0, (const f16x8*)in, (float*)out, iters, clk); \
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));       \
                    }                                                                                                    \
                    CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));                                                   \
                    printf("f16 32x32x16  data %-14s chains %d  waves/SIMD %d  iters %6d : %7.1f TFLOP/s   %.0f MHz  (%.2f ms)\n", dname[data], CH, wps, iters, \
                           256.0 * wps * 4 * iters * CH * 32768.0 / (ms * 1e-3) / 1e12, hc[1] ? 100.0 * hc[0] / hc[1] : 0.0, ms); \
                }
                if (wps == 1 || wps == 2) { RUN16(1) RUN16(2) }
                RUN16(4)
                if (wps <= 2) RUN16(8)
            }
        }
    }
    for (int data = 0; data < 3; data += 2) {
        for (int i = 0; i < 512; ++i) hd[i] = data ? (double)h[i] : 0.0;
        CK(hipMemcpy(in, hd, sizeof(hd), hipMemcpyHostToDevice));
        for (int wps = 1; wps <= 8; wps *= 2) {
            float ms = 0; unsigned long long hc[2];
            const int iters = 32768;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_f64<4>, dim3(256 * wps), dim3(256), 0, 0, (const double*)in, (double*)out, iters, clk);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
            printf("f64 16x16x4   data %-14s chains 4  waves/SIMD %d  iters %6d : %7.1f TFLOP/s   %.0f MHz  (%.2f ms)\n", data ? dname[2] : dname[0], wps, iters,
                   256.0 * wps * 4 * iters * 4 * 2048.0 / (ms * 1e-3) / 1e12, hc[1] ? 100.0 * hc[0] / hc[1] : 0.0, ms);
        }
    }
    return 0;
}

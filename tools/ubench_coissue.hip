// Does VALU work from a second wave on the same SIMD slow the f64 MFMA pipe?  512-thread workgroups (2 waves per SIMD):
// waves 0-3 issue v_mfma_f64_16x16x4_f64 back to back, waves 4-7 run a loop of one VALU flavour.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_coissue.hip -o tools/ubench_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(512) void k_co(double* out, int iters, int valu_iters) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
        float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
        int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
        for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (KIND == 1) { d0 = fma(d0, 1.0000001, 0.5); d1 = fma(d1, 1.0000001, 0.5); d2 = fma(d2, 1.0000001, 0.5); d3 = fma(d3, 1.0000001, 0.5); }
                if (KIND == 2) { d0 = fmax(d0, d1 + 0.0); asm volatile("" : "+v"(d0)); d1 = fmax(d1, d2); asm volatile("" : "+v"(d1)); d2 = fmax(d2, d3); asm volatile("" : "+v"(d2)); d3 = fmax(d3, d0); asm volatile("" : "+v"(d3)); }
                if (KIND == 3) { f0 = fmaf(f0, 1.0001f, 0.5f); f1 = fmaf(f1, 1.0001f, 0.5f); f2 = fmaf(f2, 1.0001f, 0.5f); f3 = fmaf(f3, 1.0001f, 0.5f); }
                if (KIND == 4) { i0 = min(i0 + 3, i1); asm volatile("" : "+v"(i0)); i1 = min(i1 + 5, i2); asm volatile("" : "+v"(i1)); i2 = min(i2 + 7, i3); asm volatile("" : "+v"(i2)); i3 = min(i3 + 1, i0); asm volatile("" : "+v"(i3)); }
                if (KIND == 5) { i0 = __builtin_amdgcn_update_dpp(i0, i1, 0xB1, 0xF, 0xF, false); i1 = __builtin_amdgcn_update_dpp(i1, i2, 0x4E, 0xF, 0xF, false); i2 = __builtin_amdgcn_update_dpp(i2, i3, 0x141, 0xF, 0xF, false); i3 = __builtin_amdgcn_update_dpp(i3, i0, 0x140, 0xF, 0xF, false); }
                if (KIND == 6) { i0 = (d0 > d1) ? i1 : i0; asm volatile("" : "+v"(i0)); i1 = (d1 > d2) ? i2 : i1; asm volatile("" : "+v"(i1)); i2 = (d2 > d3) ? i3 : i2; asm volatile("" : "+v"(i2)); i3 = (d3 > d0) ? i0 : i3; asm volatile("" : "+v"(i3)); }
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = d0 + d1 + d2 + d3 + f0 + f1 + f2 + f3 + i0 + i1 + i2 + i3;
    }
}
int main() {
    double* out; CK(hipMalloc(&out, 512 * 512 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    float ms;
    const char* names[] = {"MFMA waves alone (VALU waves idle)", "v_fma_f64", "v_max_f64", "v_fma_f32", "v_add+v_min i32", "v_mov_dpp b32", "v_cmp_gt_f64 + v_cndmask"};
#define RUN(K, vit)                                                                                             \
    for (int rep = 0; rep < 2; ++rep) {                                                                         \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_co<K>, dim3(256), dim3(512), 0, 0, out, iters, vit);       \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));              \
        if (rep) printf("%-40s valu_iters %6d: %.3f ms, MFMA rate %.1f TFLOP/s (if MFMA-bound)\n", names[K], vit, ms, \
                        256.0 * 4 * iters * 4 * 2048.0 / (ms * 1e-3) / 1e12);                                   \
    }
    RUN(0, 0)
    // VALU loops sized to finish well before the MFMA waves when run alone: 32 ops per iteration
    RUN(1, 0) RUN(1, 20000) RUN(2, 20000) RUN(3, 20000) RUN(4, 20000) RUN(5, 20000) RUN(6, 20000)
    RUN(1, 40000) RUN(2, 40000) RUN(3, 40000) RUN(4, 40000) RUN(5, 40000) RUN(6, 40000)
    return 0;
}

// Issue rate of the float64 vector-ALU instructions the fused fit's element loop is made of (dm_fitfuse.hip): per wave-instruction,
// with 1, 2 and 4 waves per SIMD and eight independent chains per wave, on all 256 CUs.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu_f64.hip -o tools/ubench_valu_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void k_valu(double* out, int iters, double sarg) {
    double d[8];
    for (int q = 0; q < 8; ++q) d[q] = 1.0 + 1e-3 * (threadIdx.x + 64 * q);
    const double s = sarg;                      // (kernel argument: an SGPR operand)
    int e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (KIND == 0) d[q] = fma(d[q], d[(q + 1) & 7], d[(q + 2) & 7]);
                if (KIND == 1) d[q] = fma(d[q], s, d[q]);
                if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[q]) : "v"(d[(q + 1) & 7]));
                if (KIND == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[q]) : "v"(d[(q + 1) & 7]));
                if (KIND == 4) asm volatile("v_max_f64 %0, %0, %0 clamp" : "+v"(d[q]));
                if (KIND == 5) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[q]));
                if (KIND == 6) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(d[q]));
                if (KIND == 7) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(e[q]) : "v"(d[q]));
                if (KIND == 8) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(e[q]) : "v"(d[q]), "v"(d[(q + 1) & 7]), "v"(e[(q + 1) & 7]) : "vcc");
                if (KIND == 9) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[q]) : "v"(e[q]));
                if (KIND == 10) { float f = (float)d[q]; asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f)); d[q] = f; }
                if (KIND == 11) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[q]) : "v"(d[(q + 1) & 7]));
            }
    }
    double t = 0;
    for (int q = 0; q < 8; ++q) t += d[q] + e[q];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = t;
}

int main() {
    double* out; CK(hipMalloc(&out, 256 * 4 * 1024 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    const char* names[] = {"v_fma_f64 (vgpr x3, compiler)", "v_fma_f64 (one sgpr operand)", "v_add_f64", "v_mul_f64", "v_max_f64 clamp", "v_rcp_f64",
                           "v_frexp_mant_f64", "v_frexp_exp_i32_f64", "v_cmp_gt_f64 + v_cndmask_b32", "v_cvt_f64_i32", "cvt + v_fma_f32 + cvt (3 instr)",
                           "v_fma_f64 (asm)"};
    float ms;
#define RUN(K)                                                                                                                  \
    for (int wps = 1; wps <= 4; wps *= 2) {                                                                                     \
        for (int rep = 0; rep < 2; ++rep) {                                                                                     \
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_valu<K>, dim3(256), dim3(256 * wps), 0, 0, out, iters, 1.0000001);    \
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));                          \
        }                                                                                                                       \
        const double n_instr = (double)iters * 32 * wps * (K == 8 ? 2 : (K == 10 ? 3 : 1));                                     \
        printf("%-34s %d wave(s)/SIMD: %.3f ms -> %.2f ns per wave-instruction per SIMD = %.1f clocks at 2.4 GHz\n", names[K], wps, ms, \
               ms * 1e6 / n_instr, ms * 1e6 / n_instr * 2.4);                                                                   \
    }
    RUN(0) RUN(11) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10)
    return 0;
}

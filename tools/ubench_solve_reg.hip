// Phase timing of the register-resident solver (densematcher_amd/csrc/dm_chol_reg.h): one wave per system; cycle counts
// (s_memtime) of  [1] diagonal-block chains  [2] panels  [3] trailing updates  [4] forward  [5] backward substitution, for a
// launch that fills the GPU (4 waves per CU x rounds) and for a single wave.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDMREG_TIMING -I densematcher_amd/csrc -I include tools/ubench_solve_reg.hip -o /tmp/usr
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__device__ long long g_dmreg_t[8];
#include "dm_chol_reg.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NB>
__global__ __launch_bounds__(256, 1) void k_solve(const double* __restrict__ img, const double* __restrict__ rhsv, double* __restrict__ x, int nsys) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sys = blockIdx.x * 4 + wave;
    if (sys >= nsys) return;
    const int c = lane & 15, g = lane >> 4;
    constexpr int NBLK = NB * (NB + 1) / 2;
    f64x4 T[NBLK];
#pragma unroll
    for (int q = 0; q < NBLK; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[q][r] = img[((long long)(sys & 63) * NBLK + q) * 256 + 64 * r + lane];
    const double* rv = rhsv + (long long)(sys & 63) * NB * 16;
    double* xo = x + (long long)sys * NB * 16;
    __shared__ double sh_rhs[4][NB * 16];                   // the right-hand side waits in LDS (fetched with the image)
    for (int q = lane; q < NB * 16; q += 64) sh_rhs[wave][q] = rv[q];
    auto rhs = [&](int J) {
        f64x4 v = {0.0, 0.0, 0.0, 0.0};
        if (c == 0) { for (int r = 0; r < 4; ++r) v[r] = sh_rhs[wave][J * 16 + g + 4 * r]; }
        return v;
    };
    auto store = [&](int J, const f64x4& v) {
        if (c == 0) { for (int r = 0; r < 4; ++r) xo[J * 16 + g + 4 * r] = v[r]; }
    };
#ifndef UB_EV
#define UB_EV 0
#endif
    extern __shared__ __attribute__((aligned(16))) double sh_ev[];       // -DUB_EV=n: panel blocks of the first n block columns parked in LDS
    if (!dmreg::solve<NB, UB_EV>(T, rhs, store, lane, sh_ev + wave * (dmreg::ev_slot(NB, UB_EV, UB_EV + 1) * 256))) xo[0] = -1.0;
}

// relative error of v_rcp_f64 (how many Newton steps does the pivot reciprocal need?)
__global__ void k_rcp(const double* in, double* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_rcp(in[i]);
}

int main() {
    {
        const int n = 1 << 20;
        std::vector<double> h(n), r(n);
        srand(5);
        for (auto& v : h) v = ldexp(1.0 + rand() / (double)RAND_MAX + rand() / ((double)RAND_MAX * RAND_MAX), rand() % 40 - 20);
        double *di, *dout;
        CK(hipMalloc(&di, n * 8)); CK(hipMalloc(&dout, n * 8));
        CK(hipMemcpy(di, h.data(), n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_rcp, dim3(n / 256), dim3(256), 0, 0, di, dout, n);
        CK(hipMemcpy(r.data(), dout, n * 8, hipMemcpyDeviceToHost));
        long double worst = 0;
        for (int i = 0; i < n; ++i) { long double e = fabsl((long double)r[i] * (long double)h[i] - 1.0L); if (e > worst) worst = e; }
        printf("v_rcp_f64: max |x rcp(x) - 1| over 2^20 inputs = %.3Le = 2^%.1Lf\n", worst, log2l(worst));
    }
    constexpr int NB = 8, n = NB * 16, NBLK = NB * (NB + 1) / 2, NS = 64;
    std::vector<double> A((size_t)NS * n * n), img((size_t)NS * NBLK * 256), rhs((size_t)NS * n);
    srand(3);
    for (int s = 0; s < NS; ++s) {
        std::vector<double> G((size_t)n * 40);
        for (auto& v : G) v = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                double a = (i == j) ? 1.0 + 0.01 * i : 0.0;
                for (int q = 0; q < 40; ++q) a += G[i * 40 + q] * G[j * 40 + q];
                A[((size_t)s * n + i) * n + j] = A[((size_t)s * n + j) * n + i] = a;
            }
        for (int i = 0; i < n; ++i) rhs[(size_t)s * n + i] = rand() / (double)RAND_MAX;
        for (int I = 0; I < NB; ++I)
            for (int K = 0; K <= I; ++K)
                for (int kk = 0; kk < 16; ++kk)
                    for (int ii = 0; ii < 16; ++ii)      // T_IK[kk][ii] = A[16 I + ii][16 K + kk]
                        img[((size_t)s * NBLK + I * (I + 1) / 2 + K) * 256 + kk * 16 + ii] = A[((size_t)s * n + 16 * I + ii) * n + 16 * K + kk];
    }
    double *dimg, *drhs, *dx;
    const int nsys_full = 256 * 4 * 8;
    CK(hipMalloc(&dimg, img.size() * 8)); CK(hipMalloc(&drhs, rhs.size() * 8)); CK(hipMalloc(&dx, (size_t)nsys_full * n * 8));
    CK(hipMemcpy(dimg, img.data(), img.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(drhs, rhs.data(), rhs.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nsys : {1, 4, 256 * 4, nsys_full}) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            const size_t evb = (size_t)4 * dmreg::ev_slot(NB, UB_EV, UB_EV + 1) * 256 * sizeof(double);
            CK(hipFuncSetAttribute((const void*)k_solve<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)evb));
            hipLaunchKernelGGL(k_solve<NB>, dim3((nsys + 3) / 4), dim3(256), evb, 0, dimg, drhs, dx, nsys);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        long long t[8];
        CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_dmreg_t), sizeof(t)));
        long long tot = 0; for (int q = 0; q < 6; ++q) tot += t[q];
        printf("%5d systems: %8.1f us   wave 0 cycles: chains %lld  panels %lld  trailing %lld  forward %lld  backward %lld  (total %lld)\n",
               nsys, ms * 1e3, t[1], t[2], t[3], t[4], t[5], tot);
    }
    // residual of system 0
    std::vector<double> x(n);
    CK(hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost));
    double res = 0;
    for (int i = 0; i < n; ++i) { double a = -rhs[i]; for (int j = 0; j < n; ++j) a += A[(size_t)i * n + j] * x[j]; res = fmax(res, fabs(a)); }
    printf("max |A x - b| of system 0: %.3e\n", res);
    return 0;
}

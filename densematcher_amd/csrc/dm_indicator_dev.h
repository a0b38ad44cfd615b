// The mapped indicator M = (Phi2 C) Phi1^T diag(a1) of small maps (k1 <= 32) as ONE arithmetic shared by everybody who needs its
// entries (reference pyFM/spectral/convert.py:144): dm_mapped_indicator (the dense matrix, when a caller wants it) and the linear
// assignment's on-the-fly cost rows (dm_assign.hip: at 64 pairs the three dense matrices per pair are 6 GB that no cache holds, and a
// search step waits for a 16 KiB row from HBM; the factors of a pair are 0.5 MB).  Both evaluate
//     M_ij = ( sum_k E2[i][k] * P1[j][k] ) * a1[j]      k ascending, one fma per term from 0, then one multiplication
// on the padded float64 factors E2 = Phi2 C (fma chain over the rows of C, ascending) and P1 = Phi1[:, :k1] built by the kernels
// below -- so an assignment computed from the factors is the assignment of the dense matrix, bit for bit.
#pragma once
#include "dm_device.h"

template <int KP>
__device__ __forceinline__ double ind_value(const double* __restrict__ e, const double (&p)[KP], double a1) {
    double g = 0.0;
#pragma unroll
    for (int k = 0; k < KP; ++k) g = fma(e[k], p[k], g);
    return g * a1;
}

// E2p (B, N2, KP) = Phi2[:, :k2] C, columns >= k1 zero;  grid (ceil(N2 / 256), B)
template <typename TR>
__global__ __launch_bounds__(256) void ind_e2_kernel(const TR* __restrict__ Phi2, int ld2, int N2, int k1, int k2, const double* __restrict__ C,
                                                     int KP, double* __restrict__ E2p, int32_t* __restrict__ bad) {
    __shared__ double Cs[32 * 32];
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    for (int e = threadIdx.x; e < k2 * k1; e += 256) Cs[e] = C[(long long)b * k2 * k1 + e];
    __syncthreads();
    if (i >= N2) return;
    const TR* row = Phi2 + ((long long)b * N2 + i) * ld2;
    double* out = E2p + ((long long)b * N2 + i) * KP;
    bool nonfinite = false;
    for (int c = 0; c < KP; ++c) {
        double s = 0.0;
        if (c < k1)
            for (int a = 0; a < k2; ++a) s = fma((double)row[a], Cs[a * k1 + c], s);
        nonfinite = nonfinite || !isfinite(s);
        out[c] = s;
    }
    if (nonfinite && bad) atomicOr(bad + b, 2);
}
// P1p (B, N1, KP) = Phi1[:, :k1] zero padded, a1p (B, N1) = float64 masses
template <typename TR>
__global__ __launch_bounds__(256) void ind_p1_kernel(const TR* __restrict__ Phi1, int ld1, const TR* __restrict__ mass1, int N1, int k1, int KP,
                                                     double* __restrict__ P1p, double* __restrict__ a1p, int32_t* __restrict__ bad) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N1) return;
    const TR* row = Phi1 + ((long long)b * N1 + j) * ld1;
    double* out = P1p + ((long long)b * N1 + j) * KP;
    bool nonfinite = false;
    for (int c = 0; c < KP; ++c) { const double x = c < k1 ? (double)row[c] : 0.0; nonfinite = nonfinite || !isfinite(x); out[c] = x; }
    const double a = (double)mass1[(long long)b * N1 + j];
    nonfinite = nonfinite || !isfinite(a);
    a1p[(long long)b * N1 + j] = a;
    if (nonfinite && bad) atomicOr(bad + b, 2);
}
// the dense matrix from the factors: thread = one column j (its P1 row in registers), a workgroup sweeps 64 rows; grid (ceil(N1 / 256), ceil(N2 / 64), B)
template <int KP>
__global__ __launch_bounds__(256) void ind_dense_kernel(const double* __restrict__ E2p, const double* __restrict__ P1p, const double* __restrict__ a1p,
                                                        int N1, int N2, double* __restrict__ M) {
    const int b = blockIdx.z, j = blockIdx.x * 256 + threadIdx.x, i0 = blockIdx.y * 64;
    if (j >= N1) return;
    double p[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) p[k] = P1p[((long long)b * N1 + j) * KP + k];
    const double a1 = a1p[(long long)b * N1 + j];
    const int i1 = min(N2, i0 + 64);
    for (int i = i0; i < i1; ++i)
        M[((long long)b * N2 + i) * N1 + j] = ind_value<KP>(E2p + ((long long)b * N2 + i) * KP, p, a1);
}

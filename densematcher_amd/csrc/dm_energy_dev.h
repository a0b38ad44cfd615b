// Device code shared by dm_energy.hip (dm_fmap_energy_grad) and dm_fitfuse.hip (the one-launch-per-evaluation fit of small maps):
// the workgroup sum and the quadratic terms of the functional-map energy (base_functions.py:31-121).
#pragma once
#include "dm_device.h"

struct OutNT {
    double* p; long long stride_b; int ld;
    __device__ __forceinline__ void store(int b, int i, int j, double v) const { p[b * stride_b + (long long)i * ld + j] = v; }
};

// deterministic workgroup sum (256 threads): returns the total in every thread
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---- quadratic terms -------------------------------------------------------------------------------------------------
// grad = w_d (C P - Q) + w_l C * ev;  e_quad[b] = 1/2 w_d (sum C (CP - 2Q) + |B|^2) + 1/2 w_l sum C^2 ev      (one workgroup per pair)
struct quad_args {
    const double* C; const double* CP; const double* PQ; const float* Bm; const double* lam1; const double* lam2; int D; double w_d, w_l;
};
// (element e of the pair is written by thread e mod 256: a caller that goes on with the same mapping needs no barrier)
__device__ __forceinline__ double quad_pair(const quad_args& qa, int b, int t, int k1, int k2, double* __restrict__ grad, double* sh) {
    const double* C = qa.C; const double* CP = qa.CP; const double* PQ = qa.PQ; const float* Bm = qa.Bm;
    const double* lam1 = qa.lam1; const double* lam2 = qa.lam2;
    const int D = qa.D;
    const double w_d = qa.w_d, w_l = qa.w_l;
    double mx = 0.0;
    for (int j = t; j < k1; j += 256) mx = fmax(mx, lam1[(long long)b * k1 + j]);
    for (int i = t; i < k2; i += 256) mx = fmax(mx, lam2[(long long)b * k2 + i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if ((t & 63) == 0) sh[t >> 6] = mx;
    __syncthreads();
    const double scale = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
    const double* Cb = C + (long long)b * k2 * k1;
    const double* CPb = CP ? CP + (long long)b * k2 * k1 : nullptr;
    const double* Q = PQ + ((long long)b * (k1 + k2) + k1) * k1;
    double acc = 0.0;
    for (int e = t; e < k2 * k1; e += 256) {
        const int i = e / k1, j = e - i * k1;
        double cp;
        if (CP) cp = CPb[e];
        else {                                            // small maps: (C P)_ij by this thread (P = the first k1 rows of PQ), a launch less
            const double* Pm = PQ + (long long)b * (k1 + k2) * k1;
            cp = 0.0;
            for (int k = 0; k < k1; ++k) cp = fma(Cb[(long long)i * k1 + k], Pm[(long long)k * k1 + j], cp);
        }
        const double c = Cb[e], q = Q[e];
        const double dl = lam1[(long long)b * k1 + j] / scale - lam2[(long long)b * k2 + i] / scale;   // functional.py:404-405
        const double ev = dl * dl;
        grad[(long long)b * k2 * k1 + e] = w_d * (cp - q) + w_l * c * ev;
        acc += 0.5 * w_d * c * (cp - 2.0 * q) + 0.5 * w_l * c * c * ev;
    }
    double bn = 0.0;
    for (int e = t; e < k2 * D; e += 256) { const double x = (double)Bm[(long long)b * k2 * D + e]; bn += x * x; }
    return block_sum_256(acc + 0.5 * w_d * bn, sh);       // (its first barrier: every thread has read the maxima)
}

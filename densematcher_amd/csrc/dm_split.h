// Device helpers shared by the fp16-split nearest-neighbour passes (dm_simnn.hip, dm_knnsplit.hip, dm_zoomfuse.hip).
#pragma once
#include "dm_device.h"
#include "dm_internal.h"

#define DM_NEG_INF_F32 (-__builtin_huge_valf())

// running (best, index, second best) of a row, merged with another partial; equal scores: the lower index wins
__device__ __forceinline__ void top2_merge(float& b, int& i, float& s, float ob, int oi, float os) {
    if (ob > b || (ob == b && oi < i)) { s = fmaxf(b, os); b = ob; i = oi; }
    else { s = fmaxf(s, ob); }
}

// power of two s with (max of the `count` partial maxima) * s in [1, 2)  (1 when the operand is all zero)
__device__ __forceinline__ double ks_scale(const double* __restrict__ amax, int count) {
    double m = 0.0;
    for (int q = 0; q < count; ++q) m = fmax(m, amax[q]);
    int ex = 0;
    if (!(m > 0.0) || !(m < DM_INF_F64)) return 1.0;
    (void)frexp(m, &ex);                                          // m = f 2^ex, f in [0.5, 1)
    return ldexp(1.0, 1 - ex);
}

__device__ __forceinline__ void split2(double v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (double)hi);
}

// The same split through hardware conversions only (f64 -> f32 -> f16; a direct f64 -> f16 conversion is a software
// routine of ~30 instructions): hi is the fp16 nearest to fl32(v) -- within half an fp16 ulp (1 + 2^-13) of v --, the
// remainder v - hi is exact in float64, and lo rounds it to fp16 via fp32 (relative error 2^-11 (1 + 2^-13)):
// |v - hi - lo| <= 2^-22 (1 + 2^-12) |v|, inside the 25 % slack of the bound the callers budget for 2^-22.
// (the f64 -> f32 step is an inline asm: written as casts, the compiler folds the two truncations back into the direct one)
__device__ __forceinline__ float cvt_f32_f64_hw(double v) { float f; asm("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(v)); return f; }
__device__ __forceinline__ void split2_hw(double v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)cvt_f32_f64_hw(v);
    lo = (_Float16)cvt_f32_f64_hw(v - (double)(float)hi);
}

// one entry of a basis row scaled by the power of two sx and split for the tile kernels (fs_build_rows_kernel and the embedding that
// writes the same rows on the fly: the same bits).  fp32 basis: x sx, |.| < 2, is exact in fp32 and so is x sx - h: the pieces equal
// those of the float64 split; fp64 basis: the split runs in float64 (split2_hw)
template <typename TR>
__device__ __forceinline__ void fs_split_entry(TR xin, double sx, _Float16& h, _Float16& l) {
    if constexpr (sizeof(TR) == 4) {
        const float x = (float)((double)xin * sx);
        h = (_Float16)x; l = (_Float16)(x - (float)h);
    } else {
        split2_hw((double)xin * sx, h, l);
    }
}


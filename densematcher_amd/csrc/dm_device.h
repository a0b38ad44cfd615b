// Device-side helpers shared by the kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define DM_INF_F64 __builtin_huge_val()
#define DM_IDX_NONE 0x7fffffff

// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) * B(4x16), one f64 of A and of B per
// lane.  Lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15].
// Result register r of lane l is D[row = (l >> 4) + 4 r][col = l & 15]
// (the f64 form does NOT use the f32 C/D map; cdna_hip_programming.md section 3).
__device__ __forceinline__ f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// XCD-aware bijective remap of a 1-D grid: hardware block b runs on XCD b % 8;
// give every XCD one contiguous range of logical tile ids so that the tiles of
// one mesh pair (which share operand panels) meet in the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// (value, index) candidates; "better" is strictly larger value, or equal value
// and lower index (NumPy argmax / argmin first-index rule).
__device__ __forceinline__ void argmax_merge(double& v, int& j, double ov, int oj) {
    if (ov > v || (ov == v && oj < j)) { v = ov; j = oj; }
}
__device__ __forceinline__ void argmin_merge(double& v, int& j, double ov, int oj) {
    if (ov < v || (ov == v && oj < j)) { v = ov; j = oj; }
}

// DPP exchanges inside a 16-lane row: 0xB1 = quad_perm [1,0,3,2], 0x4E = quad_perm [2,3,0,1],
// 0x141 = row_half_mirror, 0x140 = row_mirror.  Applied in this order with max/min they leave the
// reduction over the 16 lanes in every lane (VALU speed, no LDS round trip).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int x) {
    return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false);
}

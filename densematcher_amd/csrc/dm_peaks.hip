// dm_measure_peak: on-box peak probes for bench.py's roofline block ("peak denominators measured on the box
// ... recorded next to the public-spec numbers").  No reference counterpart.
//
//   DM_PEAK_MFMA_F16_ZERO    v_mfma_f32_32x32x16_f16, all-zero operands: the instruction-issue ceiling (the part holds its
//                            2.4 GHz: ~2.48 PFLOP/s, what MI355X_MICROARCH.md quotes as 99.8 % of 2.5 PF)
//   DM_PEAK_MFMA_F16_RANDOM  the same loop on N(0,1) operands: the power-limited ceiling of REAL data -- the switching
//                            activity of the multipliers pulls the shader clock to ~1.7 GHz (~1.72 PFLOP/s; tools/
//                            ubench_mfma_clock.hip prints the clock next to the rate).  This is what a feature-similarity
//                            kernel can at most sustain.
//   DM_PEAK_MFMA_F64         v_mfma_f64_16x16x4_f64
//   DM_PEAK_HBM_COPY         float4 stream copy of 1 GiB of the context workspace, read + written bytes per second
#include <math.h>

#include "dm_device.h"
#include "dm_internal.h"

template <int CH>
__global__ __launch_bounds__(256) void peak_f16_kernel(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
    f32x16 a[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[c][r] = 0.f;
    const f16x8 x = in[threadIdx.x], y = in[256 + threadIdx.x];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) a[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += a[c][c & 15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// (compile-time trip count and four named accumulators: with a run-time loop over an accumulator ARRAY the compiler shuttles
//  the f64 accumulators between VGPRs and AGPRs every iteration and the probe reads 46 instead of 73 TFLOP/s)
template <int ITER>
__global__ __launch_bounds__(256) void peak_f64_kernel(const double* __restrict__ in, double* __restrict__ out) {
    f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = in[threadIdx.x], y = in[256 + threadIdx.x];
    for (int i = 0; i < ITER; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

__global__ __launch_bounds__(256) void peak_copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

extern "C" int dm_measure_peak(dm_ctx* ctx, int which, double* value) {
    if (!ctx || !value) return DM_EINVAL;
    DM_REQUIRE(ctx, which >= DM_PEAK_MFMA_F16_ZERO && which <= DM_PEAK_HBM_COPY, "unknown probe");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const int ncu = ctx->n_cu > 0 ? ctx->n_cu : 256;
    const size_t copy_bytes = (size_t)1 << 30;
    int rc = dm_ws_reserve(ctx, which == DM_PEAK_HBM_COPY ? 2 * copy_bytes + 4096 : ((size_t)64 << 20));
    if (rc) return rc;
    hipEvent_t e0, e1;
    DM_CHECK_HIP(ctx, hipEventCreate(&e0));
    DM_CHECK_HIP(ctx, hipEventCreate(&e1));
    float ms = 0.f;
    double flops_or_bytes = 0.0;
    if (which == DM_PEAK_HBM_COPY) {
        char* a = (char*)dm_ws_take(ctx, copy_bytes);
        char* b = (char*)dm_ws_take(ctx, copy_bytes);
        if (!a || !b) return dm_fail(ctx, DM_ENOMEM, "measure_peak: workspace not reserved");
        DM_CHECK_HIP(ctx, hipMemsetAsync(a, 1, copy_bytes, ctx->stream));
        for (int rep = 0; rep < 3; ++rep) {
            DM_CHECK_HIP(ctx, hipEventRecord(e0, ctx->stream));
            hipLaunchKernelGGL(peak_copy_kernel, dim3(ncu * 16), dim3(256), 0, ctx->stream, (const float4*)a, (float4*)b, copy_bytes / 16);
            DM_CHECK_HIP(ctx, hipEventRecord(e1, ctx->stream));
            DM_CHECK_HIP(ctx, hipEventSynchronize(e1));
            DM_CHECK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
        }
        flops_or_bytes = 2.0 * (double)copy_bytes;
    } else {
        char* buf = (char*)dm_ws_take(ctx, (size_t)32 << 20);
        if (!buf) return dm_fail(ctx, DM_ENOMEM, "measure_peak: workspace not reserved");
        // operands: 512 x 8 halves (or 512 doubles), a fixed pseudo-random N(0,1) sequence (sum of 12 uniforms) or zeros
        unsigned long long st = 0x9E3779B97F4A7C15ull;
        auto gauss = [&]() {
            double s = 0.0;
            for (int q = 0; q < 12; ++q) { st = st * 6364136223846793005ull + 1442695040888963407ull; s += (double)(st >> 11) / 9007199254740992.0; }
            return s - 6.0;
        };
        if (which == DM_PEAK_MFMA_F64) {
            double h[512];
            for (double& v : h) v = gauss();
            DM_CHECK_HIP(ctx, hipMemcpyAsync(buf, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
            DM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            constexpr int ITER = 8192;
            const int grid = ncu * 8;
            for (int rep = 0; rep < 2; ++rep) {
                DM_CHECK_HIP(ctx, hipEventRecord(e0, ctx->stream));
                hipLaunchKernelGGL(peak_f64_kernel<ITER>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)buf, (double*)(buf + 8192));
                DM_CHECK_HIP(ctx, hipEventRecord(e1, ctx->stream));
                DM_CHECK_HIP(ctx, hipEventSynchronize(e1));
                DM_CHECK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
            }
            flops_or_bytes = (double)grid * 4 * ITER * 4 * 2048.0;
        } else {
            _Float16 h[512 * 8];
            for (_Float16& v : h) v = (_Float16)(which == DM_PEAK_MFMA_F16_RANDOM ? gauss() : 0.0);
            DM_CHECK_HIP(ctx, hipMemcpyAsync(buf, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
            DM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const int iters = 32768, grid = ncu * 2;          // two waves per SIMD, four independent accumulator chains each
            for (int rep = 0; rep < 2; ++rep) {
                DM_CHECK_HIP(ctx, hipEventRecord(e0, ctx->stream));
                hipLaunchKernelGGL(peak_f16_kernel<4>, dim3(grid), dim3(256), 0, ctx->stream, (const f16x8*)buf, (float*)(buf + 16384), iters);
                DM_CHECK_HIP(ctx, hipEventRecord(e1, ctx->stream));
                DM_CHECK_HIP(ctx, hipEventSynchronize(e1));
                DM_CHECK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
            }
            flops_or_bytes = (double)grid * 4 * iters * 4 * 32768.0;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *value = flops_or_bytes / ((double)ms * 1e-3);
    return DM_OK;
}

// Vertex map -> functional map (dm_p2p_to_fm) and ZoomOut refinement (dm_zoomout).
//
// Reference arithmetic reproduced (oracle/dm_oracle.py: p2p_to_fm, zoomout_refine):
//   C' = Phi2[:, :k2']^T (a2 * Phi1[p21, :k1'])                      pyFM/spectral/convert.py:39-48
//   repeat nit: p21 = NN(tree = Phi1[:, :k] C^T, query = Phi2[:, :k]) pyFM/refine/zoomout.py:40
//               C <- p2p_to_FM(p21) with k + step columns             pyFM/refine/zoomout.py:42
// The NN step is the fused G-tile kernel of dm_p2p.hip restricted to the row argmin; Phi2^T is
// staged K-major in float64 once for all iterations, the whole loop runs on the stream with no
// host synchronisation.
#include "dm_gemm_f64.h"
#include "dm_internal.h"

struct OutFM {
    double* direct; int ldc; long long strideC;     // nsplit == 1
    double* partial; int B, k2, k1;                 // (nsplit, B, k2, k1)
    __device__ __forceinline__ void store(int b, int split, int m, int c, double v) const {
        if (direct) direct[b * strideC + (long long)m * ldc + c] = v;
        else partial[(((long long)split * B + b) * k2 + m) * k1 + c] = v;
    }
};

__global__ __launch_bounds__(256) void splitk_reduce_fm_kernel(const double* __restrict__ partial, int nsplit, int B,
                                                               int k2, int k1, double* __restrict__ C, int ldc,
                                                               long long strideC) {
    const long long n = (long long)B * k2 * k1;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int q = 0; q < nsplit; ++q) s += partial[(long long)q * n + i];
    const int c = (int)(i % k1);
    const long long bm = i / k1;
    const int m = (int)(bm % k2);
    const int b = (int)(bm / k2);
    C[b * strideC + (long long)m * ldc + c] = s;
}

// split-K by a fixed chunk of vertices: the summation order of a pair must not depend on the batch it is in
constexpr int FM_KCHUNK = 256;
static int fm_split(int B, int N2, int k1, int k2) {
    (void)B; (void)k1; (void)k2;
    return dm_cdiv(N2, FM_KCHUNK);
}

size_t dm_p2pfm_ws_bytes(int B, int N2, int k1, int k2) {
    const int nsplit = fm_split(B, N2, k1, k2);
    return nsplit > 1 ? dm_align_up((size_t)nsplit * B * k2 * k1 * 8) + 4096 : 4096;
}

template <typename TR>
int dm_launch_p2p_to_fm(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21, const TR* Phi1,
                        int ld1, const TR* Phi2, int ld2, const double* mass2, double* C, int ldc,
                        long long strideC) {
    const int nsplit = fm_split(B, N2, k1, k2);
    const int kchunk = FM_KCHUNK;
    double* partial = nullptr;
    if (nsplit > 1) {
        partial = (double*)dm_ws_take(ctx, (size_t)nsplit * B * k2 * k1 * 8);
        if (!partial) return dm_fail(ctx, DM_ENOMEM, "p2p_to_fm: workspace not reserved");
    }
    RowsScaled<TR, double> opx{Phi2, (long long)N2 * ld2, ld2, k2, nullptr, 0};
    RowsGatherScaled<TR, double> opy{Phi1, (long long)N1 * ld1, ld1, k1, p21, (long long)N2, N1, mass2, (long long)N2};
    OutFM out{nsplit > 1 ? nullptr : C, ldc, strideC, partial, B, k2, k1};
    dim3 grid(dm_cdiv(k2, TN_T) * dm_cdiv(k1, TN_T), nsplit, B);
    DM_LAUNCH(ctx, "p2pfm_tn_f64", (gemm_tn_f64<RowsScaled<TR, double>, RowsGatherScaled<TR, double>, OutFM>), grid, dim3(256), 0, opx,
              opy, out, k2, k1, N2, kchunk);
    if (nsplit > 1) {
        const long long n = (long long)B * k2 * k1;
        DM_LAUNCH(ctx, "splitk_reduce", splitk_reduce_fm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, partial,
                  nsplit, B, k2, k1, C, ldc, strideC);
    }
    return DM_OK;
}
template int dm_launch_p2p_to_fm<float>(dm_ctx*, int, int, int, int, int, const int32_t*, const float*, int, const float*, int,
                                        const double*, double*, int, long long);
template int dm_launch_p2p_to_fm<double>(dm_ctx*, int, int, int, int, int, const int32_t*, const double*, int, const double*, int,
                                         const double*, double*, int, long long);

template <typename TR>
static int p2p_to_fm_impl(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21, const TR* Phi1,
                          int ld1, const TR* Phi2, int ld2, const TR* mass2_in, double* C) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N1 > 0 && N2 > 0 && k1 > 0 && k2 > 0, "sizes must be positive");
    DM_REQUIRE(ctx, p21 && Phi1 && Phi2 && mass2_in && C, "null pointer");
    DM_REQUIRE(ctx, ld1 >= k1 && ld2 >= k2, "eigenvector row stride smaller than the map size");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = dm_ws_reserve(ctx, dm_p2pfm_ws_bytes(B, N2, k1, k2) + dm_align_up((size_t)B * N2 * 8));
    if (rc) return rc;
    const double* mass2 = nullptr;
    rc = dm_widen_mass(ctx, (long long)B * N2, mass2_in, (double*)dm_ws_take(ctx, (size_t)B * N2 * 8), &mass2);
    if (rc) return rc;
    return dm_launch_p2p_to_fm<TR>(ctx, B, N1, N2, k1, k2, p21, Phi1, ld1, Phi2, ld2, mass2, C, k1, (long long)k2 * k1);
}
extern "C" int dm_p2p_to_fm(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21, const float* Phi1,
                            int ld1, const float* Phi2, int ld2, const float* mass2, double* C) {
    return p2p_to_fm_impl<float>(ctx, B, N1, N2, k1, k2, p21, Phi1, ld1, Phi2, ld2, mass2, C);
}
extern "C" int dm_p2p_to_fm_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21, const double* Phi1,
                                int ld1, const double* Phi2, int ld2, const double* mass2, double* C) {
    return p2p_to_fm_impl<double>(ctx, B, N1, N2, k1, k2, p21, Phi1, ld1, Phi2, ld2, mass2, C);
}

static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

template <typename TR>
static int zoomout_impl(dm_ctx* ctx, int B, int N1, int N2, int k0, int nit, int step, const TR* Phi1, int ld1,
                        const TR* Phi2, int ld2, const TR* mass2_in, const double* C0, double* Cout,
                        int32_t* p21_out) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N1 > 0 && N2 > 0 && k0 > 0 && nit >= 0 && step > 0, "sizes must be positive");
    DM_REQUIRE(ctx, Phi1 && Phi2 && mass2_in && C0 && Cout, "null pointer");
    const int kf = k0 + nit * step;
    DM_REQUIRE(ctx, ld1 >= kf && ld2 >= kf, "not enough eigenvectors for k0 + nit*step (zoomout.py:87-92)");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));

    const int N1pad = pad_to(N1, 128), N2pad = pad_to(N2, 128), Kpad = pad_to(kf, 16);
    const size_t bytes_AT = (size_t)B * Kpad * N2pad * 8, bytes_BT = (size_t)B * Kpad * N1pad * 8;
    const size_t bytes_C = (size_t)B * kf * kf * 8;
    const size_t need = dm_align_up(bytes_AT) + dm_align_up(bytes_BT) + 2 * dm_align_up(bytes_C) +
                        dm_align_up((size_t)B * N1pad * 8) + dm_align_up((size_t)B * N2 * 4) +
                        dm_gred_ws_bytes(B, N2, N1) + dm_knn_split_prep_bytes(B, N2, kf) + dm_knn_split_ws_bytes(B, N2, N1, kf) +
                        dm_align_up((size_t)B * (N1pad / DM_EMB_COLS + 1) * 8) + dm_p2pfm_ws_bytes(B, N2, kf, kf) +
                        dm_align_up((size_t)B * N2 * 8);
    int rc = dm_ws_reserve(ctx, need);
    if (rc) return rc;
    const double* mass2 = nullptr;
    rc = dm_widen_mass(ctx, (long long)B * N2, mass2_in, (double*)dm_ws_take(ctx, (size_t)B * N2 * 8), &mass2);
    if (rc) return rc;
    double* AT = (double*)dm_ws_take(ctx, bytes_AT);
    double* BT = (double*)dm_ws_take(ctx, bytes_BT);
    double* Ca = (double*)dm_ws_take(ctx, bytes_C);
    double* Cb = (double*)dm_ws_take(ctx, bytes_C);
    double* n1 = (double*)dm_ws_take(ctx, (size_t)B * N1pad * 8);
    int32_t* p21 = (int32_t*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    double* amaxS = (double*)dm_ws_take(ctx, (size_t)B * (N1pad / DM_EMB_COLS + 1) * 8);   // max |emb1| per block of columns (dm_launch_embed)
    if (!AT || !BT || !Ca || !Cb || !n1 || !p21 || !amaxS) return dm_fail(ctx, DM_ENOMEM, "zoomout: workspace not reserved");

    // Phi2^T for all kf columns, once.  Row c of AT only enters G when c < current k because the
    // matching row of BT (emb1^T) is zero beyond the current map size.
    rc = dm_launch_phiT<TR>(ctx, B, N2, kf, Phi2, ld2, AT, Kpad, N2pad);
    if (rc) return rc;
    // target side of the nearest-neighbour search (fp16 split of Phi2, all kf columns), once for the whole call
    dm_knn_split_state knn;
    rc = dm_knn_split_prepare(ctx, B, N2, N2pad, Kpad, kf, AT, &knn);
    if (rc) return rc;
    const size_t ws_mark = ctx->ws_off;
    DM_CHECK_HIP(ctx, hipMemcpyAsync(Ca, C0, (size_t)B * k0 * k0 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    // emb1^T buffer: rows >= current k and columns >= N1 must read as zero; rows only ever grow, so one clear suffices
    DM_CHECK_HIP(ctx, hipMemsetAsync(BT, 0, bytes_BT, ctx->stream));

    double* cur = Ca;
    double* nxt = Cb;
    int k = k0;
    for (int it = 0; it <= nit; ++it) {
        const bool last = (it == nit);
        if (last && !p21_out) break;
        ctx->ws_off = ws_mark;
        // emb1 = Phi1[:, :k] C^T  ->  BT (rows >= k zero), n1
        rc = dm_launch_embed<TR>(ctx, B, N1, k, k, Phi1, ld1, cur, k, (long long)k * k, 0, BT, Kpad, N1pad, n1, 0, amaxS);
        if (rc) return rc;
        dm_gred_args a;
        a.B = B; a.N2 = N2; a.N1 = N1; a.Kloop = pad_to(k, 16); a.Ktrue = k;
        a.AT = AT; a.N2pad = N2pad; a.BT = BT; a.N1pad = N1pad; a.Kpad = Kpad;
        a.n1 = n1; a.n2 = nullptr; a.mass1 = nullptr;
        a.knn21 = last ? p21_out : p21; a.knn12 = nullptr; a.ind21 = nullptr; a.ind12 = nullptr;
        rc = dm_launch_knn21(ctx, a, knn, amaxS, dm_cdiv(N1pad, DM_EMB_COLS));
        if (rc) return rc;
        if (last) break;
        const int kn = k + step;
        rc = dm_launch_p2p_to_fm<TR>(ctx, B, N1, N2, kn, kn, p21, Phi1, ld1, Phi2, ld2, mass2, nxt, kn, (long long)kn * kn);
        if (rc) return rc;
        double* tmp = cur; cur = nxt; nxt = tmp;
        k = kn;
    }
    DM_CHECK_HIP(ctx, hipMemcpyAsync(Cout, cur, (size_t)B * kf * kf * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return DM_OK;
}
extern "C" int dm_zoomout(dm_ctx* ctx, int B, int N1, int N2, int k0, int nit, int step, const float* Phi1, int ld1,
                          const float* Phi2, int ld2, const float* mass2, const double* C0, double* Cout,
                          int32_t* p21_out) {
    return zoomout_impl<float>(ctx, B, N1, N2, k0, nit, step, Phi1, ld1, Phi2, ld2, mass2, C0, Cout, p21_out);
}
extern "C" int dm_zoomout_f64(dm_ctx* ctx, int B, int N1, int N2, int k0, int nit, int step, const double* Phi1, int ld1,
                              const double* Phi2, int ld2, const double* mass2, const double* C0, double* Cout,
                              int32_t* p21_out) {
    return zoomout_impl<double>(ctx, B, N1, N2, k0, nit, step, Phi1, ld1, Phi2, ld2, mass2, C0, Cout, p21_out);
}

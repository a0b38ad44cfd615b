// Spectral ICP refinement (dm_icp) -- SURVEY.md "next #1".
//
// Reference arithmetic reproduced (oracle/dm_oracle.py: icp_refine; pyFM/refine/icp.py:10-40,43-107):
//   repeat nit times:  p21 = NN(tree = Phi1 C^T, query = Phi2)                       icp.py:36 -> convert.py:138-140
//                      Chat = lstsq(Phi2[:, :k2], Phi1[p21, :k1])   (no mass)         icp.py:37 -> convert.py:51
//                      U, _, Vt = svd(Chat);  C = U eye(k2,k1) Vt                     icp.py:38-40
// GPU formulation: the least-squares step is the normal equations (Phi2^T Phi2) Chat = Phi2^T Phi1[p21] solved with
// the blocked LDS Cholesky (the Gram matrix is factored per right-hand-side column; it is iteration independent);
// U eye Vt is the orthogonal polar factor of Chat, obtained without an SVD on the float64 matrix cores by odd
// matrix polynomials (they act on the singular values only, U and V are untouched):
//   lift:   NS_LIFT steps of  X <- a X + X (b T + c T^2),  T = X^T X,  (a, b, c) = (3.4445, -4.7750, 2.0315):
//           multiplies a small singular value by 3.44 per step and keeps every one inside about [0.68, 1.13];
//   polish: NS_POLISH Newton-Schulz steps  X <- 1.5 X - 0.5 X T  (quadratic convergence from that interval).
//           From 0.68: 0.863, 0.973, 0.99891, 1 - 1.8e-6, 1 - 4.9e-12, 1 - 4e-23; from 1.13 faster: six steps (r04: eight).
// 12 + 6 steps reach |X^T X - I| ~ 1e-16 for sigma_min / sigma_max down to ~4e-7 (plain Newton-Schulz gains only a
// factor 1.5 per step: 22 steps stalled at 8e-4 on a Chat with sigma_min / sigma_max = 6e-4,
// tests/test_gpu_parity.py::test_refine_ragged).  The final |X^T X - I| is reported.
#include "dm_chol.h"
#include "dm_gemm_f64.h"
#include "dm_internal.h"

static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

constexpr int NS_LIFT = 12, NS_POLISH = 6;
// (the k x k products of the polar iteration with all four stages of operand loads in flight from the start, dm_gemm_f64.h NPRE = 4:
//  16.2 -> 19.1 us per launch; they stay at one stage ahead)
constexpr double NS_A = 3.4445, NS_B = -4.7750, NS_C = 2.0315;

// ---- helpers ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iota_ones_kernel(int32_t* __restrict__ idx, double* __restrict__ ones, int N, int B) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * B) return;
    idx[i] = (int32_t)(i % N);
    ones[i] = 1.0;
}

// row-major symmetric matrix G (B, n, n) -> blocked transposed LDS image (see dm_chol.h), padding = identity
__global__ __launch_bounds__(256) void blockify_kernel(const double* __restrict__ G, int n, int NB, double* __restrict__ img) {
    const int b = blockIdx.y, q = blockIdx.x, t = threadIdx.x;
    int I = 0;
    while ((I + 1) * (I + 2) / 2 <= q) ++I;
    const int K = q - I * (I + 1) / 2;
    const int kk = t >> 4, ii = t & 15;
    const int r = I * 16 + ii, c = K * 16 + kk;
    const double v = (r < n && c < n) ? G[((long long)b * n + c) * n + r] : ((r == c) ? 1.0 : 0.0);
    img[((long long)b * (NB * (NB + 1) / 2) + q) * 256 + kk * 16 + ii] = v;
}

// X[b][:, c] = G[b]^-1 R[b][:, c] (R = identity when null): one workgroup per (column c, pair b)
__global__ __launch_bounds__(256) void spd_multi_rhs_kernel(const double* __restrict__ img, const double* __restrict__ R, int n,
                                                            int nrhs, int NB, double* __restrict__ X, int32_t* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int nblk = NB * (NB + 1) / 2;
    double* T = sm;
    double* LT = T + nblk * 256;
    double* Ws = LT + 256;
    double* rhs = Ws + 256;
    double* xv = rhs + NB * 16;
    double* red = xv + NB * 16 + 16;
    const int b = blockIdx.y, c = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    {
        const f64x2* src = reinterpret_cast<const f64x2*>(img + (long long)b * nblk * 256);
        f64x2* dst = reinterpret_cast<f64x2*>(T);
        const int nvec = nblk * 128;
        for (int q0 = 0; q0 < nvec; q0 += 256 * 8) {
            f64x2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u * 256 + t;
                v[u] = (q < nvec) ? src[q] : f64x2{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u * 256 + t;
                if (q < nvec) dst[q] = v[u];
            }
        }
        // R == nullptr: right-hand side = unit vector c, i.e. column c of the inverse
        for (int r = t; r < NB * 16; r += 256) rhs[r] = (r < n) ? (R ? R[((long long)b * n + r) * nrhs + c] : (r == c ? 1.0 : 0.0)) : 0.0;
        int* tri_rc = reinterpret_cast<int*>(LT);
        for (int u = t; u < 128; u += 256) {
            int a_ = 0;
            while ((a_ + 1) * (a_ + 2) / 2 <= u) ++a_;
            tri_rc[u] = (a_ << 8) | (u - a_ * (a_ + 1) / 2);
        }
        if (t == 0) red[5] = 0.0;
    }
    __syncthreads();
    const bool ok = blocked_chol_solve(T, Ws, rhs, xv, red, reinterpret_cast<const int*>(LT), NB, t, lane, wave);
    if (!ok) {
        if (t == 0) atomicMax(&info[b], c + 1);
        for (int r = t; r < n; r += 256) X[((long long)b * n + r) * nrhs + c] = 0.0;
        return;
    }
    for (int r = t; r < n; r += 256) X[((long long)b * n + r) * nrhs + c] = xv[r];
}

// alpha_b = sqrt(|X|_1 |X|_inf) >= sigma_max;  X <- X / alpha      (one workgroup per pair)
__global__ __launch_bounds__(256) void polar_scale_kernel(double* __restrict__ X, int k2, int k1) {
    __shared__ double rs[256], cs[256];
    const int b = blockIdx.x, t = threadIdx.x;
    double* M = X + (long long)b * k2 * k1;
    double r = 0.0, c = 0.0;
    if (t < k2) for (int j = 0; j < k1; ++j) r += fabs(M[(long long)t * k1 + j]);
    if (t < k1) for (int i = 0; i < k2; ++i) c += fabs(M[(long long)i * k1 + t]);
    rs[t] = r; cs[t] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) { rs[t] = fmax(rs[t], rs[t + off]); cs[t] = fmax(cs[t], cs[t + off]); }
        __syncthreads();
    }
    const double alpha = sqrt(rs[0] * cs[0]);
    const double inv = alpha > 0.0 ? 1.0 / alpha : 0.0;
    for (int e = t; e < k2 * k1; e += 256) M[e] *= inv;
}

struct RowsF64 {                       // K-major f64 operand for gemm_tn_f64
    const double* p; long long stride_b; int ld; int ncols;
    __device__ __forceinline__ void load4(int b, int n, int col0, double (&v)[4]) const {
        const double* row = p + b * stride_b + (long long)n * ld;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (col0 + e < ncols) ? row[col0 + e] : 0.0;
    }
};
struct OutPlainTN {
    double* p; long long stride_b; int ld;
    __device__ __forceinline__ void store(int b, int, int m, int c, double v) const { p[b * stride_b + (long long)m * ld + c] = v; }
};
struct OutPlainNT {
    double* p; long long stride_b; int ld;
    __device__ __forceinline__ void store(int b, int i, int j, double v) const { p[b * stride_b + (long long)i * ld + j] = v; }
};
struct OutAxpby {                      // Xnew = alpha Xold + beta (product)
    const double* xo; double* xn; long long stride_b; int ld; double alpha, beta;
    __device__ __forceinline__ void store(int b, int i, int j, double v) const {
        const long long o = b * stride_b + (long long)i * ld + j;
        xn[o] = alpha * xo[o] + beta * v;
    }
};
// Newton-Schulz start for the inverse of a symmetric positive definite G (k2 > 176: the Gram matrix does not fit the
// in-LDS Cholesky): X0 = G / |G|_1^2, every eigenvalue of X0 G in (0, 1]   (one workgroup per pair)
__global__ __launch_bounds__(256) void ns_inverse_init_kernel(const double* __restrict__ G, int n, double* __restrict__ X) {
    __shared__ double cs[256];
    const int b = blockIdx.x, t = threadIdx.x;
    const double* M = G + (long long)b * n * n;
    double c = 0.0;
    if (t < n) for (int i = 0; i < n; ++i) c += fabs(M[(long long)i * n + t]);
    cs[t] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) cs[t] = fmax(cs[t], cs[t + off]);
        __syncthreads();
    }
    const double inv = cs[0] > 0.0 ? 1.0 / (cs[0] * cs[0]) : 0.0;
    for (int e = t; e < n * n; e += 256) X[(long long)b * n * n + e] = M[e] * inv;
}
// info[b] = n + 1 when the Newton-Schulz inverse did not converge (max |G X - I| > 1e-9)
__global__ __launch_bounds__(256) void ns_inverse_check_kernel(const double* __restrict__ Tm, int n, int32_t* __restrict__ info) {
    __shared__ double sh[256];
    const int b = blockIdx.x, t = threadIdx.x;
    double m = 0.0;
    for (int e = t; e < n * n; e += 256) m = fmax(m, fabs(Tm[(long long)b * n * n + e] - ((e / n == e % n) ? 1.0 : 0.0)));
    sh[t] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) sh[t] = fmax(sh[t], sh[t + off]);
        __syncthreads();
    }
    if (t == 0 && !(sh[0] <= 1e-9)) info[b] = n + 1;
}

// resid[b] = max |T - I| (one workgroup per pair)
__global__ __launch_bounds__(256) void ortho_resid_kernel(const double* __restrict__ Tm, int k, double* __restrict__ resid) {
    __shared__ double sh[256];
    const int b = blockIdx.x, t = threadIdx.x;
    double m = 0.0;
    for (int e = t; e < k * k; e += 256) m = fmax(m, fabs(Tm[(long long)b * k * k + e] - ((e / k == e % k) ? 1.0 : 0.0)));
    sh[t] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) sh[t] = fmax(sh[t], sh[t + off]);
        __syncthreads();
    }
    if (t == 0) resid[b] = sh[0];
}

// Ginv = G^-1 for B symmetric positive definite k x k matrices (G = Phi2^T Phi2, well conditioned: the basis is
// mass-orthonormal).  k <= 176: k unit right-hand sides on the blocked LDS Cholesky (img = scratch for its blocked image);
// 177 <= k <= 256: Newton-Schulz  X <- 2 X - (X G) X  on the float64 matrix cores (Gx, Gt = two k x k scratch matrices per
// pair; every iterate is a polynomial in G, hence symmetric and commuting with G); the error 1 - lambda(X G) squares per
// step from 1 - 1/cond^2 at the start: 2 log2(cond) + 6 steps, 36 cover cond up to 3e4; a pair that did not converge is
// reported in info (k + 1), a Gram matrix that is not positive definite as column + 1.
static size_t gram_inverse_ws_bytes(int B, int k) {
    const int NB = (k + 15) / 16;
    return NB <= 11 ? dm_align_up((size_t)B * (NB * (NB + 1) / 2) * 256 * 8) : 2 * dm_align_up((size_t)B * k * k * 8);
}
static int gram_inverse(dm_ctx* ctx, int B, int k2, const double* G, double* Ginv, double* img, double* Gx, double* Gt, int32_t* info) {
    const int NB = (k2 + 15) / 16, nblk = NB * (NB + 1) / 2;
    const bool chol = NB <= 11;
    if (chol) {
        DM_LAUNCH(ctx, "blockify", blockify_kernel, dim3(nblk, B), dim3(256), 0, G, k2, NB, img);
        const size_t lds = ((size_t)(nblk + 2) * 256 + 2 * NB * 16 + 16 + 8) * sizeof(double);
        int rc = dm_grant_lds(ctx, (const void*)spd_multi_rhs_kernel, lds);
        if (rc) return rc;
        DM_LAUNCH(ctx, "icp_normal_eq_chol", spd_multi_rhs_kernel, dim3(k2, B), dim3(256), lds, img, (const double*)nullptr, k2, k2, NB,
                  Ginv, info);
        return DM_OK;
    }
    constexpr int NS_INV = 36;
    DM_LAUNCH(ctx, "ns_inverse_init", ns_inverse_init_kernel, dim3(B), dim3(256), 0, G, k2, Ginv);
    double* xo = Ginv;
    double* xn = Gx;
    const dim3 grid(dm_cdiv(k2, NT_T) * dm_cdiv(k2, NT_T), 1, B);
    // (the second operand is read transposed: O = A B^T in the kernel, and writing X G as X G^T / T X as T X^T would double
    //  the antisymmetric rounding error of X at every step once the iteration has converged)
    for (int q = 0; q < NS_INV; ++q) {
        KRowsF64 x_{xo, (long long)k2 * k2, k2, k2, k2, 0};
        KRowsF64 gT{G, (long long)k2 * k2, k2, k2, k2, 1};
        OutPlainNT ot{Gt, (long long)k2 * k2, k2};
        DM_LAUNCH(ctx, "ns_inverse_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutPlainNT>), grid, dim3(256), 0, x_, gT, ot, k2, k2, k2);
        KRowsF64 t_{Gt, (long long)k2 * k2, k2, k2, k2, 0};
        KRowsF64 xT{xo, (long long)k2 * k2, k2, k2, k2, 1};
        OutAxpby on{xo, xn, (long long)k2 * k2, k2, 2.0, -1.0};
        DM_LAUNCH(ctx, "ns_inverse_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutAxpby>), grid, dim3(256), 0, t_, xT, on, k2, k2, k2);
        double* tmp = xo; xo = xn; xn = tmp;
    }
    // (NS_INV is even: the result is back in Ginv)
    KRowsF64 x_{Ginv, (long long)k2 * k2, k2, k2, k2, 0};
    KRowsF64 gT{G, (long long)k2 * k2, k2, k2, k2, 1};
    OutPlainNT ot{Gt, (long long)k2 * k2, k2};
    DM_LAUNCH(ctx, "ns_inverse_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutPlainNT>), grid, dim3(256), 0, x_, gT, ot, k2, k2, k2);
    DM_LAUNCH(ctx, "ns_inverse_check", ns_inverse_check_kernel, dim3(B), dim3(256), 0, Gt, k2, info);
    return DM_OK;
}

// ---- least-squares vertex map -> functional map ------------------------------------------------------------------
// C = argmin |Phi2[:, :k2] X - Phi1[p21, :k1]|_F  (no mass): normal equations (Phi2^T Phi2) C = Phi2^T Phi1[p21].
template <typename TR>
static int p2p_to_fm_lstsq_impl(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21, const TR* Phi1,
                                int ld1, const TR* Phi2, int ld2, double* C, int32_t* info) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N1 > 0 && N2 > 0 && k1 > 0 && k2 > 0, "sizes must be positive");
    DM_REQUIRE(ctx, p21 && Phi1 && Phi2 && C && info, "null pointer");
    DM_REQUIRE(ctx, ld1 >= k1 && ld2 >= k2, "eigenvector row stride smaller than the map size");
    DM_REQUIRE(ctx, k2 <= 256, "the least-squares map needs k2 <= 256");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bG = (size_t)B * k2 * k2 * 8, bC = (size_t)B * k2 * k1 * 8;
    const int NB = (k2 + 15) / 16;
    const size_t need = 2 * dm_align_up(bG) + dm_align_up(bC) + gram_inverse_ws_bytes(B, k2) + 3 * dm_align_up((size_t)B * N2 * 4) +
                        dm_p2pfm_ws_bytes(B, N2, max(k1, k2), k2) + 65536;
    int rc = dm_ws_reserve(ctx, need);
    if (rc) return rc;
    double* G = (double*)dm_ws_take(ctx, bG);
    double* Ginv = (double*)dm_ws_take(ctx, bG);
    double* R = (double*)dm_ws_take(ctx, bC);
    double* img = nullptr; double* Gx = nullptr; double* Gt = nullptr;
    if (NB <= 11) img = (double*)dm_ws_take(ctx, (size_t)B * (NB * (NB + 1) / 2) * 256 * 8);
    else { Gx = (double*)dm_ws_take(ctx, bG); Gt = (double*)dm_ws_take(ctx, bG); }
    int32_t* iota = (int32_t*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    double* ones = (double*)dm_ws_take(ctx, (size_t)B * N2 * 8);
    if (!G || !Ginv || !R || (NB <= 11 ? !img : (!Gx || !Gt)) || !iota || !ones) return dm_fail(ctx, DM_ENOMEM, "p2p_to_fm_lstsq: workspace not reserved");
    DM_CHECK_HIP(ctx, hipMemsetAsync(info, 0, (size_t)B * 4, ctx->stream));
    DM_LAUNCH(ctx, "iota_ones", iota_ones_kernel, dim3((unsigned)(((long long)B * N2 + 255) / 256)), dim3(256), 0, iota, ones, N2, B);
    const size_t ws_mark = ctx->ws_off;
    rc = dm_launch_p2p_to_fm<TR>(ctx, B, N2, N2, k2, k2, iota, Phi2, ld2, Phi2, ld2, ones, G, k2, (long long)k2 * k2);
    if (rc) return rc;
    rc = gram_inverse(ctx, B, k2, G, Ginv, img, Gx, Gt, info);
    if (rc) return rc;
    ctx->ws_off = ws_mark;
    rc = dm_launch_p2p_to_fm<TR>(ctx, B, N1, N2, k1, k2, p21, Phi1, ld1, Phi2, ld2, ones, R, k1, (long long)k2 * k1);
    if (rc) return rc;
    KRowsF64 ga{Ginv, (long long)k2 * k2, k2, k2, k2, 0};
    KRowsF64 rb{R, (long long)k2 * k1, k1, k1, k2, 1};
    OutPlainNT oc{C, (long long)k2 * k1, k1};
    DM_LAUNCH(ctx, "icp_apply_inverse_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutPlainNT>),
              dim3(dm_cdiv(k2, NT_T) * dm_cdiv(k1, NT_T), 1, B), dim3(256), 0, ga, rb, oc, k2, k1, k2);
    // one step of iterative refinement on the normal equations, C += G^-1 (R - G C): removes what the explicit inverse lost
    // (Cholesky on unit right-hand sides / Newton-Schulz); the conditioning of G = Phi2^T Phi2 itself (cond(Phi2)^2) remains
    {
        KRowsF64 gg{G, (long long)k2 * k2, k2, k2, k2, 0};
        KRowsF64 cT{C, (long long)k2 * k1, k1, k1, k2, 1};
        OutAxpby orho{R, R, (long long)k2 * k1, k1, 1.0, -1.0};                 // R <- R - G C
        DM_LAUNCH(ctx, "lstsq_residual_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutAxpby>),
                  dim3(dm_cdiv(k2, NT_T) * dm_cdiv(k1, NT_T), 1, B), dim3(256), 0, gg, cT, orho, k2, k1, k2);
        KRowsF64 rT{R, (long long)k2 * k1, k1, k1, k2, 1};
        OutAxpby ocx{C, C, (long long)k2 * k1, k1, 1.0, 1.0};                   // C <- C + G^-1 rho
        DM_LAUNCH(ctx, "icp_apply_inverse_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutAxpby>),
                  dim3(dm_cdiv(k2, NT_T) * dm_cdiv(k1, NT_T), 1, B), dim3(256), 0, ga, rT, ocx, k2, k1, k2);
    }
    return DM_OK;
}
extern "C" int dm_p2p_to_fm_lstsq(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21, const float* Phi1,
                                  int ld1, const float* Phi2, int ld2, double* C, int32_t* info) {
    return p2p_to_fm_lstsq_impl<float>(ctx, B, N1, N2, k1, k2, p21, Phi1, ld1, Phi2, ld2, C, info);
}
extern "C" int dm_p2p_to_fm_lstsq_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21, const double* Phi1,
                                      int ld1, const double* Phi2, int ld2, double* C, int32_t* info) {
    return p2p_to_fm_lstsq_impl<double>(ctx, B, N1, N2, k1, k2, p21, Phi1, ld1, Phi2, ld2, C, info);
}

template <typename TR>
static int icp_impl(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const TR* Phi1, int ld1, const TR* Phi2,
                    int ld2, const double* C0, int nit, double* Cout, double* resid, int32_t* info) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N1 > 0 && N2 > 0 && k1 > 0 && k2 > 0 && nit >= 0, "sizes must be positive");
    DM_REQUIRE(ctx, Phi1 && Phi2 && C0 && Cout && info, "null pointer");
    DM_REQUIRE(ctx, ld1 >= k1 && ld2 >= k2, "eigenvector row stride smaller than the map size");
    DM_REQUIRE(ctx, k1 <= 256 && k2 <= 256, "ICP on the GPU needs k1, k2 <= 256");
    DM_REQUIRE(ctx, k2 >= k1, "the polar factor U eye(k2,k1) V^T needs k2 >= k1");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));

    const int N1pad = pad_to(N1, 128), N2pad = pad_to(N2, 128), Kpad = pad_to(k2, 16);
    const int NB = (k2 + 15) / 16, nblk = NB * (NB + 1) / 2;
    const bool chol = NB <= 11;                 // the blocked LDS Cholesky holds the Gram matrix up to k2 = 176 (gram_inverse)
    const size_t bAT = (size_t)B * Kpad * N2pad * 8, bBT = (size_t)B * Kpad * N1pad * 8;
    const size_t bC = (size_t)B * k2 * k1 * 8, bG = (size_t)B * k2 * k2 * 8, bImg = (size_t)B * nblk * 256 * 8;
    const size_t bT = (size_t)B * k1 * k1 * 8;
    const size_t need = dm_align_up(bAT) + dm_align_up(bBT) + 4 * dm_align_up(bC) + 4 * dm_align_up(bG) + dm_align_up(bImg) +
                        2 * dm_align_up(bT) + dm_align_up((size_t)B * N1pad * 8) + 4 * dm_align_up((size_t)B * N2 * 4) +
                        dm_gred_ws_bytes(B, N2, N1) + dm_knn_split_prep_bytes(B, N2, k2) + dm_knn_split_ws_bytes(B, N2, N1, k2) +
                        dm_align_up((size_t)B * (N1pad / DM_EMB_COLS + 1) * 8) + dm_p2pfm_ws_bytes(B, N2, max(k1, k2), k2) + dm_p2pfm_xs_bytes(B, N2, k2) + 65536;
    int rc = dm_ws_reserve(ctx, need);
    if (rc) return rc;
    double* AT = (double*)dm_ws_take(ctx, bAT);
    double* BT = (double*)dm_ws_take(ctx, bBT);
    double* Ccur = (double*)dm_ws_take(ctx, bC);
    double* R = (double*)dm_ws_take(ctx, bC);
    double* Xa = (double*)dm_ws_take(ctx, bC);
    double* Xb = (double*)dm_ws_take(ctx, bC);
    double* G = (double*)dm_ws_take(ctx, bG);
    double* Ginv = (double*)dm_ws_take(ctx, bG);
    double* img = (double*)dm_ws_take(ctx, bImg);
    double* Tm = (double*)dm_ws_take(ctx, bT);
    double* Wm = (double*)dm_ws_take(ctx, bT);
    double* n1 = (double*)dm_ws_take(ctx, (size_t)B * N1pad * 8);
    int32_t* p21 = (int32_t*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    int32_t* iota = (int32_t*)dm_ws_take(ctx, (size_t)B * N2 * 4);
    double* ones = (double*)dm_ws_take(ctx, (size_t)B * N2 * 8);
    double* amaxS = (double*)dm_ws_take(ctx, (size_t)B * (N1pad / DM_EMB_COLS + 1) * 8);
    double* Gx = chol ? nullptr : (double*)dm_ws_take(ctx, bG);
    double* Gt = chol ? nullptr : (double*)dm_ws_take(ctx, bG);
    if (!AT || !BT || !Ccur || !R || !Xa || !Xb || !G || !Ginv || !img || !Tm || !Wm || !n1 || !p21 || !iota || !ones || !amaxS ||
        (!chol && (!Gx || !Gt)))
        return dm_fail(ctx, DM_ENOMEM, "icp: workspace not reserved");

    DM_CHECK_HIP(ctx, hipMemsetAsync(info, 0, (size_t)B * 4, ctx->stream));
    DM_CHECK_HIP(ctx, hipMemcpyAsync(Ccur, C0, bC, hipMemcpyDeviceToDevice, ctx->stream));
    DM_LAUNCH(ctx, "iota_ones", iota_ones_kernel, dim3((unsigned)(((long long)B * N2 + 255) / 256)), dim3(256), 0, iota, ones, N2, B);
    // iteration independent: Phi2^T (K-major f64) and the Gram matrix Phi2^T Phi2 with its blocked image
    rc = dm_launch_phiT<TR>(ctx, B, N2, k2, Phi2, ld2, AT, Kpad, N2pad);
    if (rc) return rc;
    dm_knn_split_state knn;                    // fp16 split of Phi2 for the nearest-neighbour searches, once
    rc = dm_knn_split_prepare(ctx, B, N2, N2pad, Kpad, k2, AT, &knn);
    if (rc) return rc;
    // the left operand of every p2p_to_FM of the call (Phi2, unit masses), once (dm_p2pfm_prescale)
    double* Xs = nullptr;
    const int ldx = (k2 + 15) / 16 * 16;
    if (ctx->opt_p2pfm_direct) {
        Xs = (double*)dm_ws_take(ctx, dm_p2pfm_xs_bytes(B, N2, k2));
        if (!Xs) return dm_fail(ctx, DM_ENOMEM, "icp: workspace not reserved");
        rc = dm_p2pfm_prescale<TR>(ctx, B, N2, k2, Phi2, ld2, ones, Xs);
        if (rc) return rc;
    }
    const size_t ws_mark = ctx->ws_off;
    rc = dm_launch_p2p_to_fm<TR>(ctx, B, N2, N2, k2, k2, iota, Phi2, ld2, Phi2, ld2, ones, G, k2, (long long)k2 * k2, Xs, ldx);
    if (rc) return rc;
    DM_CHECK_HIP(ctx, hipMemsetAsync(BT, 0, bBT, ctx->stream));
    // the Gram matrix does not change over the iterations: invert it once and apply the inverse by a GEMM per iteration
    rc = gram_inverse(ctx, B, k2, G, Ginv, img, Gx, Gt, info);
    if (rc) return rc;

    for (int it = 0; it < nit; ++it) {
        ctx->ws_off = ws_mark;
        // p21 = NN(tree = Phi1 C^T, query = Phi2)
        rc = dm_launch_embed<TR>(ctx, B, N1, k2, k1, Phi1, ld1, Ccur, k1, (long long)k2 * k1, 0, BT, Kpad, N1pad, n1, 0, amaxS);
        if (rc) return rc;
        dm_gred_args a;
        a.B = B; a.N2 = N2; a.N1 = N1; a.Kloop = Kpad;
        a.AT = AT; a.N2pad = N2pad; a.BT = BT; a.N1pad = N1pad; a.Kpad = Kpad;
        a.n1 = n1; a.n2 = nullptr; a.mass1 = nullptr;
        a.knn21 = p21; a.knn12 = nullptr; a.ind21 = nullptr; a.ind12 = nullptr;
        a.Ktrue = k2;
        rc = dm_launch_knn21(ctx, a, knn, amaxS, dm_cdiv(N1pad, DM_EMB_COLS));
        if (rc) return rc;
        // R = Phi2^T Phi1[p21]   (k2 x k1);   Chat = (Phi2^T Phi2)^-1 R
        rc = dm_launch_p2p_to_fm<TR>(ctx, B, N1, N2, k1, k2, p21, Phi1, ld1, Phi2, ld2, ones, R, k1, (long long)k2 * k1, Xs, ldx);
        if (rc) return rc;
        {   // Chat = (Phi2^T Phi2)^-1 R with the inverse computed once before the loop
            KRowsF64 ga{Ginv, (long long)k2 * k2, k2, k2, k2, 0};
            KRowsF64 rb{R, (long long)k2 * k1, k1, k1, k2, 1};
            OutPlainNT oc{Xa, (long long)k2 * k1, k1};
            DM_LAUNCH(ctx, "icp_apply_inverse_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutPlainNT>),
                      dim3(dm_cdiv(k2, NT_T) * dm_cdiv(k1, NT_T), 1, B), dim3(256), 0, ga, rb, oc, k2, k1, k2);
        }
        // polar factor of Chat by Newton-Schulz
        DM_LAUNCH(ctx, "polar_scale", polar_scale_kernel, dim3(B), dim3(256), 0, Xa, k2, k1);
        double* xo = Xa;
        double* xn = Xb;
        for (int q = 0; q < NS_LIFT + NS_POLISH; ++q) {
            const bool lift = q < NS_LIFT;
            RowsF64 opx{xo, (long long)k2 * k1, k1, k1};
            OutPlainTN ot{Tm, (long long)k1 * k1, k1};
            DM_LAUNCH(ctx, "polar_xtx_tn_f64", (gemm_tn_f64<RowsF64, RowsF64, OutPlainTN>), dim3(dm_cdiv(k1, TN_T) * dm_cdiv(k1, TN_T), 1, B),
                      dim3(256), 0, opx, opx, ot, k1, k1, k2, pad_to(k2, TN_BK));
            if (lift) {
                // W = b T + c T T^T   (T is symmetric)
                KRowsF64 ta{Tm, (long long)k1 * k1, k1, k1, k1, 0};
                OutAxpby ow{Tm, Wm, (long long)k1 * k1, k1, NS_B, NS_C};
                DM_LAUNCH(ctx, "polar_poly_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutAxpby>),
                          dim3(dm_cdiv(k1, NT_T) * dm_cdiv(k1, NT_T), 1, B), dim3(256), 0, ta, ta, ow, k1, k1, k1);
            }
            KRowsF64 opa{xo, (long long)k2 * k1, k1, k2, k1, 0};
            KRowsF64 opb{lift ? Wm : Tm, (long long)k1 * k1, k1, k1, k1, 0};
            OutAxpby on{xo, xn, (long long)k2 * k1, k1, lift ? NS_A : 1.5, lift ? 1.0 : -0.5};
            DM_LAUNCH(ctx, "polar_update_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutAxpby>),
                      dim3(dm_cdiv(k2, NT_T) * dm_cdiv(k1, NT_T), 1, B), dim3(256), 0, opa, opb, on, k2, k1, k1);
            double* tmp = xo; xo = xn; xn = tmp;
        }
        DM_CHECK_HIP(ctx, hipMemcpyAsync(Ccur, xo, bC, hipMemcpyDeviceToDevice, ctx->stream));
        if (resid && it == nit - 1) {
            RowsF64 opx{xo, (long long)k2 * k1, k1, k1};
            OutPlainTN ot{Tm, (long long)k1 * k1, k1};
            DM_LAUNCH(ctx, "polar_xtx_tn_f64", (gemm_tn_f64<RowsF64, RowsF64, OutPlainTN>), dim3(dm_cdiv(k1, TN_T) * dm_cdiv(k1, TN_T), 1, B),
                      dim3(256), 0, opx, opx, ot, k1, k1, k2, pad_to(k2, TN_BK));
            DM_LAUNCH(ctx, "ortho_resid", ortho_resid_kernel, dim3(B), dim3(256), 0, Tm, k1, resid);
        }
    }
    if (resid && nit == 0) DM_CHECK_HIP(ctx, hipMemsetAsync(resid, 0, (size_t)B * 8, ctx->stream));
    DM_CHECK_HIP(ctx, hipMemcpyAsync(Cout, Ccur, bC, hipMemcpyDeviceToDevice, ctx->stream));
    return DM_OK;
}
extern "C" int dm_icp(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const float* Phi1, int ld1, const float* Phi2,
                      int ld2, const double* C0, int nit, double* Cout, double* resid, int32_t* info) {
    return icp_impl<float>(ctx, B, N1, N2, k1, k2, Phi1, ld1, Phi2, ld2, C0, nit, Cout, resid, info);
}
extern "C" int dm_icp_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const double* Phi1, int ld1, const double* Phi2,
                          int ld2, const double* C0, int nit, double* Cout, double* resid, int32_t* info) {
    return icp_impl<double>(ctx, B, N1, N2, k1, k2, Phi1, ld1, Phi2, ld2, C0, nit, Cout, resid, info);
}

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
// r05: the schedule is per pair and follows what the iteration measures on the T = X^T X every step computes anyway (polar_decide:
// one workgroup per pair, T in the LDS):
//   * first step: s^2 = |T|_inf >= sigma_max^2 is the scaling, folded into the step's coefficients (the row-sum norm of T is far
//     tighter than sqrt(|X|_1 |X|_inf), which overestimates sigma_max of a 128 x 128 orthogonal matrix nine times);
//   * LIFT while rho = |T - I|_2 > 0.62, estimated from below by a few power iterations per step whose vector is kept from step to
//     step (T's eigenvectors are X's right singular vectors: the same at every step; a singular value still far below 1 is an
//     eigenvalue ~1 of I - T next to others below 0.54: it dominates after a handful of iterations);
//   * then Newton-Schulz until r = |T - I|_inf (>= rho: a safe bound) is below 3e-8: that step is the last (its error
//     1.5 (r / 2)^2 = 3e-16); nothing runs for the pair after it.  An optimistic rho costs Newton-Schulz steps (x 1.5 per step,
//     always convergent below sqrt 3), never the result; NS_MAX_STEPS caps the schedule.
//   The host stops launching once no pair is left, and launches the lifts' T^2 product only while some pair lifts (lagged reads of
//   two words through page-locked memory: the queue never drains).  A well-conditioned Chat takes 0-1 lifts + 5-6 polish steps
//   (20 launches) instead of 12 + 6 (49); an ill-conditioned one as many lifts as it needs (481 of 570 launches).
#include "dm_chol.h"
#include "dm_gemm_f64.h"
#include "dm_internal.h"

static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

constexpr int NS_LIFT = 12, NS_POLISH = 6;
constexpr int NS_POWER0 = 6, NS_POWER = 3;                 // power iterations of the first / of a later lifting step
constexpr double NS_RHO = 0.62;                            // lift while |T - I|_2 is (estimated) above
constexpr int NS_MAX_STEPS = NS_LIFT + 28;                 // cap of the schedule
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
// ---- the polar iteration's per-pair schedule --------------------------------------------------------------------
// a step is  X <- a X + b X M,  M = T = X^T X (Newton-Schulz) or M = W = wb T + wc T^2 (lift)
struct polar_state { double a, b, wb, wc; int mode, lift, lifting, count; };   // mode 0: step, 1: step and stop, 2: done; lift: M = W this step
// One workgroup per pair, after the T of the step: the step's coefficients.  T (k x k, k <= 256) is copied to the LDS when it fits
// (lds_T != 0; else it is read where it is); vecs: (B, 256) the pair's power-iteration vector.  counts[0] += pairs that step,
// counts[1] += pairs that lift at this step.
__global__ __launch_bounds__(256) void polar_decide_kernel(const double* __restrict__ Tm, int k, polar_state* __restrict__ st,
                                                           double* __restrict__ vecs, int32_t* __restrict__ counts, int lds_T) {
    extern __shared__ __attribute__((aligned(16))) double pd_smem[];
    __shared__ double red[8], vec[2][256];
    const int b = blockIdx.x, t = threadIdx.x;
    // Everything the kernel reads from memory is requested at once, before the first use: T (into registers, on its way to the LDS), the
    // pair's vector and its state -- one round trip instead of three dependent ones (each ~2 us on this part; the kernel's work is ~5).
    const double* Tg = Tm + (long long)b * k * k;
    const int n = k * k, n2 = n >> 1;
    const bool packed = lds_T && (n & 1) == 0;                 // (uniform)
    constexpr int PRE = 36;                                    // 36 x 256 x 16 bytes = 144 KiB: every T that fits the LDS
    f64x2 tx[PRE];
    if (packed) {
        const f64x2* src = reinterpret_cast<const f64x2*>(Tg);
#pragma unroll
        for (int u = 0; u < PRE; ++u) { const int e = 256 * u + t; tx[u] = src[e < n2 ? e : n2 - 1]; }
    }
    const double gv_t = vecs[(long long)b * 256 + t];
    const polar_state old = st[b];                             // (uniform)
    if (old.mode >= 1) {                                       // its last step ran in the previous round (or it was done already)
        if (t == 0 && old.mode == 1) st[b] = polar_state{1.0, 0.0, 0.0, 0.0, 2, 0, 0, old.count};
        return;
    }
    const double* T = Tg;
    if (lds_T) {                                               // (uniform)
        if (packed) {
            f64x2* dst = reinterpret_cast<f64x2*>(pd_smem);
#pragma unroll
            for (int u = 0; u < PRE; ++u) { const int e = 256 * u + t; if (e < n2) dst[e] = tx[u]; }
        } else {
            for (int e = t; e < n; e += 256) pd_smem[e] = Tg[e];
        }
        __syncthreads();
        T = pd_smem;
    }
    const bool first = old.count == 0;
    // column sums (T is symmetric) of |T| (first step) or |T - I|: thread (j, half of the rows); consecutive threads read consecutive words
    const int j = t & 127, half = t >> 7;
    const int ibeg = half * ((k + 1) / 2), iend = half ? k : (k + 1) / 2;
    // (columns past k read column k - 1 and are dropped at the end: no branch in the loops, eight independent reads per pass)
    const int ja = min(j, k - 1), jb = min(j + 128, k - 1);
    const double dsub = first ? 0.0 : 1.0;
    double c0 = 0.0, c1 = 0.0;
    {
        int i = ibeg;
        for (; i + 8 <= iend; i += 8) {
            double xa[8], xb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { xa[u] = T[(i + u) * k + ja]; xb[u] = T[(i + u) * k + jb]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { c0 += fabs(xa[u] - ((i + u == ja) ? dsub : 0.0)); c1 += fabs(xb[u] - ((i + u == jb) ? dsub : 0.0)); }
        }
        for (; i < iend; ++i) { c0 += fabs(T[i * k + ja] - ((i == ja) ? dsub : 0.0)); c1 += fabs(T[i * k + jb] - ((i == jb) ? dsub : 0.0)); }
        if (j >= k) c0 = 0.0;
        if (j + 128 >= k) c1 = 0.0;
    }
    if (half == 1) { vec[0][j] = c0; vec[1][j] = c1; }
    __syncthreads();
    double m = 0.0;
    if (half == 0) { m = fmax(c0 + vec[0][j], c1 + vec[1][j]); }
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    const double rinf = fmax(red[0], red[1]);                  // |T|_inf (first step) or |T - I|_inf
    __syncthreads();
    if (first && !(rinf > 0.0)) {                              // X = 0 stays 0 (the residual reports it); (uniform)
        if (t == 0) st[b] = polar_state{1.0, 0.0, 0.0, 0.0, 2, 0, 0, 1};
        return;
    }
    const double inv_s2 = first ? 1.0 / rinf : 1.0;            // E = I - T / s^2
    bool lifting = first || old.lifting;                       // (uniform)
    double rho = 0.0;
    if (lifting) {
        // power iteration on E, from the pair's vector of the previous step
        double* gv = vecs + (long long)b * 256;
        vec[0][t] = first ? ((t < k) ? 1.0 + 0.37 * ((t * 7) % 5) : 0.0) : gv_t;
        __syncthreads();
        const int nit = first ? NS_POWER0 : NS_POWER;
        for (int it = 0; it < nit; ++it) {
            const int cur = it & 1;
            double y0 = 0.0, y1 = 0.0;
            {
                int i = ibeg;
                for (; i + 8 <= iend; i += 8) {
                    double xa[8], xb[8], vi[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { xa[u] = T[(i + u) * k + ja]; xb[u] = T[(i + u) * k + jb]; vi[u] = vec[cur][i + u]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) { y0 = fma(xa[u], vi[u], y0); y1 = fma(xb[u], vi[u], y1); }
                }
                for (; i < iend; ++i) { const double vi = vec[cur][i]; y0 = fma(T[i * k + ja], vi, y0); y1 = fma(T[i * k + jb], vi, y1); }
                if (j >= k) y0 = 0.0;
                if (j + 128 >= k) y1 = 0.0;
            }
            if (half == 1) { vec[cur ^ 1][j] = y0; if (j + 128 < 256) vec[cur ^ 1][j + 128] = y1; }
            __syncthreads();
            if (half == 0) { vec[cur ^ 1][j] += y0; if (j + 128 < 256) vec[cur ^ 1][j + 128] += y1; }
            __syncthreads();
            const double v_t = t < k ? vec[cur][t] : 0.0;
            const double w_t = t < k ? v_t - inv_s2 * vec[cur ^ 1][t] : 0.0;       // (E v)_t
            double p1 = v_t * v_t, p2 = w_t * w_t;
            for (int off = 32; off > 0; off >>= 1) { p1 += __shfl_xor(p1, off); p2 += __shfl_xor(p2, off); }
            __syncthreads();
            if ((t & 63) == 0) { red[t >> 6] = p1; red[4 + (t >> 6)] = p2; }
            __syncthreads();
            const double vn = (red[0] + red[1]) + (red[2] + red[3]), wn = (red[4] + red[5]) + (red[6] + red[7]);
            rho = vn > 0.0 ? sqrt(wn / vn) : 0.0;                                  // |E v| / |v| <= |E|_2
            vec[cur ^ 1][t] = wn > 0.0 ? w_t / sqrt(wn) : v_t;
            __syncthreads();
        }
        gv[t] = vec[nit & 1][t];
        lifting = rho > NS_RHO;
    }
    if (t == 0) {
        const double is = first ? sqrt(inv_s2) : 1.0, is2 = is * is;             // 1 / s
        polar_state s;
        if (lifting) s = polar_state{NS_A * is, 1.0, NS_B * is * is2, NS_C * is * is2 * is2, 0, 1, 1, old.count + 1};
        else {
            s = polar_state{1.5 * is, -0.5 * is * is2, 0.0, 0.0, 0, 0, 0, old.count + 1};
            if (!first && rinf < 1e-14) s = polar_state{1.0, 0.0, 0.0, 0.0, 2, 0, 0, old.count};
            else if (!first && rinf < 3e-8) s.mode = 1;
        }
        st[b] = s;
        if (s.mode != 2) atomicAdd(&counts[0], 1);
        if (s.lift) atomicAdd(&counts[1], 1);
    }
}
struct OutPolarT {                     // T = X^T X; nothing to do for a pair that has taken its last step
    double* p; long long stride_b; int ld; const polar_state* st;
    __device__ __forceinline__ bool skip(int b) const { return st[b].mode >= 1; }
    __device__ __forceinline__ void store(int b, int, int m, int c, double v) const { p[b * stride_b + (long long)m * ld + c] = v; }
};
struct OutPolarW {                     // W = wb T + wc T T for the pairs that lift
    const double* tm; double* wm; long long stride_b; int ld; const polar_state* st;
    __device__ __forceinline__ bool skip(int b) const { return st[b].lift == 0; }
    __device__ __forceinline__ void store_skipped(int, int, int) const {}
    __device__ __forceinline__ void store(int b, int i, int j, double v) const {
        const long long o = b * stride_b + (long long)i * ld + j;
        wm[o] = st[b].wb * tm[o] + st[b].wc * v;
    }
};
struct OutPolarX {                     // Xnew = a Xold + b Xold M; a finished pair's X is carried over to the other buffer
    const double* xo; double* xn; long long stride_b; int ld; const polar_state* st;
    __device__ __forceinline__ bool skip(int b) const { return st[b].mode == 2; }
    __device__ __forceinline__ void store_skipped(int b, int i, int j) const {
        const long long o = b * stride_b + (long long)i * ld + j;
        xn[o] = xo[o];
    }
    __device__ __forceinline__ void store(int b, int i, int j, double v) const {
        const long long o = b * stride_b + (long long)i * ld + j;
        xn[o] = st[b].a * xo[o] + st[b].b * v;
    }
};
struct KRowsPolarM {                   // the step's M: W for a pair that lifts, T otherwise (same interface as KRowsF64)
    KRowsF64 t, w; const polar_state* st;
    __device__ __forceinline__ const KRowsF64& of(int b) const { return st[b].lift ? w : t; }
    __device__ __forceinline__ bool fast_ok(int K, int rows_used) const { return t.fast_ok(K, rows_used) && w.fast_ok(K, rows_used); }
    __device__ __forceinline__ void load8_fast(int b, int row, int k0, double (&v)[8]) const { of(b).load8_fast(b, row, k0, v); }
    __device__ __forceinline__ double cvt(const double& x, int) const { return x; }
    __device__ __forceinline__ void load8(int b, int row, int k0, double (&v)[8]) const { of(b).load8(b, row, k0, v); }
};
// Maps up to 32 x 32 (the documented call: n_ev = 15 ... 30): the WHOLE polar iteration of a pair in one workgroup -- X, T, W in the LDS,
// the same schedule as polar_decide (scaling by |T|_inf at the first step, lifts while the estimated |T - I|_2 > NS_RHO, Newton-Schulz
// until |T - I|_inf < 3e-8, then one last step), decisions free of launches: an ICP iteration's ~20 launches become one.  256 threads:
// thread t owns the entries e = t, t + 256, ... of every 32 x 32 product (ascending sums over the contraction index).
__global__ __launch_bounds__(256) void polar_small_kernel(const double* __restrict__ Xin, double* __restrict__ Xout, int k2, int k1) {
    constexpr int LD = 33;
    __shared__ double Xs[2][32 * LD], Ts[32 * LD], Ws[32 * LD], vec[32], bc[8];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const double* Xg = Xin + (long long)b * k2 * k1;
    for (int e = t; e < 32 * 32; e += 256) {
        const int r = e >> 5, c = e & 31;
        Xs[0][r * LD + c] = (r < k2 && c < k1) ? Xg[r * k1 + c] : 0.0;
    }
    if (t < 32) vec[t] = (t < k1) ? 1.0 + 0.37 * ((t * 7) % 5) : 0.0;
    __syncthreads();
    int cur = 0, lifts = 0;
    bool lifting = true;                                       // (uniform: decided from LDS broadcasts)
    for (int step = 0; step < NS_MAX_STEPS; ++step) {
        const double* X = Xs[cur];
        // T = X^T X
        for (int e = t; e < 32 * 32; e += 256) {
            const int a = e >> 5, c = e & 31;
            double acc = 0.0;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) acc = fma(X[r * LD + a], X[r * LD + c], acc);
            Ts[a * LD + c] = acc;
        }
        __syncthreads();
        const bool first = step == 0;
        // wave 0: row sums, power iteration, the step's coefficients -> bc[0..5] = a, b, wb, wc, mode (0 step, 1 last, 2 done), lift
        if (t < 64) {
            double rs = 0.0;
            if (lane < k1) for (int c = 0; c < k1; ++c) rs += fabs(Ts[lane * LD + c] - ((!first && c == lane) ? 1.0 : 0.0));
            double rinf = rs;
            for (int off = 32; off > 0; off >>= 1) rinf = fmax(rinf, __shfl_xor(rinf, off));
            double a_ = 1.0, b_ = 0.0, wb = 0.0, wc = 0.0;
            int mode = 0, lift = 0;
            if (first && !(rinf > 0.0)) mode = 2;
            else {
                const double inv_s2 = first ? 1.0 / rinf : 1.0;
                bool lf = lifting;
                if (lf) {
                    double v = lane < 32 ? vec[lane] : 0.0, rho = 0.0;
                    const int nit = first ? NS_POWER0 : NS_POWER;
                    for (int it = 0; it < nit; ++it) {
                        double y = 0.0;
                        if (lane < k1) for (int c = 0; c < k1; ++c) y = fma(Ts[lane * LD + c], __shfl(v, c), y);
                        const double wv = lane < k1 ? v - inv_s2 * y : 0.0;
                        double p1 = v * v, p2 = wv * wv;
                        for (int off = 32; off > 0; off >>= 1) { p1 += __shfl_xor(p1, off); p2 += __shfl_xor(p2, off); }
                        rho = p1 > 0.0 ? sqrt(p2 / p1) : 0.0;
                        v = p2 > 0.0 ? wv / sqrt(p2) : v;
                    }
                    if (lane < 32) vec[lane] = v;
                    lf = rho > NS_RHO && lifts < NS_LIFT;
                }
                const double is = first ? sqrt(inv_s2) : 1.0, is2 = is * is;
                if (lf) { a_ = NS_A * is; b_ = 1.0; wb = NS_B * is * is2; wc = NS_C * is * is2 * is2; lift = 1; }
                else {
                    a_ = 1.5 * is; b_ = -0.5 * is * is2;
                    if (!first && rinf < 1e-14) mode = 2;
                    else if (!first && rinf < 3e-8) mode = 1;
                }
            }
            if (lane == 0) { bc[0] = a_; bc[1] = b_; bc[2] = wb; bc[3] = wc; bc[4] = (double)mode; bc[5] = (double)lift; }
        }
        __syncthreads();
        const double ca = bc[0], cb = bc[1], cwb = bc[2], cwc = bc[3];
        const int mode = (int)bc[4];
        const bool lift = bc[5] != 0.0;
        lifting = lift;
        lifts += lift ? 1 : 0;
        if (mode == 2) break;                                  // (uniform)
        const double* M = Ts;
        if (lift) {                                            // W = wb T + wc T T
            for (int e = t; e < 32 * 32; e += 256) {
                const int a = e >> 5, c = e & 31;
                double acc = 0.0;
#pragma unroll 8
                for (int j = 0; j < 32; ++j) acc = fma(Ts[a * LD + j], Ts[j * LD + c], acc);
                Ws[a * LD + c] = cwb * Ts[a * LD + c] + cwc * acc;
            }
            __syncthreads();
            M = Ws;
        }
        double* Xn = Xs[cur ^ 1];
        for (int e = t; e < 32 * 32; e += 256) {
            const int r = e >> 5, c = e & 31;
            double acc = 0.0;
#pragma unroll 8
            for (int j = 0; j < 32; ++j) acc = fma(X[r * LD + j], M[j * LD + c], acc);
            Xn[r * LD + c] = ca * X[r * LD + c] + cb * acc;
        }
        cur ^= 1;
        __syncthreads();
        if (mode == 1) break;
    }
    double* Xo = Xout + (long long)b * k2 * k1;
    for (int e = t; e < k2 * k1; e += 256) { const int r = e / k1, c = e - r * k1; Xo[e] = Xs[cur][r * LD + c]; }
}
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
    const size_t bState = (size_t)B * sizeof(polar_state), bCnt = (size_t)(nit > 0 ? nit : 1) * NS_MAX_STEPS * 8;
    const size_t need = dm_align_up(bAT) + dm_align_up(bBT) + 4 * dm_align_up(bC) + 4 * dm_align_up(bG) + dm_align_up(bImg) +
                        2 * dm_align_up(bT) + dm_align_up((size_t)B * N1pad * 8) + 4 * dm_align_up((size_t)B * N2 * 4) +
                        dm_align_up(bState) + dm_align_up(bCnt) + dm_align_up((size_t)B * 256 * 8) +
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
    polar_state* pst = (polar_state*)dm_ws_take(ctx, bState);
    int32_t* pcnt = (int32_t*)dm_ws_take(ctx, bCnt);
    double* pvec = (double*)dm_ws_take(ctx, (size_t)B * 256 * 8);
    const size_t lds_decide = (size_t)k1 * k1 * 8 <= 140 * 1024 ? (size_t)k1 * k1 * 8 : 0;          // T of the step in the LDS when it fits
    if (lds_decide) { rc = dm_grant_lds(ctx, (const void*)polar_decide_kernel, lds_decide); if (rc) return rc; }
    int32_t* host_words = nullptr;
    hipEvent_t host_ev = nullptr;
    rc = dm_pinned_words(ctx, &host_words, &host_ev);
    if (rc) return rc;
    if (!pst || !pcnt || !pvec) return dm_fail(ctx, DM_ENOMEM, "icp: workspace not reserved");
    if (!AT || !BT || !Ccur || !R || !Xa || !Xb || !G || !Ginv || !img || !Tm || !Wm || !n1 || !p21 || !iota || !ones || !amaxS ||
        (!chol && (!Gx || !Gt)))
        return dm_fail(ctx, DM_ENOMEM, "icp: workspace not reserved");

    DM_CHECK_HIP(ctx, hipMemsetAsync(info, 0, (size_t)B * 4, ctx->stream));
    DM_CHECK_HIP(ctx, hipMemsetAsync(pcnt, 0, bCnt, ctx->stream));
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
        // polar factor of Chat: every pair steps by its own schedule (polar_decide), the host stops launching when none is left
        double* xo = Xa;
        double* xn = Xb;
        const bool small_polar = k1 <= 32 && k2 <= 32;                                       // one launch for the whole iteration
        if (small_polar) {
            DM_LAUNCH(ctx, "polar_small", polar_small_kernel, dim3(B), dim3(256), 0, (const double*)Xa, Xb, k2, k1);
            xo = Xb; xn = Xa;
        } else {
        DM_CHECK_HIP(ctx, hipMemsetAsync(pst, 0, bState, ctx->stream));                      // mode 0, no step taken
        const dim3 grid_t(dm_cdiv(k1, TN_T) * dm_cdiv(k1, TN_T), 1, B), grid_w(dm_cdiv(k1, NT_T) * dm_cdiv(k1, NT_T), 1, B),
                   grid_x(dm_cdiv(k2, NT_T) * dm_cdiv(k1, NT_T), 1, B);
        auto launch_xtx = [&]() -> int {
            RowsF64 opx{xo, (long long)k2 * k1, k1, k1};
            OutPolarT ot{Tm, (long long)k1 * k1, k1, pst};
            DM_LAUNCH(ctx, "polar_xtx_tn_f64", (gemm_tn_f64<RowsF64, RowsF64, OutPolarT>), grid_t, dim3(256), 0, opx, opx, ot, k1, k1, k2,
                      pad_to(k2, TN_BK));
            return DM_OK;
        };
        rc = launch_xtx();
        if (rc) return rc;
        // The host learns how many pairs lift one step late (the lagged read): a pair decides to stop lifting on its own, so the T^2
        // launch follows "some pair lifted at the last step that has been read" and runs one step longer than needed, never shorter
        // (a pair only ever goes from lifting to not lifting).
        // A decision costs a launch (8 us for a small map, 19 us at k = 128: a third of a step there): a batch whose pairs keep lifting
        // pays more for its decisions than the lifts it saves (measured at k = 128, the bench's call from a poor start: 15.0 ms
        // against 14.1 on the fixed schedule; with L lifts and P polish steps (L + P) decisions + 3 L + 2 P products against 48
        // products break even near L = 6).  Such a batch continues on the r04 schedule -- lifts up to twelve in all, six
        // Newton-Schulz steps, no decisions: large maps as soon as any pair lifts at all, small ones when a pair still lifts at its
        // seventh step.  The others (a fit's result refined: the documented call) take the measured schedule.
        const int adaptive_lifts = k1 >= 96 ? 0 : 6;
        bool any_lift = true, fixed = false;
        for (int q = 0; q < NS_MAX_STEPS; ++q) {
            if (fixed) {
                if (q >= NS_LIFT + NS_POLISH) break;
                const bool lift = q < NS_LIFT;
                if (lift) {
                    KRowsF64 ta{Tm, (long long)k1 * k1, k1, k1, k1, 0};
                    OutAxpby ow{Tm, Wm, (long long)k1 * k1, k1, NS_B, NS_C};
                    DM_LAUNCH(ctx, "polar_poly_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutAxpby>), grid_w, dim3(256), 0, ta, ta, ow, k1, k1, k1);
                }
                KRowsF64 opa{xo, (long long)k2 * k1, k1, k2, k1, 0};
                KRowsF64 opb{lift ? Wm : Tm, (long long)k1 * k1, k1, k1, k1, 0};
                OutAxpby on{xo, xn, (long long)k2 * k1, k1, lift ? NS_A : 1.5, lift ? 1.0 : -0.5};
                DM_LAUNCH(ctx, "polar_update_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutAxpby>), grid_x, dim3(256), 0, opa, opb, on, k2, k1, k1);
                double* tmp = xo; xo = xn; xn = tmp;
                if (q + 1 < NS_LIFT + NS_POLISH) {
                    RowsF64 opx{xo, (long long)k2 * k1, k1, k1};
                    OutPlainTN ot{Tm, (long long)k1 * k1, k1};
                    DM_LAUNCH(ctx, "polar_xtx_tn_f64", (gemm_tn_f64<RowsF64, RowsF64, OutPlainTN>), grid_t, dim3(256), 0, opx, opx, ot, k1, k1, k2,
                              pad_to(k2, TN_BK));
                }
                continue;
            }
            int32_t* cnt_q = pcnt + ((size_t)it * NS_MAX_STEPS + q) * 2;                     // [pairs that step, pairs that lift]
            DM_LAUNCH(ctx, "polar_decide", polar_decide_kernel, dim3(B), dim3(256), lds_decide, Tm, k1, pst, pvec, cnt_q, lds_decide ? 1 : 0);
            DM_CHECK_HIP(ctx, hipMemcpyAsync(host_words, cnt_q, 8, hipMemcpyDeviceToHost, ctx->stream));
            DM_CHECK_HIP(ctx, hipEventRecord(host_ev, ctx->stream));
            if (any_lift) {                                   // W = wb T + wc T T^T   (T is symmetric) for the pairs that lift
                KRowsF64 ta{Tm, (long long)k1 * k1, k1, k1, k1, 0};
                OutPolarW ow{Tm, Wm, (long long)k1 * k1, k1, pst};
                DM_LAUNCH(ctx, "polar_poly_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutPolarW>), grid_w, dim3(256), 0, ta, ta, ow, k1, k1, k1);
            }
            {
                KRowsF64 opa{xo, (long long)k2 * k1, k1, k2, k1, 0};
                KRowsPolarM opb{KRowsF64{Tm, (long long)k1 * k1, k1, k1, k1, 0}, KRowsF64{Wm, (long long)k1 * k1, k1, k1, k1, 0}, pst};
                OutPolarX on{xo, xn, (long long)k2 * k1, k1, pst};
                DM_LAUNCH(ctx, "polar_update_nt_f64", (gemm_nt_f64<KRowsF64, KRowsPolarM, OutPolarX>), grid_x, dim3(256), 0, opa, opb, on, k2, k1, k1);
            }
            double* tmp = xo; xo = xn; xn = tmp;
            if (q + 1 < NS_MAX_STEPS) {
                rc = launch_xtx();                                                           // (the next step's T: the queue stays full while the host waits)
                if (rc) return rc;
            }
            DM_CHECK_HIP(ctx, hipEventSynchronize(host_ev));
            any_lift = host_words[1] > 0;
            if (host_words[0] == 0) break;                                                   // every pair was done before this step: X is where xo points
            if (q == adaptive_lifts && any_lift) fixed = true;
        }
        }   // (!small_polar)
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

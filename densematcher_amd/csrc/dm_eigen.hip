// Laplace-Beltrami eigenbasis on the device (dm_eigenbasis): SURVEY.md 8(f) #4.
//
// Reference computation replaced: TriMesh.process -> laplacian_spectrum (pyFM/mesh/trimesh.py:440-531) ->
// laplacian.laplacian_spectrum (pyFM/mesh/laplacian.py:143-182): scipy.sparse.linalg.eigsh(W, k, M=A, sigma=-0.01)
// (ARPACK shift-invert, a host sparse factorisation + Lanczos): the k smallest eigenpairs of W phi = lambda A phi,
// Phi^T A Phi = I.  ARPACK starts from a random vector and multiple eigenvalues come out in an arbitrary basis, so
// there is no bit parity to keep; the tests compare eigenvalues, invariant subspaces and the
// vertex maps built on the basis.
//
// GPU formulation: Chebyshev-filtered subspace iteration on the symmetric standard form L = A^-1/2 W A^-1/2 (the caller
// passes L in ELL format: the sparsity pattern of a mesh Laplacian is its vertex adjacency, host bookkeeping as in the
// reference).  m = k + guard vectors; every outer iteration is
//     Rayleigh-Ritz:  H = X^T L X (m x m)  ->  cyclic two-sided Jacobi eigensolver (one workgroup per mesh)  ->  X <- X Q
//     filter:         Y = T_d((L - c) / e) X   three-term recurrence, d sparse products (SpMM) with fused axpby,
//                     damping [theta_m, lambda_max(Gershgorin)], normalised at theta_0
//     orthonormalise: columns scaled to unit length (a filtered Ritz vector keeps its direction: the block stays well
//                     conditioned, cond < 1e3 with the degree ramp 4, 8, 16, 30 ...), then the orthogonal polar factor
//                     by the matrix-polynomial iteration of dm_icp.hip (GEMMs only, no Cholesky, no size limit)
// All dense work runs on the f64 matrix cores through the gemm tiles of dm_gemm_f64.h.  No host synchronisation: the
// iteration count is an argument and the residual max_j |L x_j - theta_j x_j| is returned per mesh.
#include "dm_gemm_f64.h"
#include "dm_internal.h"

static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

constexpr int EIG_NS_LIFT = 12, EIG_NS_POLISH = 8;
constexpr double EIG_NS_A = 3.4445, EIG_NS_B = -4.7750, EIG_NS_C = 2.0315;
constexpr int EIG_MAX_DEG = 64;

// ---- functors ---------------------------------------------------------------------------------------------------------
struct EigRowsTN {                  // K-major float64 operand for gemm_tn_f64: rows n of a (B, nrows, ld) matrix
    const double* p; long long stride_b; int ld; int ncols;
    __device__ __forceinline__ void load4(int b, int n, int col0, double (&v)[4]) const {
        const double* row = p + b * stride_b + (long long)n * ld;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (col0 + e < ncols) ? row[col0 + e] : 0.0;
    }
};
struct EigOutTNPartial {
    double* p; long long Z; int M; int N;
    __device__ __forceinline__ void store(int z, int split, int m, int c, double v) const {
        p[(((long long)split * Z + z) * M + m) * N + c] = v;
    }
};
struct EigOutNT {
    double* p; long long stride_b; int ld;
    __device__ __forceinline__ void store(int b, int i, int j, double v) const { p[b * stride_b + (long long)i * ld + j] = v; }
};
struct EigOutAxpby {                // Xnew = alpha Xold + beta (product)
    const double* xo; double* xn; long long stride_b; int ld; double alpha, beta;
    __device__ __forceinline__ void store(int b, int i, int j, double v) const {
        const long long o = b * stride_b + (long long)i * ld + j;
        xn[o] = alpha * xo[o] + beta * v;
    }
};

__global__ __launch_bounds__(256) void eig_reduce_partials_kernel(const double* __restrict__ partial, int nsplit, long long n,
                                                                  double* __restrict__ out, int symmetrise_m) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // (the element and its mirror image in flight together, eight partials at a time, added in split order)
    long long j = i;
    if (symmetrise_m > 0) {         // (H + H^T) / 2: the Jacobi sweeps assume exact symmetry
        const long long mm = (long long)symmetrise_m * symmetrise_m;
        const long long b = i / mm, e = i - b * mm;
        const int r = (int)(e / symmetrise_m), c = (int)(e - (long long)r * symmetrise_m);
        j = b * mm + (long long)c * symmetrise_m + r;
    }
    double s = 0.0, s2 = 0.0;
    int q = 0;
    for (; q + 8 <= nsplit; q += 8) {
        double a[8], c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = partial[(long long)(q + u) * n + i]; c[u] = partial[(long long)(q + u) * n + j]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s += a[u]; s2 += c[u]; }
    }
    for (; q < nsplit; ++q) { s += partial[(long long)q * n + i]; s2 += partial[(long long)q * n + j]; }
    out[i] = symmetrise_m > 0 ? 0.5 * (s + s2) : s;
}

// ---- sparse product with the fused three-term recurrence ------------------------------------------------------------------
// Ynew[i][c] = alpha (sum_nz L[i][n] Y[n][c] - cc Y[i][c]) - beta Yprev[i][c];  (alpha, cc, beta) = coef[b][step] or (1, 0, 0)
// A workgroup covers 256 / cw rows x cw columns (cw = the block width rounded up to a power of two, <= 256): with one row per
// workgroup a block of 52 vectors kept 52 of 256 lanes busy and 128 meshes took 324 us per product (r05: 72 ms of a 64-pair
// compute_surface_map_batch call); same sums in the same order.
__global__ __launch_bounds__(256) void spmm_ell_kernel(const double* __restrict__ vals, const int32_t* __restrict__ cols, int N, int nnz,
                                                       const double* __restrict__ Y, const double* __restrict__ Yprev,
                                                       double* __restrict__ Ynew, int m, const double* __restrict__ coef, int step, int cw,
                                                       int xcd_B) {
    const int rpb = 256 / cw;
    int b, bx, by;
    if (xcd_B) {
        // XCD-aware order (xcd_B = the batch size; a one-dimensional grid): workgroups go to the eight XCDs round robin, each with its
        // own L2, so with the plain (row block, mesh) order every XCD gathers from every mesh's block of vectors -- eight L2 fills of each
        // (measured: 1.1 GB per product for 0.23 GB of operands).  Here workgroup w belongs to XCD w & 7 and walks the meshes
        // xcd, xcd + 8, ... one after the other: a mesh's vectors are fetched into ONE L2.
        const int nbx = (N + rpb - 1) / rpb, nby = (m + cw - 1) / cw, per_mesh = nbx * nby;
        const int w = blockIdx.x, xcd = w & 7, q = w >> 3;
        b = (q / per_mesh) * 8 + xcd;
        if (b >= xcd_B) return;
        const int r = q % per_mesh;
        by = r / nbx; bx = r - by * nbx;
    } else {
        b = blockIdx.z; bx = blockIdx.x; by = blockIdx.y;
    }
    const int i = bx * rpb + threadIdx.x / cw;              // (the row index rides on grid x: meshes above 65535 vertices)
    const int c = by * cw + (threadIdx.x & (cw - 1));
    if (c >= m || i >= N) return;
    const double* Yb = Y + (long long)b * N * m;
    const double* vr = vals + ((long long)b * N + i) * nnz;
    const int32_t* cr = cols + ((long long)b * N + i) * nnz;
    double acc = 0.0;
    for (int q = 0; q < nnz; ++q) acc += vr[q] * Yb[(long long)cr[q] * m + c];
    double alpha = 1.0, cc = 0.0, beta = 0.0;
    if (coef) { const double* k = coef + ((long long)b * EIG_MAX_DEG + step) * 3; alpha = k[0]; cc = k[1]; beta = k[2]; }
    double out = alpha * (acc - cc * Yb[(long long)i * m + c]);
    if (Yprev && beta != 0.0) out -= beta * Yprev[((long long)b * N + i) * m + c];
    Ynew[((long long)b * N + i) * m + c] = out;
}

// The same product with FOUR columns per thread (m a multiple of 4: the block sizes the solver chooses are multiples of 32): a row's
// entries (value, column) are requested once per four outputs instead of once per output and a gather is two 16-byte loads -- a third
// of the memory instructions per output; same sums in the same order.  Workgroup = 256 / (m / 4) rows (m / 4 <= 64 threads per row).
__global__ __launch_bounds__(256) void spmm_ell4_kernel(const double* __restrict__ vals, const int32_t* __restrict__ cols, int N, int nnz,
                                                        const double* __restrict__ Y, const double* __restrict__ Yprev,
                                                        double* __restrict__ Ynew, int m, const double* __restrict__ coef, int step, int tpr,
                                                        int xcd_B) {
    const int rpb = 256 / tpr;                              // tpr: threads per row (a power of two >= m / 4)
    int b, bx;
    if (xcd_B) {
        const int nbx = (N + rpb - 1) / rpb;
        const int w = blockIdx.x, xcd = w & 7, q = w >> 3;
        b = (q / nbx) * 8 + xcd;
        if (b >= xcd_B) return;
        bx = q % nbx;
    } else {
        b = blockIdx.z; bx = blockIdx.x;
    }
    const int i = bx * rpb + threadIdx.x / tpr;
    const int c = (threadIdx.x & (tpr - 1)) * 4;
    if (c >= m || i >= N) return;
    const double* Yb = Y + (long long)b * N * m;
    const double* vr = vals + ((long long)b * N + i) * nnz;
    const int32_t* cr = cols + ((long long)b * N + i) * nnz;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int q = 0; q < nnz; ++q) {
        const double v = vr[q];
        const f64x2* y = reinterpret_cast<const f64x2*>(Yb + (long long)cr[q] * m + c);
        const f64x2 y0 = y[0], y1 = y[1];
        a0 += v * y0[0]; a1 += v * y0[1]; a2 += v * y1[0]; a3 += v * y1[1];
    }
    double alpha = 1.0, cc = 0.0, beta = 0.0;
    if (coef) { const double* k = coef + ((long long)b * EIG_MAX_DEG + step) * 3; alpha = k[0]; cc = k[1]; beta = k[2]; }
    const long long o = ((long long)b * N + i) * m + c;
    const f64x2* ys = reinterpret_cast<const f64x2*>(Yb + (long long)i * m + c);
    const f64x2 s0 = ys[0], s1 = ys[1];
    double o0 = alpha * (a0 - cc * s0[0]), o1 = alpha * (a1 - cc * s0[1]), o2 = alpha * (a2 - cc * s1[0]), o3 = alpha * (a3 - cc * s1[1]);
    if (Yprev && beta != 0.0) {
        const f64x2* yp = reinterpret_cast<const f64x2*>(Yprev + o);
        const f64x2 p0 = yp[0], p1 = yp[1];
        o0 -= beta * p0[0]; o1 -= beta * p0[1]; o2 -= beta * p1[0]; o3 -= beta * p1[1];
    }
    f64x2* yo = reinterpret_cast<f64x2*>(Ynew + o);
    yo[0] = f64x2{o0, o1}; yo[1] = f64x2{o2, o3};
}

// lmax[b] = max_i sum_q |L[i][q]|   (Gershgorin bound of the largest eigenvalue)
__global__ __launch_bounds__(256) void gershgorin_kernel(const double* __restrict__ vals, int N, int nnz, double* __restrict__ lmax) {
    __shared__ double sh[256];
    const int b = blockIdx.x, t = threadIdx.x;
    double mx = 0.0;
    for (int i = t; i < N; i += 256) {
        const double* vr = vals + ((long long)b * N + i) * nnz;
        double s = 0.0;
        for (int q = 0; q < nnz; ++q) s += fabs(vr[q]);
        mx = fmax(mx, s);
    }
    sh[t] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (t < off) sh[t] = fmax(sh[t], sh[t + off]); __syncthreads(); }
    if (t == 0) lmax[b] = sh[0];
}

// coefficients of the scaled Chebyshev recurrence of degree `deg` damping [theta[m-1], lmax], normalised at theta[0]
// (Y_1 = (sigma_1 / e)(L - c) X;  Y_{j+1} = (2 sigma_{j+1} / e)(L - c) Y_j - sigma_j sigma_{j+1} Y_{j-1})
__global__ void cheb_coef_kernel(const double* __restrict__ theta, int m, const double* __restrict__ lmax, int deg, double* __restrict__ coef) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const double a0 = theta[(long long)b * m], a = theta[(long long)b * m + m - 1];
    const double bb = fmax(lmax[b], a * (1.0 + 1e-6) + 1e-300);
    const double e = 0.5 * (bb - a), c = 0.5 * (bb + a);
    const double sigma1 = e / (a0 - c);
    double sig_prev = sigma1;
    double* k = coef + (long long)b * EIG_MAX_DEG * 3;
    k[0] = sigma1 / e; k[1] = c; k[2] = 0.0;
    for (int j = 2; j <= deg; ++j) {
        const double sig = 1.0 / (2.0 / sigma1 - sig_prev);
        k[(j - 1) * 3] = 2.0 * sig / e; k[(j - 1) * 3 + 1] = c; k[(j - 1) * 3 + 2] = sig_prev * sig;
        sig_prev = sig;
    }
}

// ---- dense helpers -----------------------------------------------------------------------------------------------------------
// X[b][:, c] <- scale * X[b][:, c] / |X[b][:, c]|     (one workgroup per column)
__global__ __launch_bounds__(256) void unit_columns_kernel(double* __restrict__ X, int N, int m, double scale) {
    __shared__ double sh[256];
    const int c = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    double* col = X + (long long)b * N * m + c;
    double s = 0.0;
    for (int i = t; i < N; i += 256) { const double v = col[(long long)i * m]; s += v * v; }
    sh[t] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (t < off) sh[t] += sh[t + off]; __syncthreads(); }
    const double nrm = sqrt(sh[0]);
    const double f = nrm > 0.0 ? scale / nrm : 0.0;
    for (int i = t; i < N; i += 256) col[(long long)i * m] *= f;
}

// The same for blocks of M = 8, 16, 32 or 64 columns with ROWS read as they lie (one workgroup per mesh; a column walk reads 8 bytes of
// every 64-byte sector: 330 us for 128 meshes of 2048 x 32 against 40).  Bit-identical to unit_columns_kernel: there thread t' of a
// column's workgroup adds the squares of rows t', t' + 256, ... in ascending order and the 256 partial sums meet in a binary tree; here
// thread (r, c) = (t / M, t % M) keeps the 256 / RPP partial sums of its row classes apart (RPP = 256 / M rows per pass) and the same
// tree runs over the same 256 partials of every column in the LDS.
template <int M>
__global__ __launch_bounds__(256) void unit_columns_rows_kernel(double* __restrict__ X, int N, double scale) {
    constexpr int RPP = 256 / M, NCL = 256 / RPP;          // rows per pass; classes (i mod 256) per thread
    extern __shared__ double uc_sh[];                      // M x 256
    const int b = blockIdx.x, t = threadIdx.x, c = t % M, r = t / M;
    double* Xb = X + (long long)b * N * M;
    double acc[NCL];
#pragma unroll
    for (int u = 0; u < NCL; ++u) acc[u] = 0.0;
    for (int base = 0; base < N; base += 256) {
#pragma unroll
        for (int u = 0; u < NCL; ++u) {
            const int i = base + u * RPP + r;              // class i mod 256 = u RPP + r
            if (i < N) { const double v = Xb[(long long)i * M + c]; acc[u] += v * v; }
        }
    }
#pragma unroll
    for (int u = 0; u < NCL; ++u) uc_sh[c * 256 + u * RPP + r] = acc[u];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        for (int e = t; e < M * off; e += 256) { const int cc = e / off, q = e - cc * off; uc_sh[cc * 256 + q] += uc_sh[cc * 256 + q + off]; }
        __syncthreads();
    }
    const double nrm = sqrt(uc_sh[c * 256]);
    const double f = nrm > 0.0 ? scale / nrm : 0.0;
    for (int i = r; i < N; i += RPP) Xb[(long long)i * M + c] *= f;
}
static int launch_unit_columns(dm_ctx* ctx, double* X, int B, int N, int m, double scale);

// resid[b] = max_{c < k} |LX[:, c] - theta_c X[:, c]|   (one workgroup per column; max through ordered-bits atomicMax)
__global__ __launch_bounds__(256) void ritz_residual_kernel(const double* __restrict__ X, const double* __restrict__ LX, int N, int m,
                                                            const double* __restrict__ theta, unsigned long long* __restrict__ resid_bits) {
    __shared__ double sh[256];
    const int c = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const double th = theta[(long long)b * m + c];
    double s = 0.0;
    for (int i = t; i < N; i += 256) {
        const long long o = ((long long)b * N + i) * m + c;
        const double r = LX[o] - th * X[o];
        s += r * r;
    }
    sh[t] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (t < off) sh[t] += sh[t + off]; __syncthreads(); }
    if (t == 0) atomicMax(resid_bits + b, (unsigned long long)__double_as_longlong(sqrt(sh[0])));   // (non-negative doubles order like their bits)
}

// Cyclic two-sided Jacobi eigensolver of a symmetric m x m matrix (m <= 2048), one workgroup of 1024 threads per matrix,
// round-robin ordering (m/2 disjoint rotations per round: the row updates of a round are independent, then the column
// updates of H and of the accumulated eigenvectors V).  H is overwritten (diagonal = eigenvalues), V (m x m, columns).
// IN_LDS: both matrices live in the LDS for the duration (2 m^2 doubles; m <= 90): a round is three barriers around two
// read-modify-write passes, and from global memory each pass waits for an L2 round trip: 1.58 -> 1.19 ms per call at m = 52
// (5 to 10 sweeps of 51 rounds; four waves instead of sixteen: 1.67 ms, the passes want the threads); the two sides of a round
// applied to disjoint 2 x 2 blocks in one pass: 1.06 ms; the sweep's convergence measure kept in registers instead of two
// 64-bit LDS atomicMax per pair and round: 0.37 ms.  Same rotations, same arithmetic.
template <bool IN_LDS>
__global__ __launch_bounds__(1024) void jacobi_eigh_kernel(double* __restrict__ Hs, double* __restrict__ Vs, int m, int max_sweeps) {
    extern __shared__ __attribute__((aligned(16))) double jac_sm[];
    __shared__ double rc[1024], rs[1024];          // (a round's m / 2 rotations: m <= 2048)
    __shared__ int rp[1024], rq[1024];
    __shared__ unsigned long long s_off, s_diag;
    const int b = blockIdx.x, t = threadIdx.x, nthr = blockDim.x;
    double* Hg = Hs + (long long)b * m * m;
    double* Vg = Vs + (long long)b * m * m;
    double* H = IN_LDS ? jac_sm : Hg;
    double* V = IN_LDS ? jac_sm + m * m : Vg;
    const int mm = m + (m & 1), half = mm / 2;
    if (IN_LDS) for (int e = t; e < m * m; e += nthr) H[e] = Hg[e];
    for (int e = t; e < m * m; e += nthr) V[e] = (e / m == e % m) ? 1.0 : 0.0;
    // this thread's work items of a round (the same every round): up to two 2 x 2 blocks of H (pair a, pair b) and up to four
    // (row of V, pair) items  (IN_LDS: half <= 45, 1024 threads)
    int blk_a[2], blk_b[2], vit_i[4], vit_p[4];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int e = t + u * nthr; blk_a[u] = e < half * half ? e / half : -1; blk_b[u] = e < half * half ? e % half : 0; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = t + u * nthr; vit_i[u] = e < half * m ? e / half : -1; vit_p[u] = e < half * m ? e % half : 0; }
    __syncthreads();
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (t == 0) { s_off = 0ull; s_diag = 0ull; }
        __syncthreads();
        double my_off = 0.0, my_diag = 0.0;               // (this pair slot's largest pivot / diagonal entry of the sweep: one atomic per sweep)
        int pc = t % (mm - 1), qc = (2 * (mm - 1) - t) % (mm - 1);       // (r + t) mod (mm - 1), (r - t) mod (mm - 1) at r = 0, advanced by one per round
        for (int r = 0; r < mm - 1; ++r) {
            if (t < half) {
                int p, q;
                if (t == 0) { p = mm - 1; q = r; }
                else { p = pc; q = qc; }
                if (p > q) { const int x = p; p = q; q = x; }
                double c = 1.0, s = 0.0;
                if (q < m) {
                    const double a = H[(long long)p * m + p], d = H[(long long)q * m + q], bq = H[(long long)p * m + q];
                    my_off = fmax(my_off, fabs(bq));
                    my_diag = fmax(my_diag, fmax(fabs(a), fabs(d)));
                    if (fabs(bq) > 1e-300) {
                        const double tau = (d - a) / (2.0 * bq);
                        const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + tt * tt);
                        s = tt * c;
                    }
                }
                rp[t] = p; rq[t] = q; rc[t] = c; rs[t] = s;
            }
            pc = pc + 1 == mm - 1 ? 0 : pc + 1; qc = qc + 1 == mm - 1 ? 0 : qc + 1;
            __syncthreads();
            if constexpr (IN_LDS) {
                // one pass: the 2 x 2 block (pair a, pair b) of H takes its row rotation, then its column rotation, in place
                // (the blocks of a round are disjoint: no barrier between the two sides); V's columns in the same pass
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (blk_a[u] >= 0) {
                        const int a = blk_a[u], bb = blk_b[u];
                        const double sa = rs[a], sb = rs[bb];
                        if (sa != 0.0 || sb != 0.0) {
                            const double ca = rc[a], cb = rc[bb];
                            const int pa = rp[a], qa = rq[a], pb = rp[bb], qb = rq[bb];
                            const bool vqa = qa < m, vqb = qb < m;
                            double h00 = H[pa * m + pb], h01 = vqb ? H[pa * m + qb] : 0.0;
                            double h10 = vqa ? H[qa * m + pb] : 0.0, h11 = (vqa && vqb) ? H[qa * m + qb] : 0.0;
                            if (sa != 0.0) {
                                const double t00 = ca * h00 - sa * h10, t10 = sa * h00 + ca * h10;
                                const double t01 = ca * h01 - sa * h11, t11 = sa * h01 + ca * h11;
                                h00 = t00; h10 = t10; h01 = t01; h11 = t11;
                            }
                            if (sb != 0.0) {
                                const double n00 = cb * h00 - sb * h01, n01 = sb * h00 + cb * h01;
                                const double n10 = cb * h10 - sb * h11, n11 = sb * h10 + cb * h11;
                                h00 = n00; h01 = n01; h10 = n10; h11 = n11;
                            }
                            H[pa * m + pb] = h00;
                            if (vqb) H[pa * m + qb] = h01;
                            if (vqa) H[qa * m + pb] = h10;
                            if (vqa && vqb) H[qa * m + qb] = h11;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (vit_i[u] >= 0) {
                        const int pr = vit_p[u];
                        const double sv = rs[pr];
                        if (sv != 0.0) {
                            const double cv = rc[pr];
                            const int op = vit_i[u] * m + rp[pr], oq = vit_i[u] * m + rq[pr];
                            const double vp = V[op], vq = V[oq];
                            V[op] = cv * vp - sv * vq; V[oq] = sv * vp + cv * vq;
                        }
                    }
                }
                __syncthreads();
            } else {
            for (int idx = t; idx < half * m; idx += nthr) {              // rows p, q <- J^T rows
                const int pr = idx / m, j = idx - pr * m;
                const double s = rs[pr];
                if (s != 0.0) {
                    const double c = rc[pr];
                    const long long op = (long long)rp[pr] * m + j, oq = (long long)rq[pr] * m + j;
                    const double hp = H[op], hq = H[oq];
                    H[op] = c * hp - s * hq; H[oq] = s * hp + c * hq;
                }
            }
            __syncthreads();
            for (int idx = t; idx < half * m; idx += nthr) {              // columns p, q <- columns J   (H and V)
                const int i = idx / half, pr = idx - i * half;
                const double s = rs[pr];
                if (s != 0.0) {
                    const double c = rc[pr];
                    const long long op = (long long)i * m + rp[pr], oq = (long long)i * m + rq[pr];
                    const double hp = H[op], hq = H[oq];
                    H[op] = c * hp - s * hq; H[oq] = s * hp + c * hq;
                    const double vp = V[op], vq = V[oq];
                    V[op] = c * vp - s * vq; V[oq] = s * vp + c * vq;
                }
            }
            __syncthreads();
            }
        }
        if (t < half) {
            atomicMax(&s_off, (unsigned long long)__double_as_longlong(my_off));
            atomicMax(&s_diag, (unsigned long long)__double_as_longlong(my_diag));
        }
        __syncthreads();
        const double off = __longlong_as_double((long long)s_off), dg = __longlong_as_double((long long)s_diag);
        __syncthreads();
        if (off <= 1e-15 * dg) break;
    }
    if (IN_LDS) for (int e = t; e < m * m; e += nthr) { Hg[e] = H[e]; Vg[e] = V[e]; }
}

// theta[b] = sorted diagonal of H; Q[b][:, rank] = V[b][:, j]   (ascending; equal values keep their index order)
__global__ __launch_bounds__(256) void ritz_sort_kernel(const double* __restrict__ Hs, const double* __restrict__ Vs, int m,
                                                        double* __restrict__ theta, double* __restrict__ Q) {
    extern __shared__ double sd[];               // m diagonal values, then m ranks (as ints)
    int* rank = reinterpret_cast<int*>(sd + m);
    const int b = blockIdx.x, t = threadIdx.x;
    const double* H = Hs + (long long)b * m * m;
    const double* V = Vs + (long long)b * m * m;
    for (int j = t; j < m; j += 256) sd[j] = H[(long long)j * m + j];
    __syncthreads();
    for (int j = t; j < m; j += 256) {
        const double v = sd[j];
        int r = 0;
        for (int i = 0; i < m; ++i) r += (sd[i] < v || (sd[i] == v && i < j)) ? 1 : 0;
        rank[j] = r;
        theta[(long long)b * m + r] = v;
    }
    __syncthreads();
    for (int e = t; e < m * m; e += 256) {
        const int i = e / m, j = e - i * m;
        Q[(long long)b * m * m + (long long)i * m + rank[j]] = V[e];
    }
}

// Phi[b][i][c] = X[b][i][c] / sqrt(mass[b][i]) for c < k, sign fixed so that the entry of largest magnitude is positive;
// lam[b][c] = theta[b][c]
__global__ __launch_bounds__(256) void eig_finish_kernel(const double* __restrict__ X, int N, int m, int k, const float* __restrict__ mass,
                                                         const double* __restrict__ theta, double* __restrict__ Phi, double* __restrict__ lam) {
    __shared__ double sh[256];
    __shared__ int shi[256];
    const int c = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    double best = -1.0; int bi = 0;
    for (int i = t; i < N; i += 256) {
        const double v = fabs(X[((long long)b * N + i) * m + c]);
        if (v > best) { best = v; bi = i; }
    }
    sh[t] = best; shi[t] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off && (sh[t + off] > sh[t] || (sh[t + off] == sh[t] && shi[t + off] < shi[t]))) { sh[t] = sh[t + off]; shi[t] = shi[t + off]; }
        __syncthreads();
    }
    const double sgn = X[((long long)b * N + shi[0]) * m + c] < 0.0 ? -1.0 : 1.0;
    for (int i = t; i < N; i += 256)
        Phi[((long long)b * N + i) * k + c] = sgn * X[((long long)b * N + i) * m + c] / sqrt((double)mass[(long long)b * N + i]);
    if (t == 0) lam[(long long)b * k + c] = theta[(long long)b * m + c];
}

// Cholesky-QR building block: G = X^T X (m x m, symmetric positive definite) -> M = L^-T with G = L L^T, so that X M has
// orthonormal columns.  One workgroup per mesh, G in the LDS (m <= 128: 129 KiB); right-looking factorisation, then column c of
// L^-1 by forward substitution in thread c.  `shift` (relative to the mean diagonal entry) is added to the diagonal first: the
// first of the two passes runs with 1e-13 so that a block at the edge of the factorisation's reach (condition 1e7) still factors;
// the second pass sees a block orthonormal to ~1e-10 and runs unshifted.  A non-positive pivot sets fail[b] (the residual the
// call returns for that mesh is then +inf: the caller's convergence test can never pass on a broken basis).
constexpr int EIG_CHOLQR_MAX = 128;
__global__ __launch_bounds__(256) void eig_chol_inv_kernel(const double* __restrict__ Gs, double* __restrict__ Ms, int m, double shift, int* __restrict__ fail) {
    extern __shared__ __attribute__((aligned(16))) double chol_sm[];
    __shared__ double s_piv;
    __shared__ double red[4];
    const int b = blockIdx.x, t = threadIdx.x, ld = m + 1;
    double* A = chol_sm;                                   // [m][m + 1]
    const double* G = Gs + (long long)b * m * m;
    double tr = 0.0;
    for (int e = t; e < m * m; e += 256) { const int i = e / m, j = e - i * m; const double x = G[e]; A[i * ld + j] = x; if (i == j) tr += x; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tr += __shfl_xor(tr, off);
    if ((t & 63) == 0) red[t >> 6] = tr;
    __syncthreads();
    const double add = shift * ((red[0] + red[1]) + (red[2] + red[3])) / (double)m;
    for (int j = t; j < m; j += 256) A[j * ld + j] += add;
    __syncthreads();
    bool bad = false;
    for (int j = 0; j < m; ++j) {
        if (t == 0) {
            const double d = A[j * ld + j];
            s_piv = (d > 0.0 && isfinite(d)) ? sqrt(d) : -1.0;
        }
        __syncthreads();
        double piv = s_piv;
        if (!(piv > 0.0)) { bad = true; piv = 1.0; }
        const double inv = 1.0 / piv;
        for (int i = j + t; i < m; i += 256) A[i * ld + j] = (i == j) ? piv : A[i * ld + j] * inv;      // column j of L
        __syncthreads();
        // trailing update: A[i][c] -= L[i][j] L[c][j] for j < c <= i
        const int nrem = m - j - 1;
        for (int e = t; e < nrem * nrem; e += 256) {
            const int i = j + 1 + e / nrem, c = j + 1 + e % nrem;
            if (c <= i) A[i * ld + c] -= A[i * ld + j] * A[c * ld + j];
        }
        __syncthreads();
    }
    if (bad && t == 0) atomicOr(fail + b, 1);
    // M = L^-T: thread c solves L y = e_c (y_i = 0 for i < c); y_i, i > c, waits in the unused upper triangle A[c][i], then row c of M
    double* M = Ms + (long long)b * m * m;
    for (int c = t; c < m; c += 256) {
        const double yc = 1.0 / A[c * ld + c];
        for (int i = c + 1; i < m; ++i) {
            double sacc = -A[i * ld + c] * yc;
            for (int k = c + 1; k < i; ++k) sacc = fma(-A[i * ld + k], A[c * ld + k], sacc);
            A[c * ld + i] = sacc / A[i * ld + i];
        }
        for (int i = 0; i < m; ++i) M[(long long)c * m + i] = i < c ? 0.0 : (i == c ? yc : A[c * ld + i]);
    }
}
// resid[b] = +inf for the meshes whose orthonormalisation failed
__global__ void eig_fail_kernel(const int* __restrict__ fail, int B, double* __restrict__ resid) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && fail[b]) resid[b] = DM_INF_F64;
}

// ---- building blocks on the host side ------------------------------------------------------------------------------------------
struct eig_ws {
    int B, N, m;
    int* fail;                      // (B) set when a Cholesky-QR pass met a non-positive pivot
    double *T, *W, *part;           // m x m scratch (x2) and split-K partials
    int nsplit, kchunk;             // split-K of the Gram products: rows per chunk by the block size only (never by the batch)
};
// G (B, m, m) = P^T R over the N rows (split-K, fixed order); symmetrised when asked
static int eig_gram(dm_ctx* ctx, const eig_ws& w, const double* P, const double* R, double* G, int symmetrise) {
    EigRowsTN px{P, (long long)w.N * w.m, w.m, w.m};
    EigRowsTN ry{R, (long long)w.N * w.m, w.m, w.m};
    EigOutTNPartial out{w.part, w.B, w.m, w.m};
    DM_LAUNCH(ctx, "eig_gram_tn_f64", (gemm_tn_f64<EigRowsTN, EigRowsTN, EigOutTNPartial>),
              dim3(dm_cdiv(w.m, TN_T) * dm_cdiv(w.m, TN_T), w.nsplit, w.B), dim3(256), 0, px, ry, out, w.m, w.m, w.N, w.kchunk);
    const long long n = (long long)w.B * w.m * w.m;
    DM_LAUNCH(ctx, "eig_reduce", eig_reduce_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, w.part, w.nsplit, n, G,
              symmetrise ? w.m : 0);
    return DM_OK;
}
// Xout = alpha Xin + beta Xin M   (N x m times m x m; M read as is, it is symmetric or already the wanted factor)
static int eig_apply(dm_ctx* ctx, const eig_ws& w, const double* Xin, const double* M, int transM, double alpha, double beta, double* Xout) {
    KRowsF64 xa{Xin, (long long)w.N * w.m, w.m, w.N, w.m, 0};
    KRowsF64 mb{M, (long long)w.m * w.m, w.m, w.m, w.m, transM};
    EigOutAxpby out{Xin, Xout, (long long)w.N * w.m, w.m, alpha, beta};
    DM_LAUNCH(ctx, "eig_apply_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, EigOutAxpby>), dim3(dm_cdiv(w.N, NT_T) * dm_cdiv(w.m, NT_T), 1, w.B),
              dim3(256), 0, xa, mb, out, w.N, w.m, w.m);
    return DM_OK;
}
// orthogonal polar factor of X (singular values in (0, 1]) by odd matrix polynomials; returns the buffer holding it
static int eig_polar(dm_ctx* ctx, const eig_ws& w, double* Xa, double* Xb, double** result) {
    double* xo = Xa;
    double* xn = Xb;
    for (int q = 0; q < EIG_NS_LIFT + EIG_NS_POLISH; ++q) {
        const bool lift = q < EIG_NS_LIFT;
        int rc = eig_gram(ctx, w, xo, xo, w.T, 1);                                  // T = X^T X
        if (rc) return rc;
        if (lift) {                                                                  // W = b T + c T T
            KRowsF64 ta{w.T, (long long)w.m * w.m, w.m, w.m, w.m, 0};
            EigOutAxpby ow{w.T, w.W, (long long)w.m * w.m, w.m, EIG_NS_B, EIG_NS_C};
            DM_LAUNCH(ctx, "eig_poly_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, EigOutAxpby>), dim3(dm_cdiv(w.m, NT_T) * dm_cdiv(w.m, NT_T), 1, w.B),
                      dim3(256), 0, ta, ta, ow, w.m, w.m, w.m);
        }
        rc = eig_apply(ctx, w, xo, lift ? w.W : w.T, 0, lift ? EIG_NS_A : 1.5, lift ? 1.0 : -0.5, xn);
        if (rc) return rc;
        double* tmp = xo; xo = xn; xn = tmp;
    }
    *result = xo;
    return DM_OK;
}

// orthonormal basis of span(X) (columns of unit length on entry): two Cholesky-QR passes for blocks up to 128 vectors -- two Gram
// products, two small factorisations and two block products, 8 launches -- else the matrix-polynomial polar iteration (60 launches:
// round 4 used it for every size; at 128 meshes its products were 39 of the eigensolver's 107 ms, and a third of its launches)
static int eig_orthonormalize(dm_ctx* ctx, const eig_ws& w, double* Xa, double* Xb, double** result) {
    if (w.m > EIG_CHOLQR_MAX) return eig_polar(ctx, w, Xa, Xb, result);
    const size_t lds = (size_t)w.m * (w.m + 1) * 8;
    int rc = dm_grant_lds(ctx, (const void*)eig_chol_inv_kernel, lds);
    if (rc) return rc;
    double* xo = Xa;
    double* xn = Xb;
    for (int pass = 0; pass < 2; ++pass) {
        rc = eig_gram(ctx, w, xo, xo, w.T, 1);
        if (rc) return rc;
        DM_LAUNCH(ctx, "eig_chol_inv", eig_chol_inv_kernel, dim3(w.B), dim3(256), lds, (const double*)w.T, w.W, w.m, pass == 0 ? 1e-13 : 0.0, w.fail);
        rc = eig_apply(ctx, w, xo, w.W, 1, 0.0, 1.0, xn);
        if (rc) return rc;
        double* tmp = xo; xo = xn; xn = tmp;
    }
    *result = xo;
    return DM_OK;
}

static int launch_unit_columns(dm_ctx* ctx, double* X, int B, int N, int m, double scale) {
#define DM_UC_ROWS(M_)                                                                                                              \
    {                                                                                                                               \
        const size_t lds = (size_t)(M_) * 256 * 8;                                                                                  \
        int rc = dm_grant_lds(ctx, (const void*)unit_columns_rows_kernel<M_>, lds);                                                 \
        if (rc) return rc;                                                                                                          \
        DM_LAUNCH(ctx, "eig_unit_columns", unit_columns_rows_kernel<M_>, dim3(B), dim3(256), lds, X, N, scale);                     \
        return DM_OK;                                                                                                               \
    }
    if (m == 8) DM_UC_ROWS(8)
    if (m == 16) DM_UC_ROWS(16)
    if (m == 32) DM_UC_ROWS(32)
    if (m == 64) DM_UC_ROWS(64)
#undef DM_UC_ROWS
    DM_LAUNCH(ctx, "eig_unit_columns", unit_columns_kernel, dim3(m, B), dim3(256), 0, X, N, m, scale);
    return DM_OK;
}

extern "C" int dm_eigenbasis(dm_ctx* ctx, int B, int N, int nnz, const int32_t* ell_cols, const double* ell_vals, const float* mass,
                             int k, int guard, int n_iter, int degree, int warm_start, double* X /* B*N*(k+guard), in/out */,
                             double* lam /* B*k */, double* Phi /* B*N*k */, double* resid /* B */) {
    if (!ctx) return DM_EINVAL;
    // warm_start == 2: the DENSE route for meshes too small for the filtered iteration -- X holds an orthonormal basis of the whole
    // space (the identity: k + guard = N <= 2048), no filter, ONE Rayleigh-Ritz step = the eigendecomposition of L itself
    const bool dense = warm_start == 2;
    if (dense) n_iter = 0;
    DM_REQUIRE(ctx, B > 0 && N > 0 && nnz > 0 && k > 0 && guard >= 0 && (n_iter > 0 || dense), "sizes must be positive");
    DM_REQUIRE(ctx, !dense || k + guard == N, "the dense route takes the whole space: k + guard = N");
    DM_REQUIRE(ctx, ell_cols && ell_vals && mass && X && lam && Phi && resid, "null pointer");
    const int m = k + guard;
    // (the dense route's Jacobi eigensolve runs from global memory past 90 vectors, one workgroup per mesh: ~0.3 s at N = 1024, seconds
    //  at N = 2048 -- slow, but a mesh whose wanted range passes the middle of its spectrum is solved like the reference's ARPACK solves it)
    DM_REQUIRE(ctx, m <= N && m <= (dense ? 2048 : 512), "k + guard must be <= min(N, 512) (dense route: N <= 2048)");
    DM_REQUIRE(ctx, degree >= 2 && degree <= EIG_MAX_DEG, "filter degree must be in [2, 64]");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bX = (size_t)B * N * m * 8, bM = (size_t)B * m * m * 8;
    eig_ws w;
    w.B = B; w.N = N; w.m = m;
    // (a block of one 64 x 64 tile is four workgroups per mesh at 512 rows per chunk: 33 us per product, latency; 128 rows: 16)
    w.kchunk = m <= TN_T ? 128 : 512;
    w.nsplit = dm_cdiv(N, w.kchunk);
    int rc = dm_ws_reserve(ctx, 3 * dm_align_up(bX) + 5 * dm_align_up(bM) + dm_align_up((size_t)w.nsplit * bM) + dm_align_up((size_t)B * m * 8) +
                                    dm_align_up((size_t)B * EIG_MAX_DEG * 3 * 8) + 3 * dm_align_up((size_t)B * 8) + 4096);
    if (rc) return rc;
    double* Ya = (double*)dm_ws_take(ctx, bX);
    double* Yb = (double*)dm_ws_take(ctx, bX);
    double* Yc = (double*)dm_ws_take(ctx, bX);
    double* H = (double*)dm_ws_take(ctx, bM);
    double* V = (double*)dm_ws_take(ctx, bM);
    double* Q = (double*)dm_ws_take(ctx, bM);
    w.T = (double*)dm_ws_take(ctx, bM);
    w.W = (double*)dm_ws_take(ctx, bM);
    w.part = (double*)dm_ws_take(ctx, (size_t)w.nsplit * bM);
    double* theta = (double*)dm_ws_take(ctx, (size_t)B * m * 8);
    double* coef = (double*)dm_ws_take(ctx, (size_t)B * EIG_MAX_DEG * 3 * 8);
    double* lmax = (double*)dm_ws_take(ctx, (size_t)B * 8);
    unsigned long long* rbits = (unsigned long long*)dm_ws_take(ctx, (size_t)B * 8);
    w.fail = (int*)dm_ws_take(ctx, (size_t)B * 4);
    if (!w.fail) return dm_fail(ctx, DM_ENOMEM, "eigenbasis: workspace not reserved");
    DM_CHECK_HIP(ctx, hipMemsetAsync(w.fail, 0, (size_t)B * 4, ctx->stream));
    if (!Ya || !Yb || !Yc || !H || !V || !Q || !w.T || !w.W || !w.part || !theta || !coef || !lmax || !rbits)
        return dm_fail(ctx, DM_ENOMEM, "eigenbasis: workspace not reserved");
    DM_LAUNCH(ctx, "eig_gershgorin", gershgorin_kernel, dim3(B), dim3(256), 0, ell_vals, N, nnz, lmax);
    int spmm_cw = 16;
    while (spmm_cw < m && spmm_cw < 256) spmm_cw <<= 1;
    // (a batch of sixteen meshes or more: the XCD-aware one-dimensional order, see the kernel; fewer would leave XCDs idle)
    const int spmm_xcd = B >= 16 ? B : 0;
    const dim3 gsp = spmm_xcd ? dim3((unsigned)(dm_cdiv(N, 256 / spmm_cw) * dm_cdiv(m, spmm_cw) * dm_cdiv(B, 8) * 8))
                              : dim3(dm_cdiv(N, 256 / spmm_cw), dm_cdiv(m, spmm_cw), B);
    // four columns per thread where the block allows it (spmm_ell4_kernel)
    const bool spmm4 = (m % 4 == 0) && m <= 256;
    int spmm_tpr = 1;
    while (spmm_tpr < m / 4) spmm_tpr <<= 1;
    const int nbx4 = dm_cdiv(N, 256 / spmm_tpr);
    const dim3 gsp4 = spmm_xcd ? dim3((unsigned)(nbx4 * dm_cdiv(B, 8) * 8)) : dim3(nbx4, 1, B);
    auto launch_spmm = [&](const double* yin, const double* yprev, double* ynew, const double* cf, int step) -> int {
        if (spmm4) DM_LAUNCH(ctx, "eig_spmm", spmm_ell4_kernel, gsp4, dim3(256), 0, ell_vals, ell_cols, N, nnz, yin, yprev, ynew, m, cf, step, spmm_tpr, spmm_xcd);
        else DM_LAUNCH(ctx, "eig_spmm", spmm_ell_kernel, gsp, dim3(256), 0, ell_vals, ell_cols, N, nnz, yin, yprev, ynew, m, cf, step, spmm_cw, spmm_xcd);
        return DM_OK;
    };
    const double* Xcur = X;

    if (!warm_start) {      // the caller's X holds a random block: orthonormalise it (unit columns / sqrt(m): singular values <= 1)
        rc = launch_unit_columns(ctx, X, B, N, m, 1.0 / sqrt((double)m));
        if (rc) return rc;
        double* r = nullptr;
        DM_CHECK_HIP(ctx, hipMemcpyAsync(Ya, X, bX, hipMemcpyDeviceToDevice, ctx->stream));
        rc = eig_orthonormalize(ctx, w, Ya, Yb, &r);
        if (rc) return rc;
        DM_CHECK_HIP(ctx, hipMemcpyAsync(X, r, bX, hipMemcpyDeviceToDevice, ctx->stream));
    }
    for (int it = 0; it <= n_iter; ++it) {
        // Rayleigh-Ritz on span(Xcur): H = X^T L X, eigen-decomposition, X <- X Q
        rc = launch_spmm(Xcur, nullptr, Ya, nullptr, 0);
        if (rc) return rc;
        rc = eig_gram(ctx, w, Xcur, Ya, H, 1);
        if (rc) return rc;
        const size_t jac_lds = (size_t)2 * m * m * 8;
        if (jac_lds <= 128 * 1024) {
            int rc_ = dm_grant_lds(ctx, (const void*)jacobi_eigh_kernel<true>, jac_lds);
            if (rc_) return rc_;
            DM_LAUNCH(ctx, "eig_jacobi", jacobi_eigh_kernel<true>, dim3(B), dim3(1024), jac_lds, H, V, m, 30);
        } else {
            DM_LAUNCH(ctx, "eig_jacobi", jacobi_eigh_kernel<false>, dim3(B), dim3(1024), 0, H, V, m, 30);
        }
        DM_LAUNCH(ctx, "eig_ritz_sort", ritz_sort_kernel, dim3(B), dim3(256), (size_t)m * 8 + (size_t)m * 4, (const double*)H, (const double*)V, m,
                  theta, Q);
        rc = eig_apply(ctx, w, Xcur, Q, 1, 0.0, 1.0, Yb);                            // Ritz vectors: (X Q)_ic = sum_k X_ik Q_kc
        if (rc) return rc;
        DM_CHECK_HIP(ctx, hipMemcpyAsync(X, Yb, bX, hipMemcpyDeviceToDevice, ctx->stream));
        if (it == n_iter) break;
        // filter: degree ramps 4, 8, 16, ... up to `degree` on a cold start
        int deg = degree;
        if (!warm_start) { const int ramp = 4 << it; if (it < 8 && ramp < deg) deg = ramp; }
        DM_LAUNCH(ctx, "eig_cheb_coef", cheb_coef_kernel, dim3(B), dim3(64), 0, (const double*)theta, m, (const double*)lmax, deg, coef);
        const double* y0 = X;
        double* bufs[3] = {Ya, Yb, Yc};
        const double* yprev = nullptr;
        const double* ycur = y0;
        for (int j = 1; j <= deg; ++j) {
            double* ynew = bufs[j % 3];
            rc = launch_spmm(ycur, yprev, ynew, coef, j - 1);
            if (rc) return rc;
            yprev = ycur; ycur = ynew;
        }
        // orthonormalise the filtered block
        double* Yf = const_cast<double*>(ycur);
        rc = launch_unit_columns(ctx, Yf, B, N, m, 1.0 / sqrt((double)m));
        if (rc) return rc;
        double* other = (Yf == Ya) ? Yb : Ya;
        double* r = nullptr;
        rc = eig_orthonormalize(ctx, w, Yf, other, &r);
        if (rc) return rc;
        DM_CHECK_HIP(ctx, hipMemcpyAsync(X, r, bX, hipMemcpyDeviceToDevice, ctx->stream));   // (the three Y buffers are scratch again)
    }
    // residual of the k wanted pairs, outputs
    rc = launch_spmm(X, nullptr, Ya, nullptr, 0);
    if (rc) return rc;
    DM_CHECK_HIP(ctx, hipMemsetAsync(rbits, 0, (size_t)B * 8, ctx->stream));
    DM_LAUNCH(ctx, "eig_residual", ritz_residual_kernel, dim3(k, B), dim3(256), 0, (const double*)X, (const double*)Ya, N, m, (const double*)theta, rbits);
    DM_CHECK_HIP(ctx, hipMemcpyAsync(resid, rbits, (size_t)B * 8, hipMemcpyDeviceToDevice, ctx->stream));
    DM_LAUNCH(ctx, "eig_fail", eig_fail_kernel, dim3(dm_cdiv(B, 256)), dim3(256), 0, (const int*)w.fail, B, resid);
    DM_LAUNCH(ctx, "eig_finish", eig_finish_kernel, dim3(k, B), dim3(256), 0, (const double*)X, N, m, k, mass, (const double*)theta, Phi, lam);
    return DM_OK;
}

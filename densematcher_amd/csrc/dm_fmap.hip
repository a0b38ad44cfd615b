// Spectral projection (dm_project) and the functional-map solve (dm_fmap_solve).
//
// Reference arithmetic reproduced (oracle/dm_oracle.py: project, ev_sqdiff, fmap_solve):
//   A = Phi1^T (a1 * F1), B = Phi2^T (a2 * F2)                 pyFM/optimize/base_functions.py:526-532
//   ev_ij = (lam1_j / s - lam2_i / s)^2, s = max(lam1, lam2)   pyFM/functional.py:404-405
//   minimiser of w_d/2 |C A - B|^2 + w_l/2 sum C^2 ev with column 0 pinned (base_functions.py:49,95,759):
//   for every row i:  (P[f,f] + w_l diag(ev[i,f])) C[i,f] = Q[i,f] - P[f,0] C[i,0],
//   P = w_d A A^T, Q = w_d B A^T, f = 1..k1-1   
#include <stdlib.h>

#include "dm_gemm_f64.h"
#include "dm_internal.h"
#include "dm_pcg.h"

// =================================================================================================
// dm_project:  Ared[b] = Phi[b][:, :k]^T (mass[b] * F[b])
// =================================================================================================
struct OutProj {
    float* direct;        // (B, k, D) when nsplit == 1
    double* partial;      // (nsplit, B, k, D) otherwise
    int B, k, D;
    __device__ __forceinline__ void store(int b, int split, int m, int c, double v) const {
        if (direct) direct[((long long)b * k + m) * D + c] = (float)v;
        else partial[(((long long)split * B + b) * k + m) * D + c] = v;
    }
};

__global__ __launch_bounds__(256) void splitk_reduce_f32_kernel(const double* __restrict__ partial, int nsplit,
                                                                long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int q = 0; q < nsplit; ++q) s += partial[(long long)q * n + i];
    out[i] = (float)s;
}

template <typename TR>
static int project_impl(dm_ctx* ctx, int B, int N, int D, int k, const TR* Phi, int ld, const TR* mass,
                        const void* F, int f_dtype, float* Ared) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N > 0 && D > 0 && k > 0, "sizes must be positive");
    DM_REQUIRE(ctx, Phi && mass && F && Ared, "null pointer");
    DM_REQUIRE(ctx, ld >= k, "eigenvector row stride smaller than k");
    const int want_f64 = f_dtype & DM_PROJECT_F64;
    f_dtype &= ~DM_PROJECT_F64;
    DM_REQUIRE(ctx, f_dtype == DM_F16 || f_dtype == DM_F32, "f_dtype must be DM_F16 or DM_F32 (| DM_PROJECT_F64)");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (f_dtype == DM_F16 && !want_f64) return dm_project_f16split<TR>(ctx, B, N, D, k, Phi, ld, mass, F, Ared);
    const int tiles = dm_cdiv(k, TN_T) * dm_cdiv(D, TN_T);
    // split-K by a fixed chunk of vertices: the summation order of a pair must not depend on the batch it is in
    const int kchunk = 512;
    const int nsplit = dm_cdiv(N, kchunk);
    double* partial = nullptr;
    if (nsplit > 1) {
        int rc = dm_ws_reserve(ctx, (size_t)nsplit * B * k * D * 8);
        if (rc) return rc;
        partial = (double*)dm_ws_take(ctx, (size_t)nsplit * B * k * D * 8);
    }
    // the mass multiplies the descriptor rows, as in the reference (A @ descr), the basis stays unscaled
    RowsScaled<TR, TR> opx{Phi, (long long)N * ld, ld, k, nullptr, 0};
    OutProj out{nsplit > 1 ? nullptr : Ared, partial, B, k, D};
    dim3 grid(tiles, nsplit, B);
    if (f_dtype == DM_F16) {
        RowsF16Scaled<TR> opy{(const _Float16*)F, (long long)N * D, D, D, mass, (long long)N};
        DM_LAUNCH(ctx, "project_tn_f64", (gemm_tn_f64<RowsScaled<TR, TR>, RowsF16Scaled<TR>, OutProj>), grid, dim3(256), 0, opx,
                  opy, out, k, D, N, kchunk);
    } else {
        RowsScaled<float, TR> opy{(const float*)F, (long long)N * D, D, D, mass, (long long)N};
        DM_LAUNCH(ctx, "project_tn_f64", (gemm_tn_f64<RowsScaled<TR, TR>, RowsScaled<float, TR>, OutProj>), grid, dim3(256), 0, opx,
                  opy, out, k, D, N, kchunk);
    }
    if (nsplit > 1) {
        const long long n = (long long)B * k * D;
        DM_LAUNCH(ctx, "splitk_reduce", splitk_reduce_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                  partial, nsplit, n, Ared);
    }
    return DM_OK;
}
extern "C" int dm_project(dm_ctx* ctx, int B, int N, int D, int k, const float* Phi, int ld, const float* mass,
                          const void* F, int f_dtype, float* Ared) {
    return project_impl<float>(ctx, B, N, D, k, Phi, ld, mass, F, f_dtype, Ared);
}
extern "C" int dm_project_f64(dm_ctx* ctx, int B, int N, int D, int k, const double* Phi, int ld, const double* mass,
                              const void* F, int f_dtype, float* Ared) {
    return project_impl<double>(ctx, B, N, D, k, Phi, ld, mass, F, f_dtype, Ared);
}


// =================================================================================================
// dm_fmap_c00: sign(Phi1[0,0] Phi2[0,0]) sqrt(area2 / area1)          pyFM/functional.py:654-658
// =================================================================================================
template <typename TR>
__global__ __launch_bounds__(256) void c00_kernel(dm_c00_args<TR> z) {
    __shared__ double red[2][4];
    dm_c00_body<TR>(z, blockIdx.x, threadIdx.x, red);
}

template <typename TR>
static int c00_impl(dm_ctx* ctx, int B, int N1, int N2, const TR* Phi1, int ld1, const TR* Phi2, int ld2,
                    const TR* mass1, const TR* mass2, double* c00) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N1 > 0 && N2 > 0 && ld1 > 0 && ld2 > 0, "sizes must be positive");
    DM_REQUIRE(ctx, Phi1 && Phi2 && mass1 && mass2 && c00, "null pointer");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    DM_LAUNCH(ctx, "c00", c00_kernel<TR>, dim3(B), dim3(256), 0,
              (dm_c00_args<TR>{Phi1, (long long)N1 * ld1, Phi2, (long long)N2 * ld2, mass1, mass2, N1, N2, c00}));
    return DM_OK;
}
extern "C" int dm_fmap_c00(dm_ctx* ctx, int B, int N1, int N2, const float* Phi1, int ld1, const float* Phi2, int ld2,
                           const float* mass1, const float* mass2, double* c00) {
    return c00_impl<float>(ctx, B, N1, N2, Phi1, ld1, Phi2, ld2, mass1, mass2, c00);
}
extern "C" int dm_fmap_c00_f64(dm_ctx* ctx, int B, int N1, int N2, const double* Phi1, int ld1, const double* Phi2, int ld2,
                               const double* mass1, const double* mass2, double* c00) {
    return c00_impl<double>(ctx, B, N1, N2, Phi1, ld1, Phi2, ld2, mass1, mass2, c00);
}

// =================================================================================================
// Gram matrices  PQ[b] = w_d [A; Bm] A^T    ((k1 + k2) x k1, float64)
// =================================================================================================
struct OutScaled {
    double* p; long long stride_b; int ld; double scale;
    // optional second copy of P[1:,1:] in the blocked solver's LDS layout: 16x16 blocks of the lower block
    // triangle, block (I,K) at (I(I+1)/2 + K) * 256, element [c & 15][r & 15] = P[r+1][c+1]  (see fmap_solve_blocked_kernel)
    double* img; long long img_stride_b; int k1;
    __device__ __forceinline__ void store(int b, int i, int j, double v) const {
        const double x = scale * v;
        p[b * stride_b + (long long)i * ld + j] = x;
        if (img && i >= 1 && j >= 1 && i < k1) {
            const int r = i - 1, c = j - 1, I = r >> 4, K = c >> 4;
            if (I >= K) img[b * img_stride_b + (I * (I + 1) / 2 + K) * 256 + (c & 15) * 16 + (r & 15)] = x;
        }
    }
};

// =================================================================================================
// Row-wise SPD solves.  One workgroup per (pair b, row i).  The (n+1) x n augmented lower triangle
// [M_i ; rhs_i^T] (n = k1 - 1) lives in LDS, packed row-major; a right-looking Cholesky turns the
// last row into y = L^-1 rhs, then L^T x = y is solved backwards.
// =================================================================================================
__device__ __forceinline__ int tri(int r) { return r * (r + 1) / 2; }

constexpr int SP_NT = 1024;              // threads per workgroup of the packed solver (16 waves, four per SIMD)
constexpr int SP_NW = SP_NT / 64;
constexpr int SP_U = 2;                   // blocks of the trailing update a wave has in flight per trip

// Packed-storage solver (any n <= 199; the default above the blocked solver's n <= 176).  The lower triangle of
// [P_ff + w_l diag(ev_i) ; rhs] lives in LDS row-packed (159 KiB at n = 199: nothing else fits, so one workgroup per
// CU), and the workgroup is made of 16 waves because every phase is bound by instruction issue, not by the LDS or the
// matrix cores: four waves per SIMD interleave where one wave would wait out each dependent step.
__global__ __launch_bounds__(SP_NT) void fmap_solve_kernel(const double* __restrict__ PQ, const double* __restrict__ lam1,
                                                           const double* __restrict__ lam2, const double* __restrict__ c00,
                                                           double w_lap, int k1, int k2, double* __restrict__ C,
                                                           int32_t* __restrict__ info, int dbg) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = k1 - 1;
    double* M = sm;                         // tri(n + 1) entries
    double* col = sm + tri(n + 1);          // max(n + 1, 16): reduction scratch
    double* red = col + max(n + 1, 16);     // 4: [0] eigenvalue scale, [1] failure flag, [2] scratch word for masked stores, [3] = 0.0 for masked loads
    const int dummy = (int)(red + 2 - M), zero = (int)(red + 3 - M);
    const int b = blockIdx.y, i = blockIdx.x, t = threadIdx.x;
    const int lane = t & 63;
    const int swave = __builtin_amdgcn_readfirstlane(t >> 6);
    const double* P = PQ + (long long)b * (k1 + k2) * k1;
    const double* Q = P + (long long)k1 * k1;
    const double* l1 = lam1 + (long long)b * k1;
    const double* l2 = lam2 + (long long)b * k2;

    // scale = max(lam1.max(), lam2.max())   (functional.py:404)
    double mx = -DM_INF_F64;
    for (int q = t; q < k1; q += SP_NT) mx = fmax(mx, l1[q]);
    for (int q = t; q < k2; q += SP_NT) mx = fmax(mx, l2[q]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if (lane == 0) col[swave] = mx;
    __syncthreads();
    if (t == 0) {
        double m = col[0];
        for (int q = 1; q < SP_NW; ++q) m = fmax(m, col[q]);
        red[0] = m;
        red[1] = 0.0;
        red[3] = 0.0;
    }
    __syncthreads();
    const double scale = red[0];
    const double ci0 = (i == 0) ? c00[b] : 0.0;           // get_x0: column 0 is (c00, 0, ..., 0)^T
    const double l2i = l2[i] / scale;

    // load [M ; rhs]
    for (int r = t >> 4; r <= n; r += SP_NT / 16) {
        for (int c = t & 15; c <= r && c < n; c += 16) {
            double v;
            if (r < n) {
                v = P[(long long)(r + 1) * k1 + (c + 1)];
                if (r == c) {
                    const double d = l1[c + 1] / scale - l2i;
                    v += w_lap * (d * d);
                }
            } else {
                v = Q[(long long)i * k1 + (c + 1)] - P[(long long)(c + 1) * k1] * ci0;
            }
            M[tri(r) + c] = v;
        }
    }
    __syncthreads();

    // Right-looking Cholesky on the packed lower triangle, four columns per step; the right-hand side is carried as
    // row n, so the forward substitution happens in the same sweep.
    //   (1) waves 0-3 (the threads that own a row) factor the 4x4 diagonal block in registers (10 broadcast LDS reads);
    //   (2) thread t finishes the four panel entries of row j + w + t by a 4-term forward substitution;
    //   (3) the trailing triangle takes its rank-4 update on the f64 matrix cores, 16x16 blocks straight from / to the
    //       packed storage: D = (-L_panel[R]) L_panel[C]^T + M[R][C]  (v_mfma_f64_16x16x4_f64, K = the panel width).
    const int li = lane & 15, lq = lane >> 4;
    int x4[4], txl[4], dl[4];                                 // lane parts of the packed addresses (see step 3)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        x4[r4] = lq + 4 * r4;
        txl[r4] = tri(x4[r4]) + li;
        dl[r4] = li - x4[r4];
    }
    const int tli = tri(li) + lq;
    const int Rmax = n >> 4;                                  // last (possibly partial) block row; row n = right-hand side
    const bool plastA = li <= n - Rmax * 16, pcn = li < n - Rmax * 16;
    bool plast[4], pdiag[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        plast[r4] = x4[r4] <= n - Rmax * 16;
        pdiag[r4] = dl[r4] <= 0;
    }
    for (int j = 0; j < n && dbg != 3; j += 4) {
        const int w = min(4, n - j);
        double d[4][4], inv[4];
        if (swave < 4) {
            // (1) diagonal block, identity-padded beyond w
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c <= q; ++c) d[q][c] = (q < w) ? M[tri(j + q) + j + c] : (q == c ? 1.0 : 0.0);
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int c = 0; c < q; ++c) {
                    double sdot = d[q][c];
#pragma unroll
                    for (int e = 0; e < c; ++e) sdot -= d[q][e] * d[c][e];
                    d[q][c] = sdot * inv[c];
                }
                double piv = d[q][q];
#pragma unroll
                for (int e = 0; e < q; ++e) piv -= d[q][e] * d[q][e];
                ok = ok && (piv > 0.0);
                double rs = __builtin_amdgcn_rsq(piv);       // piv^-1/2: hardware seed + two Newton steps, no divide / sqrt
                const double hp = 0.5 * piv;
                rs = rs * (1.5 - hp * rs * rs);
                rs = rs * (1.5 - hp * rs * rs);
                inv[q] = rs;
                d[q][q] = rs;                                // the diagonal stores 1 / L[q][q] (all later uses multiply)
            }
            if (!ok && t == 0 && dbg != 1 && dbg != 4 && dbg != 5) red[1] = 1.0;
        }
        __syncthreads();                         // the diagonal block has been read by everyone who needs it
        if (red[1] != 0.0) break;                // uniform: one LDS word
        // (2) rows below the panel (the right-hand side row n included); one thread also stores the block's factor
        if (swave < 4) {
            const int r = j + w + t;
            if (r <= n) {
                double* Mr = M + tri(r) + j;
                double x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q < w) {
                        double sdot = Mr[q];
#pragma unroll
                        for (int e = 0; e < q; ++e) sdot -= x[e] * d[q][e];
                        x[q] = sdot * inv[q];
                        Mr[q] = x[q];
                    } else {
                        x[q] = 0.0;
                    }
                }
            }
            if (t == 255) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c <= q; ++c)
                        if (q < w) M[tri(j + q) + j + c] = d[q][c];
            }
        }
        __syncthreads();
        // (3) trailing update.  Blocks are aligned to multiples of 16; entries left of column j + w (already final),
        // above the diagonal, or beyond the matrix are masked.  A wave takes every 16th block of the trailing block
        // triangle, SP_U at a time; all loads are unconditional (masked lanes read M[0], masked stores go to a scratch
        // word) so that the code is straight-line and the LDS reads of a trip are in flight together.
        // Addresses: tri(16 R + x) = tri(16 R) + 16 R x + tri(x) with the lane parts (x, tri(x)) hoisted out of the
        // loop and the block parts on the scalar unit.
        const int jw = j + w;
        if (jw < n && dbg != 1) {
            const int Cmin = jw >> 4, T = Rmax - Cmin + 1;
            const int nblk = T * (T + 1) / 2;
            const bool kv = lq < w;
            const bool pfirst = li >= jw - Cmin * 16;           // column / row not yet final (first block row / column only)
            const bool fullc = (jw & 15) == 0, fullr = (n & 15) == 15;   // first block column / last block row complete
            int rho = 0, gam = swave;                         // block (rho, gam), gam <= rho, of the trailing triangle
            while (gam > rho) { gam -= rho + 1; ++rho; }
            for (int idx = swave; idx < nblk; idx += SP_NW * SP_U) {
                f64x4 acc[SP_U];
                double av[SP_U], bv[SP_U];
                int addr[SP_U][4];
                bool val[SP_U][4];
                int Rs[SP_U], Cs[SP_U];
#pragma unroll
                for (int u = 0; u < SP_U; ++u) {
                    Rs[u] = Cmin + rho; Cs[u] = Cmin + gam;
                    gam += SP_NW;
                    while (gam > rho) { gam -= rho + 1; ++rho; }
                }
#pragma unroll
                for (int u = 0; u < SP_U; ++u) {
                    // every lane predicate below is a per-kernel or per-panel constant; the block only selects which
                    // of them apply (scalar conditions), so a block costs no vector compares
                    const bool bvld = idx + SP_NW * u < nblk;
                    const int R = Rs[u], Cb = Cs[u];
                    const int s16R = R * 16, s16C = Cb * 16;
                    const int TRj = (s16R * (s16R + 1) >> 1) + j, TCj = (s16C * (s16C + 1) >> 1) + j;
                    const int TRC = (s16R * (s16R + 1) >> 1) + s16C;
                    if (bvld && w == 4 && (Cb > Cmin || fullc) && Cb < R && (R < Rmax || fullr)) {
                        // interior block (wave-uniform test): nothing is masked
                        av[u] = M[(int)__umul24(li, s16R) + tli + TRj];
                        bv[u] = M[(int)__umul24(li, s16C) + tli + TCj];
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            val[u][r4] = true;
                            addr[u][r4] = (int)__umul24(x4[r4], s16R) + txl[r4] + TRC;
                            acc[u][r4] = M[addr[u][r4]];
                        }
                    } else {
                        const bool cok = bvld && (Cb != Rmax || pcn) && (Cb != Cmin || pfirst);
                        const bool aok = bvld && kv && (R != Rmax || plastA) && (R != Cmin || pfirst);
                        const bool bok = cok && kv;
                        av[u] = M[aok ? (int)__umul24(li, s16R) + tli + TRj : zero];
                        bv[u] = M[bok ? (int)__umul24(li, s16C) + tli + TCj : zero];
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            val[u][r4] = cok && (R != Rmax || plast[r4]) && (R != Cb || pdiag[r4]);
                            addr[u][r4] = val[u][r4] ? (int)__umul24(x4[r4], s16R) + txl[r4] + TRC : zero;
                            acc[u][r4] = M[addr[u][r4]];
                        }
                    }
                    av[u] = -av[u];
                }
                if (dbg == 4) {                                 // experiment: loads only
                    double sink = 0.0;
#pragma unroll
                    for (int u = 0; u < SP_U; ++u) sink += av[u] + bv[u] + acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
                    if (sink == 1.2345) M[0] = sink;
                    continue;
                }
#pragma unroll
                for (int u = 0; u < SP_U; ++u) acc[u] = mfma_f64_16x16x4(av[u], bv[u], acc[u]);
                if (dbg == 5) {                                 // experiment: no stores
                    double sink = 0.0;
#pragma unroll
                    for (int u = 0; u < SP_U; ++u) sink += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
                    if (sink == 1.2345) M[0] = sink;
                    continue;
                }
#pragma unroll
                for (int u = 0; u < SP_U; ++u)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) M[val[u][r4] ? addr[u][r4] : dummy] = acc[u][r4];   // masked lanes -> scratch word
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (red[1] != 0.0) {
        if (t == 0) atomicMax(&info[b], i + 1);
        for (int c = t; c < k1; c += SP_NT) C[((long long)b * k2 + i) * k1 + c] = (c == 0) ? ci0 : 0.0;
        return;
    }
    // back substitution L^T x = y (y = row n), four unknowns per step from the bottom: the part of the four dot
    // products below the panel is spread over the threads that own a row and summed through LDS, the 4x4 triangle is
    // finished by every thread redundantly.
    double* y = M + tri(n);
    for (int j = ((n - 1) >> 2) << 2; j >= 0 && dbg != 2; j -= 4) {
        const int w = min(4, n - j);
        if (swave < 4) {
            double sdot[4] = {0.0, 0.0, 0.0, 0.0};
            const int r = j + w + t;
            if (r < n) {
                const double xr = y[r];
                const double* Mr = M + tri(r) + j;
#pragma unroll
                for (int q = 0; q < 4; ++q) sdot[q] = (q < w) ? Mr[q] * xr : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sdot[q] += dpp_f64<0xB1>(sdot[q]); sdot[q] += dpp_f64<0x4E>(sdot[q]);
                sdot[q] += dpp_f64<0x141>(sdot[q]); sdot[q] += dpp_f64<0x140>(sdot[q]);
                sdot[q] += __shfl_xor(sdot[q], 16); sdot[q] += __shfl_xor(sdot[q], 32);
            }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) col[swave * 4 + q] = sdot[q];
            }
        }
        __syncthreads();
        if (t == 0) {
            double x[4];
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                if (q < w) {
                    double v = y[j + q] - (col[q] + col[4 + q] + col[8 + q] + col[12 + q]);
#pragma unroll
                    for (int e = 3; e > q; --e)
                        if (e < w) v -= M[tri(j + e) + j + q] * x[e];
                    x[q] = v * M[tri(j + q) + j + q];       // (reciprocal diagonal)
                    y[j + q] = x[q];
                } else {
                    x[q] = 0.0;
                }
            }
        }
        __syncthreads();
    }
    double* Crow = C + ((long long)b * k2 + i) * k1;
    if (t == 0) Crow[0] = ci0;
    for (int c = t; c < n; c += SP_NT) Crow[c + 1] = y[c];
}

#ifdef DM_SOLVE_TIMING
// phase cycle counters of workgroup (0,0), thread 0 (experiment builds only: tools/solve_timing.py)
__device__ long long g_solve_dbg[16];
#define DM_SOLVE_TIMING_HAVE_COUNTERS 1
extern "C" int dm_debug_solve_timing(long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_solve_dbg), sizeof(long long) * 16) == hipSuccess ? 0 : -3;
}
#define DBG_T0() long long _t0 = clock64(); const bool _dbg = (blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0);
#define DBG_ACC(slot) { const long long _t1 = clock64(); if (_dbg) g_solve_dbg[slot] += _t1 - _t0; _t0 = _t1; }
#else
#define DBG_T0()
#define DBG_ACC(slot)
#endif

#include "dm_chol.h"

// =================================================================================================
// Blocked variant (n <= 176): the matrix lives in LDS as 16x16 blocks, lower block triangle, each
// block stored TRANSPOSED (T_IK[k][i] = A[I*16+i][K*16+k]) so that every f64-MFMA operand and
// result access is a lane-contiguous, conflict-free ds_read/ds_write_b64.  Per block column J:
//   (a) wave 0: Cholesky of the 16x16 diagonal block + W = L_JJ^-1 (the only serial part),
//   (b) panel:   L_IJ = A_IJ L_JJ^-T        ==  T_IJ <- W T_IJ                    (MFMA)
//   (c) update:  A_IK -= L_IJ L_KJ^T        ==  T_IK <- T_IK - L_KJ L_IJ^T        (MFMA)
// the right-hand side is carried along (forward solve), then L^T x = y runs block-backwards with
// the stored W_J.  Padding rows/columns (n..16*NB) are identity.
// =================================================================================================
__global__ __launch_bounds__(256) void fmap_solve_blocked_kernel(const double* __restrict__ PQ, const double* __restrict__ Timg,
                                                                 const double* __restrict__ lam1,
                                                                 const double* __restrict__ lam2, const double* __restrict__ c00,
                                                                 double w_lap, int k1, int k2, int NB, double* __restrict__ C,
                                                                 int32_t* __restrict__ info, const int32_t* __restrict__ only_if, int only_ng) {
    if (only_if && !pcg_flagged(only_if, only_ng, (int)blockIdx.y)) return;       // (the fall-back launch of the batched iteration: flagged pairs only)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = k1 - 1;
    const int nblk = NB * (NB + 1) / 2;
    double* T = sm;                         // nblk blocks of 256
    double* LT = T + nblk * 256;            // L_JJ^T scratch (row j = column j of L)
    double* Ws = LT + 256;                  // W^T scratch:  Ws[m*16 + k] = W[k][m]
    double* rhs = Ws + 256;                 // NB*16  (becomes y)
    double* xv = rhs + NB * 16;             // NB*16
    double* invp = xv + NB * 16;            // 16
    double* red = invp + 16;                // 8: [0..3] wave maxima, [4] scale, [5] failure flag
    const int b = blockIdx.y, i = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const double* P = PQ + (long long)b * (k1 + k2) * k1;
    const double* Q = P + (long long)k1 * k1;
    const double* l1 = lam1 + (long long)b * k1;
    const double* l2 = lam2 + (long long)b * k2;

    DBG_T0()
    // ---- image load, first thing: the gram kernel left P[1:,1:] in exactly the LDS layout (T_IK[kk][ii] = M[I*16+ii][K*16+kk]),
    // so the copy is nblk * 2 contiguous 1 KiB pieces that go global -> LDS by LDS-DMA (no registers); they are in flight while
    // the eigenvalue scale and the diagonal penalties are computed and are awaited in front of the first use
    {
        typedef __attribute__((address_space(1))) const void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const char* src = reinterpret_cast<const char*>(Timg + (long long)b * nblk * 256) + lane * 16;
        for (int pc = wave; pc < nblk * 2; pc += 4)                   // piece = 1 KiB = half a block
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (long long)pc * 1024), (lptr_t)(T + pc * 128), 16, 0, 0);
    }
    double mx = -DM_INF_F64;
    for (int q = t; q < k1; q += 256) mx = fmax(mx, l1[q]);
    for (int q = t; q < k2; q += 256) mx = fmax(mx, l2[q]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (t == 0) {
        red[4] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        red[5] = 0.0;
    }
    __syncthreads();
    const double scale = red[4];
    const double ci0 = (i == 0) ? c00[b] : 0.0;
    const double l2i = l2[i] / scale;
    DBG_ACC(5)
    // diagonal penalty w_lap ev[i][c+1], staged in LDS (LT is otherwise unused) so the block loop below has no
    // dependent global loads
    for (int c = t; c < n; c += 256) {
        const double d = l1[c + 1] / scale - l2i;
        LT[c] = w_lap * (d * d);
    }
    __syncthreads();
    DBG_ACC(6)

    // ---- right-hand side; then the image must have landed: the row's diagonal penalty is added and the padding rows /
    // columns (>= n) are made identity
    for (int c = t; c < NB * 16; c += 256)
        rhs[c] = (c < n) ? Q[(long long)i * k1 + (c + 1)] - P[(long long)(c + 1) * k1] * ci0 : 0.0;
    __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): this wave's LDS-DMA pieces
    DBG_ACC(8)
    __syncthreads();
    for (int c = t; c < NB * 16; c += 256) {
        const int I = c >> 4;
        double* dg = T + (I * (I + 1) / 2 + I) * 256 + (c & 15) * 17;
        if (c < n) *dg += LT[c]; else *dg = 1.0;
    }
    __syncthreads();
    // (u -> row, column) of a lower-triangular block enumeration, used by the trailing update; LT is free again
    int* tri_rc = reinterpret_cast<int*>(LT);
    for (int u = t; u < 128; u += 256) {
        int a_ = 0;
        while ((a_ + 1) * (a_ + 2) / 2 <= u) ++a_;
        tri_rc[u] = (a_ << 8) | (u - a_ * (a_ + 1) / 2);
    }
    __syncthreads();
    DBG_ACC(0)

    const bool solved = blocked_chol_solve(T, Ws, rhs, xv, red, tri_rc, NB, t, lane, wave);
    DBG_ACC(3)
    double* Crow = C + ((long long)b * k2 + i) * k1;
    if (!solved) {
        if (t == 0) atomicMax(&info[b], i + 1);
        for (int c = t; c < k1; c += 256) Crow[c] = (c == 0) ? ci0 : 0.0;
        return;
    }
    if (t == 0) Crow[0] = ci0;
    for (int c = t; c < n; c += 256) Crow[c + 1] = xv[c];
    DBG_ACC(4)
}

// =================================================================================================
// Two-phase blocked solver for 177 <= n <= 208 (NB = 12, 13): the 16x16-blocked lower triangle (up to 182 KiB) does not
// fit the LDS, so the factorisation runs as a 2 x 2 block elimination with NA = ceil(NB / 2) leading block rows:
//   phase 1  [A11; A21] resident (NA (NA+1)/2 + (NB-NA) NA blocks): columns 0 .. NA-1 are eliminated (blocked_chol_phase1),
//            giving L11, L21, y1 and b2 - L21 y1;  L11 (with the W_J) is spilled to a per-workgroup global scratch,
//   phase 2  A22 is loaded into the freed leading slots, S = A22 - L21 L21^T on the f64 matrix cores, then the ordinary
//            blocked solve of S x2 = rhs2 (blocked_chol_solve),
//   back     y1 -= L21^T x2, L11 is read back over S, x1 = L11^-T y1.
// One workgroup per CU (148 KiB of LDS at NB = 13), persistent over the (pair, row) systems so that the spill area is
// NA (NA+1)/2 blocks per RESIDENT workgroup (14 MiB in all: it never leaves the L2 / Infinity Cache).
// =================================================================================================
__global__ __launch_bounds__(256) void fmap_solve_2phase_kernel(const double* __restrict__ PQ, const double* __restrict__ Timg,
                                                                const double* __restrict__ lam1, const double* __restrict__ lam2,
                                                                const double* __restrict__ c00, double w_lap, int k1, int k2,
                                                                int NB, int B, double* __restrict__ spill, double* __restrict__ C,
                                                                int32_t* __restrict__ info, const int32_t* __restrict__ only_if, int only_ng) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = k1 - 1;
    const int NA = (NB + 1) / 2, NBr = NB - NA;
    const int r1n = NA * (NA + 1) / 2, r2n = NBr * NA;
    const int nblk_img = NB * (NB + 1) / 2;
    const PanelSlots sl{NA};
    double* T = sm;                          // r1n + r2n blocks of 256
    double* LT = T + (r1n + r2n) * 256;      // 256: diagonal penalties of the current system
    double* Ws = LT + 256;                   // 256
    double* rhs = Ws + 256;                  // NB*16
    double* xv = rhs + NB * 16;              // NB*16
    double* red = xv + NB * 16;              // 8
    int* blk = reinterpret_cast<int*>(red + 8);      // <= 96: resident blocks of phase 1, K-major
    int* cstart = blk + 96;                          // <= 16
    int* tri_rc = cstart + 16;                       // 128: relative triangle enumeration of blocked_chol_solve
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double* my_spill = spill + (long long)blockIdx.x * r1n * 256;

    if (t == 0) {
        int u = 0;
        for (int K = 0; K < NA; ++K) {
            cstart[K] = u;
            for (int I = K; I < NB; ++I) blk[u++] = (I << 8) | K;
        }
        cstart[NA] = u;
    }
    for (int u = t; u < 128; u += 256) {
        int a_ = 0;
        while ((a_ + 1) * (a_ + 2) / 2 <= u) ++a_;
        tri_rc[u] = (a_ << 8) | (u - a_ * (a_ + 1) / 2);
    }
    __syncthreads();
    const int nres = cstart[NA];

    for (long long sys = blockIdx.x; sys < (long long)B * k2; sys += gridDim.x) {
        const int b = (int)(sys / k2), i = (int)(sys - (long long)b * k2);
        if (only_if && !pcg_flagged(only_if, only_ng, b)) continue;   // (uniform: the fall-back launch of the batched iteration, flagged pairs only)
        const double* P = PQ + (long long)b * (k1 + k2) * k1;
        const double* Q = P + (long long)k1 * k1;
        const double* l1 = lam1 + (long long)b * k1;
        const double* l2 = lam2 + (long long)b * k2;
        const double* img = Timg + (long long)b * nblk_img * 256;
        double* Crow = C + ((long long)b * k2 + i) * k1;

        double mx = -DM_INF_F64;
        for (int q = t; q < k1; q += 256) mx = fmax(mx, l1[q]);
        for (int q = t; q < k2; q += 256) mx = fmax(mx, l2[q]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        if (t == 0) {
            red[4] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
            red[5] = 0.0;
        }
        __syncthreads();
        const double scale = red[4];
        const double ci0 = (i == 0) ? c00[b] : 0.0;
        const double l2i = l2[i] / scale;
        for (int c = t; c < NB * 16; c += 256) {
            double pen = 0.0;
            if (c < n) { const double d = l1[c + 1] / scale - l2i; pen = w_lap * (d * d); }
            LT[c] = pen;
            rhs[c] = (c < n) ? Q[(long long)i * k1 + (c + 1)] - P[(long long)(c + 1) * k1] * ci0 : 0.0;
        }
        // ---- phase 1 image: the resident blocks, 16-byte streams from the Gram kernel's pre-blocked copy
        for (int q = t; q < nres * 128; q += 256) {
            const int u = q >> 7, w = q & 127;
            const int rc = blk[u];
            const int I = rc >> 8, K = rc & 255;
            reinterpret_cast<f64x2*>(T + sl(I, K))[w] = reinterpret_cast<const f64x2*>(img + (long long)(I * (I + 1) / 2 + K) * 256)[w];
        }
        __syncthreads();
        for (int c = t; c < NA * 16; c += 256) {                  // (NA * 16 <= n: no padding rows in the leading part)
            const int I = c >> 4;
            T[sl(I, I) + (c & 15) * 17] += LT[c];
        }
        __syncthreads();
        bool ok = blocked_chol_phase1(T, sl, Ws, rhs, xv, red, blk, cstart, NB, t, lane, wave);
        if (ok) {
            // ---- spill L11 (and the W_J parked on its diagonal)
            for (int q = t; q < r1n * 128; q += 256) reinterpret_cast<f64x2*>(my_spill)[q] = reinterpret_cast<const f64x2*>(T)[q];
            __syncthreads();
            // ---- A22 into the leading slots (ordinary triangle layout of an NBr x NBr block matrix)
            for (int q = t; q < NBr * (NBr + 1) / 2 * 128; q += 256) {
                const int u = q >> 7, w = q & 127;
                const int rc = tri_rc[u];
                const int Ir = rc >> 8, Kr = rc & 255, I = NA + Ir, K = NA + Kr;
                reinterpret_cast<f64x2*>(T + (Ir * (Ir + 1) / 2 + Kr) * 256)[w] =
                    reinterpret_cast<const f64x2*>(img + (long long)(I * (I + 1) / 2 + K) * 256)[w];
            }
            __syncthreads();
            for (int c = NA * 16 + t; c < NB * 16; c += 256) {
                const int Ir = (c >> 4) - NA;
                double* dg = T + (Ir * (Ir + 1) / 2 + Ir) * 256 + (c & 15) * 17;
                if (c < n) *dg += LT[c]; else *dg = 1.0;          // padding rows / columns are identity
            }
            __syncthreads();
            // ---- S_IK = A22_IK - sum_J L_IJ L_KJ^T   (transposed storage: T_IK -= L_KJ L_IJ^T)
            {
                const int o0 = (lane >> 4) * 16 + (lane & 15);
                const int nS = NBr * (NBr + 1) / 2;
                for (int u = wave; u < nS; u += 4) {
                    const int rc = tri_rc[u];
                    const int Ir = rc >> 8, Kr = rc & 255;
                    double* Tik = T + (Ir * (Ir + 1) / 2 + Kr) * 256;
                    f64x4 acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = Tik[o0 + r * 64];
                    for (int J = 0; J < NA; ++J) {
                        const double* Tij = T + sl(NA + Ir, J);
                        const double* Tkj = T + sl(NA + Kr, J);
                        double opa[4], opb[4];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) { opa[ks] = -Tkj[o0 + ks * 64]; opb[ks] = Tij[o0 + ks * 64]; }
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) acc = mfma_f64_16x16x4(opa[ks], opb[ks], acc);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) Tik[o0 + r * 64] = acc[r];
                }
            }
            __syncthreads();
            ok = blocked_chol_solve(T, Ws, rhs + NA * 16, xv + NA * 16, red, tri_rc, NBr, t, lane, wave);
        }
        if (ok) {
            // ---- y1 -= L21^T x2:  (L_IK^T x_I)[k] = sum_i T_IK[k*16 + i] x_I[i]
            if (t < NA * 16) {
                const int K = t >> 4, k = t & 15;
                double a0 = 0.0;
                for (int I = NA; I < NB; ++I) {
                    const double* Tik = T + sl(I, K);
#pragma unroll
                    for (int ii = 0; ii < 16; ++ii) a0 = fma(Tik[k * 16 + ii], xv[I * 16 + ii], a0);
                }
                rhs[K * 16 + k] -= a0;
            }
            __syncthreads();
            for (int q = t; q < r1n * 128; q += 256) reinterpret_cast<f64x2*>(T)[q] = reinterpret_cast<const f64x2*>(my_spill)[q];
            __syncthreads();
            blocked_back_subst(T, TriSlots{}, rhs, xv, NA, t);
            if (t == 0) Crow[0] = ci0;
            for (int c = t; c < n; c += 256) Crow[c + 1] = xv[c];
        } else {
            if (t == 0) atomicMax(&info[b], i + 1);
            for (int c = t; c < k1; c += 256) Crow[c] = (c == 0) ? ci0 : 0.0;
        }
        __syncthreads();
    }
}

// =================================================================================================
// Register-resident variant (n <= 128): one WAVE per system, the block triangle in VGPRs (dm_chol_reg.h), four independent
// systems per CU (one per SIMD), no LDS image, no barriers.  NBT = block rows of the instantiation (the system is padded
// with identity up to 16 NBT); the image the Gram kernel wrote has NBimg = ceil(n / 16) <= NBT block rows.
// =================================================================================================
#include "dm_chol_reg.h"
#ifndef DM_SOLVE_REG_EV
#define DM_SOLVE_REG_EV 3                  // block columns of the NBT = 8 instantiation whose panel blocks wait in LDS
#endif

constexpr int solve_reg_ev(int NBT) { return NBT == 8 ? DM_SOLVE_REG_EV : 0; }
constexpr size_t solve_reg_lds(int NBT) {                  // the larger of: the pair's image (staging), the parked panel blocks of 4 waves
    const size_t img = (size_t)(NBT * (NBT + 1) / 2) * 256 * sizeof(double);
    const size_t ev = (size_t)4 * dmreg::ev_slot(NBT, solve_reg_ev(NBT), solve_reg_ev(NBT) + 1) * 256 * sizeof(double);
    return img > ev ? img : ev;
}

template <int NBT>
__global__ __launch_bounds__(256, 1) void fmap_solve_reg_kernel(const double* __restrict__ PQ, const double* __restrict__ Timg,
                                                                const double* __restrict__ lam1, const double* __restrict__ lam2,
                                                                const double* __restrict__ c00, double w_lap, int k1, int k2, int NBimg,
                                                                long long nsys, double* __restrict__ C, int32_t* __restrict__ info,
                                                                const int32_t* __restrict__ only_if, int only_ng) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long sys0 = (long long)blockIdx.x * 4;
    // only_if (nullable): the direct solver as the fall-back of the batched iteration (dm_pcg.h) -- only the pairs it flagged
    if (only_if && !pcg_flagged(only_if, only_ng, (int)(sys0 / k2)) && !pcg_flagged(only_if, only_ng, (int)(min(sys0 + 3, nsys - 1) / k2))) return;
    const long long sys = min(sys0 + wave, nsys - 1);        // (a wave past the end repeats the last system: it takes part in the barriers)
    const int b = (int)(sys / k2), i = (int)(sys - (long long)b * k2);
    const int n = k1 - 1, c = lane & 15, g = lane >> 4;
    const double* P = PQ + (long long)b * (k1 + k2) * k1;
    const double* Q = P + (long long)k1 * k1;
    const double* l1 = lam1 + (long long)b * k1;
    const double* l2 = lam2 + (long long)b * k2;
    extern __shared__ __attribute__((aligned(16))) double sh_dyn[];      // image staging, later the parked panel blocks (host: solve_reg_lds)
    f64x4 T[NBT * (NBT + 1) / 2];
    // ---- image: block (I, K), register r of lane l = entry 64 r + l of the block (the accumulator layout of the transposed block).
    // The four systems of a workgroup normally belong to one pair and share its image: it is fetched from L2 ONCE per workgroup
    // into LDS and distributed from there (four private fetches of 72 KiB per CU and round were 10 % of the kernel).
    {
        const int nblk = NBimg * (NBimg + 1) / 2;
        const bool one_pair = (int)(sys0 / k2) == (int)(min(sys0 + 3, nsys - 1) / k2);          // uniform over the workgroup
        const double* imgb = Timg + (long long)b * nblk * 256;
        if (one_pair) {
            const f64x2* src = reinterpret_cast<const f64x2*>(imgb);
            f64x2* dst = reinterpret_cast<f64x2*>(sh_dyn);
            for (int q = threadIdx.x; q < nblk * 128; q += 256) dst[q] = src[q];
            __syncthreads();
        }
#pragma unroll
        for (int I = 0; I < NBT; ++I)
#pragma unroll
            for (int K = 0; K <= I; ++K) {
                f64x4 v = {0.0, 0.0, 0.0, 0.0};
                if (I < NBimg) {                             // uniform
                    if (one_pair) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = sh_dyn[dmreg::blk(I, K) * 256 + 64 * r + lane];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = imgb[(long long)dmreg::blk(I, K) * 256 + 64 * r + lane];
                    }
                }
                T[dmreg::blk(I, K)] = v;
            }
        __syncthreads();                                     // every wave has its copy before any wave parks blocks in the same LDS
    }
    // scale = max(lam1.max(), lam2.max())   (functional.py:404)
    double mx = -DM_INF_F64;
    for (int q = lane; q < k1; q += 64) mx = fmax(mx, l1[q]);
    for (int q = lane; q < k2; q += 64) mx = fmax(mx, l2[q]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    const double scale = mx;
    const double ci0 = (i == 0) ? c00[b] : 0.0;             // get_x0: column 0 is (c00, 0, ..., 0)^T
    const double l2i = l2[i] / scale;
    // ---- diagonal penalty w_lap ev[i][idx + 1] on the diagonal entries (lane (c, c & 3), register c >> 2 of block (I, I)),
    // identity on the padding; right-hand side in the first row of block row NBT (lanes c == 0)
#pragma unroll
    for (int I = 0; I < NBT; ++I) {
        const int idx = I * 16 + c;
        double pen = 1.0;
        if (idx < n) {
            const double d = l1[idx + 1] / scale - l2i;
            pen = w_lap * (d * d);
        }
        const bool mine = g == (c & 3);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool hit = mine && (c >> 2) == r;
            const double cur = T[dmreg::blk(I, I)][r];
            T[dmreg::blk(I, I)][r] = hit ? (idx < n ? cur + pen : 1.0) : cur;
        }
    }
    double* Crow = C + ((long long)b * k2 + i) * k1;
    // the right-hand side is fetched now, with the image, and waits in LDS (1 KiB per wave; wave-private, no barrier) until
    // the forward substitution: loading it there would expose a global round trip in front of a dependent chain
    __shared__ double sh_rhs[4][NBT * 16];
    for (int q = lane; q < NBT * 16; q += 64)
        sh_rhs[wave][q] = (q < n) ? Q[(long long)i * k1 + (q + 1)] - (i == 0 ? P[(long long)(q + 1) * k1] * ci0 : 0.0) : 0.0;   // (ci0 = 0 for i > 0)
    auto rhs = [&](int J) {
        f64x4 rv = {0.0, 0.0, 0.0, 0.0};
        if (c == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) rv[r] = sh_rhs[wave][J * 16 + g + 4 * r];
        }
        return rv;
    };
    auto store = [&](int J, const f64x4& x) {
        if (c == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ix = J * 16 + g + 4 * r;
                if (ix < n) Crow[ix + 1] = x[r];
            }
        }
    };
    // NBT = 8: the panel blocks of the first block column wait in LDS for the back substitution (dm_chol_reg.h, EV)
    constexpr int EV = solve_reg_ev(NBT);
    if (sys0 + wave >= nsys) return;                         // (no barrier below)
    const bool solved = dmreg::solve<NBT, EV>(T, rhs, store, lane, sh_dyn + wave * (dmreg::ev_slot(NBT, EV, EV + 1) * 256));
    if (!solved) {
        if (lane == 0) atomicMax(&info[b], i + 1);
        for (int q = lane; q < k1; q += 64) Crow[q] = (q == 0) ? ci0 : 0.0;
        return;
    }
    if (lane == 0) Crow[0] = ci0;
}

// workspace of the closed-form solve (fmap_solve_core)
static size_t fmap_solve_ws(const dm_ctx* ctx, int B, int k1, int k2) {
    const size_t pq_bytes = (size_t)B * (k1 + k2) * k1 * 8;
    const int n = k1 - 1, NB = (n + 15) / 16;
    const bool two_phase = (NB == 12 || NB == 13) && !ctx->opt_solve_packed;
    const bool blocked = ((n >= 1 && NB <= 11) && !ctx->opt_solve_packed) || two_phase;
    const size_t img_bytes = blocked ? (size_t)B * (NB * (NB + 1) / 2) * 256 * 8 : 0;
    const int NA = (NB + 1) / 2, grid2 = ctx->n_cu > 0 ? ctx->n_cu : 256;
    const size_t spill_bytes = two_phase ? (size_t)grid2 * (NA * (NA + 1) / 2) * 256 * 8 : 0;
    const size_t pcgs_bytes = (ctx->opt_solve_pcg && n > 128 && n <= 256) ? pcgs_image_bytes(B, n) : 0;
    return dm_align_up(pq_bytes) + dm_align_up(img_bytes) + dm_align_up(spill_bytes) + dm_align_up((size_t)B * dm_cdiv(k2, 32) * 4) + dm_align_up(pcgs_bytes) + 4096;
}

// the batched iteration (dm_pcg.h) takes the systems of order 65 .. 128 (one instantiation: eight row tiles on four waves); whole
// workgroups of the direct solver must belong to one pair for its fall-back launch (k2 % 4 == 0)
static inline bool fmap_solve_pcg_ok(const dm_ctx* ctx, int k1, int k2) {
    const int n = k1 - 1;
    return ctx->opt_solve_pcg && ctx->opt_solve_reg && !ctx->opt_solve_packed && n >= 65 && n <= 128 && k2 % 4 == 0;
}

// Gram matrices + the k2 solves per pair; OPA = the stacked rows [A; Bm], OPB = the rows of A (fp32 arrays, or the split-K
// partials of the projections: dm_gemm_f64.h).  Workspace from what the caller reserved (fmap_solve_ws).
template <class OPA, class OPB>
static int fmap_solve_core(dm_ctx* ctx, int B, int k1, int k2, int D, const OPA& opa, const OPB& opb,
                           const double* lam1, const double* lam2, const double* c00, double w_descr, double w_lap,
                           double* C, int32_t* info) {
    const size_t pq_bytes = (size_t)B * (k1 + k2) * k1 * 8;
    const int n = k1 - 1;
    const int NB = (n + 15) / 16;
    const bool two_phase = (NB == 12 || NB == 13) && !ctx->opt_solve_packed;   // 177 <= n <= 208
    const bool blocked = ((n >= 1 && NB <= 11) && !ctx->opt_solve_packed) || two_phase;   // dm_set_option("solve_packed", 1): tests
    const size_t img_bytes = blocked ? (size_t)B * (NB * (NB + 1) / 2) * 256 * 8 : 0;
    const int NA = (NB + 1) / 2;
    const int grid2 = ctx->n_cu > 0 ? ctx->n_cu : 256;
    const size_t spill_bytes = two_phase ? (size_t)grid2 * (NA * (NA + 1) / 2) * 256 * 8 : 0;
    int rc = DM_OK;
    double* PQ = (double*)dm_ws_take(ctx, pq_bytes);
    double* Timg = blocked ? (double*)dm_ws_take(ctx, img_bytes) : nullptr;
    double* spill = two_phase ? (double*)dm_ws_take(ctx, spill_bytes) : nullptr;
    if (!PQ || (blocked && !Timg) || (two_phase && !spill)) return dm_fail(ctx, DM_ENOMEM, "fmap_solve: workspace not reserved");
    if (blocked) DM_CHECK_HIP(ctx, hipMemsetAsync(Timg, 0, img_bytes, ctx->stream));
    // The solver is chosen by the SIZES alone, never by the batch: a pair's bits do not depend on the batch it is in.  (Measured and not
    // adopted: the iteration costs ~125 us at n = 127 / ~330 us at n = 199 whatever the batch, the direct solvers 52 / 98 us for ONE pair --
    // they win below 16-24 pairs of k = 128 and below 4 pairs of k = 200, profiles/r06_solve_pcg_crossover.txt -- but a switch on B k2
    // makes the map of a pair solved alone differ by 5e-12 from the same pair in a batch of 64.)
    const bool pcg_small = blocked && !two_phase && NB <= 8 && ctx->opt_solve_reg && fmap_solve_pcg_ok(ctx, k1, k2);
    const bool pcg_big = (two_phase || (blocked && NB >= 9)) && ctx->opt_solve_pcg && !ctx->opt_solve_packed;
    // (with the batched iteration in front, its group-0 workgroups clear the status words: one memset launch less per call)
    if (!pcg_small && !pcg_big) DM_CHECK_HIP(ctx, hipMemsetAsync(info, 0, (size_t)B * sizeof(int32_t), ctx->stream));

    OutScaled out{PQ, (long long)(k1 + k2) * k1, k1, w_descr, Timg, (long long)(NB * (NB + 1) / 2) * 256, k1};
    dim3 grid(dm_cdiv(k1 + k2, NT_T) * dm_cdiv(k1, NT_T), 1, B);
    switch (dm_knob("DM_GRAM_NPRE", 2)) {           // experiments: stages of operand loads in flight (96 -> 71 us with the uniform fast path of the operand functors; 1 / 2 / 3 / 4 stages: 73.5 / 71.4 / 73.6 / 75.1)
        case 1: DM_LAUNCH(ctx, "gram_nt_f64", (gemm_nt_f64<OPA, OPB, OutScaled, 1>), grid, dim3(256), 0, opa, opb, out, k1 + k2, k1, D); break;
        case 4: DM_LAUNCH(ctx, "gram_nt_f64", (gemm_nt_f64<OPA, OPB, OutScaled, 4>), grid, dim3(256), 0, opa, opb, out, k1 + k2, k1, D); break;
        case 3: DM_LAUNCH(ctx, "gram_nt_f64", (gemm_nt_f64<OPA, OPB, OutScaled, 3>), grid, dim3(256), 0, opa, opb, out, k1 + k2, k1, D); break;
        default: DM_LAUNCH(ctx, "gram_nt_f64", (gemm_nt_f64<OPA, OPB, OutScaled, 2, true>), dim3(grid.x * (unsigned)B), dim3(256), 0, opa, opb, out, k1 + k2, k1, D); break;
    }

    const int32_t* only_if_big = nullptr;
    const int only_ng = dm_cdiv(k2, PCG_NS);
    if (pcg_big) {
        // r06: systems of order 129 .. 208 by the batched iteration with STREAMED matrix fragments (dm_pcg.h: fmap_solve_pcgs_kernel); the
        // LDS solvers below then only run the pairs it flagged
        int32_t* fb = (int32_t*)dm_ws_take(ctx, (size_t)B * only_ng * 4);
        double* img = (double*)dm_ws_take(ctx, pcgs_image_bytes(B, n));
        if (!fb || !img) return dm_fail(ctx, DM_ENOMEM, "fmap_solve: workspace not reserved");
        const int NTp = (n + 15) / 16, KSP = pcgs_ksp(n), ngroups = dm_cdiv(k2, PCG_NS);
        DM_LAUNCH(ctx, "fmap_solve_pcg_pack", pcgs_pack_kernel, dim3(64, B), dim3(256), 0, (const double*)PQ, k1, k2, NTp, KSP, img);
        const size_t lds_pcg = pcg_lds_bytes(PCGS_NT, PCGS_NW);
        rc = dm_grant_lds(ctx, (const void*)fmap_solve_pcgs_kernel, lds_pcg);
        if (rc) return rc;
        DM_LAUNCH(ctx, "fmap_solve_pcg", fmap_solve_pcgs_kernel, dim3((unsigned)(B * ngroups)), dim3(64 * PCGS_NW), lds_pcg, (const double*)PQ,
                  (const double*)img, lam1, lam2, c00, w_lap, k1, k2, ngroups, NTp, KSP, 1e-22, 48, 6, 3e-4, C, fb, info);
        only_if_big = fb;
    }
    if (two_phase) {
        const int NBr = NB - NA;
        const size_t lds = ((size_t)(NA * (NA + 1) / 2 + NBr * NA + 2) * 256 + 2 * NB * 16 + 8) * sizeof(double) + (96 + 16 + 128) * sizeof(int);
        rc = dm_grant_lds(ctx, (const void*)fmap_solve_2phase_kernel, lds);
        if (rc) return rc;
        const long long nsys = (long long)B * k2;
        DM_LAUNCH(ctx, "fmap_solve_chol", fmap_solve_2phase_kernel, dim3((unsigned)(nsys < grid2 ? nsys : grid2)), dim3(256), lds, PQ, Timg,
                  lam1, lam2, c00, w_lap, k1, k2, NB, B, spill, C, info, only_if_big, only_ng);
        return DM_OK;
    }
    if (blocked && !two_phase && NB <= 8 && ctx->opt_solve_reg) {
        // register-resident solver: one wave per system, four systems per CU (dm_chol_reg.h)
        const long long nsys = (long long)B * k2;
        const dim3 grid((unsigned)((nsys + 3) / 4));
        const int32_t* only_if = nullptr;
        if (pcg_small) {
            // r06: the batched Jacobi-preconditioned conjugate-gradient iteration first (dm_pcg.h); the direct solver below then only
            // runs the pairs whose iteration did not converge or met a non-positive curvature (its workgroups of the other pairs leave
            // at once)
            int32_t* fb = (int32_t*)dm_ws_take(ctx, (size_t)B * only_ng * 4);
            if (!fb) return dm_fail(ctx, DM_ENOMEM, "fmap_solve: workspace not reserved");
            const int ngroups = dm_cdiv(k2, PCG_NS);
            const size_t lds_pcg = pcg_lds_bytes(8, 4);
            rc = dm_grant_lds(ctx, (const void*)fmap_solve_pcg_kernel<8, 4>, lds_pcg);
            if (rc) return rc;
            // stop at a relative reduction of 1e-22 of r^T M^-1 r (1e-11 in that norm: the iterate is within ~1e-10 of the direct
            // solution at cond = 1e2; the bar on C is 1e-4, the closed form's own distance to the float64 minimiser 6e-8); at most 48
            // steps, and a pair whose reduction is still above 3e-4 after 6 goes to the direct solver at once (measured on three
            // descriptor families x three weightings, profiles/r06_solver_jacobi_pcg_experiment.txt: <= 4.7e-5 after six steps where
            // the iteration finishes in 20 - 29, >= 2.4e-3 for rank-deficient descriptors, which would need more than 48)
            DM_LAUNCH(ctx, "fmap_solve_pcg", (fmap_solve_pcg_kernel<8, 4>), dim3((unsigned)(B * ngroups)), dim3(256), lds_pcg, PQ, lam1, lam2, c00,
                      w_lap, k1, k2, ngroups, 1e-22, 48, 6, 3e-4, C, fb, info);
            only_if = fb;
        }
#define DM_SOLVE_REG(NBT_)                                                                                             \
        {                                                                                                              \
            rc = dm_grant_lds(ctx, (const void*)fmap_solve_reg_kernel<NBT_>, solve_reg_lds(NBT_));                     \
            if (rc) return rc;                                                                                         \
            DM_LAUNCH(ctx, "fmap_solve_chol", fmap_solve_reg_kernel<NBT_>, grid, dim3(256), solve_reg_lds(NBT_), PQ, Timg, lam1, lam2, \
                      c00, w_lap, k1, k2, NB, nsys, C, info, only_if, only_ng);                                        \
        }
        if (NB <= 2) DM_SOLVE_REG(2)
        else if (NB <= 4) DM_SOLVE_REG(4)
        else if (NB <= 6) DM_SOLVE_REG(6)
        else DM_SOLVE_REG(8)
#undef DM_SOLVE_REG
        return DM_OK;
    }
    if (blocked) {
        // blocked MFMA solver: NB(NB+1)/2 + 2 blocks of 2 KiB, vectors
        const size_t lds = ((size_t)(NB * (NB + 1) / 2 + 2) * 256 + 2 * NB * 16 + 16 + 8) * sizeof(double);
        rc = dm_grant_lds(ctx, (const void*)fmap_solve_blocked_kernel, lds);
        if (rc) return rc;
        DM_LAUNCH(ctx, "fmap_solve_chol", fmap_solve_blocked_kernel, dim3(k2, B), dim3(256), lds, PQ, Timg, lam1, lam2, c00,
                  w_lap, k1, k2, NB, C, info, only_if_big, only_ng);
        return DM_OK;
    }
    const size_t lds = ((size_t)(n + 1) * (n + 2) / 2 + (n + 1 < 16 ? 16 : n + 1) + 4) * sizeof(double);
    rc = dm_grant_lds(ctx, (const void*)fmap_solve_kernel, lds);
    if (rc) return rc;
    // DM_EXPERIMENTS builds only: 1 no trailing update, 2 no back substitution, 3 no factorisation (wrong results)
    DM_LAUNCH(ctx, "fmap_solve_chol", fmap_solve_kernel, dim3(k2, B), dim3(SP_NT), lds, PQ, lam1, lam2, c00, w_lap, k1, k2,
              C, info, dm_knob("DM_SOLVE_DEBUG", 0));
    return DM_OK;
}

extern "C" int dm_fmap_solve(dm_ctx* ctx, int B, int k1, int k2, int D, const float* A, const float* Bm,
                             const double* lam1, const double* lam2, const double* c00, double w_descr, double w_lap,
                             double* C, int32_t* info) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && k1 > 0 && k2 > 0 && D > 0, "sizes must be positive");
    DM_REQUIRE(ctx, A && Bm && lam1 && lam2 && c00 && C && info, "null pointer");
    DM_REQUIRE(ctx, k1 <= 200, "k1 > 200 does not fit the in-LDS solver");
    DM_REQUIRE(ctx, w_descr >= 0.0 && w_lap >= 0.0 && (w_descr > 0.0 || w_lap > 0.0), "weights must be >= 0 and not both 0");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = dm_ws_reserve(ctx, fmap_solve_ws(ctx, B, k1, k2));
    if (rc) return rc;
    KRowsStackedF32 opa{A, Bm, k1, k2, D};
    KRowsF32 opb{A, (long long)k1 * D, D, k1, D};
    return fmap_solve_core(ctx, B, k1, k2, D, opa, opb, lam1, lam2, c00, w_descr, w_lap, C, info);
}

// =================================================================================================
// dm_fmap_fit: FunctionalMapping.fit with the two quadratic terms in ONE call (projections, pinned column, Gram, solves)
// =================================================================================================
template <typename TR>
static int fmap_fit_impl(dm_ctx* ctx, int B, int N1, int N2, int D, int k1, int k2, const TR* Phi1, int ld1, const TR* Phi2, int ld2,
                         const TR* mass1, const TR* mass2, const void* F1, const void* F2, const double* lam1, const double* lam2,
                         double w_descr, double w_lap, double* C, int32_t* info) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N1 > 0 && N2 > 0 && D > 0 && k1 > 0 && k2 > 0, "sizes must be positive");
    DM_REQUIRE(ctx, Phi1 && Phi2 && mass1 && mass2 && F1 && F2 && lam1 && lam2 && C && info, "null pointer");
    DM_REQUIRE(ctx, ld1 >= k1 && ld2 >= k2, "eigenvector row stride smaller than k");
    DM_REQUIRE(ctx, k1 <= 200, "k1 > 200 does not fit the in-LDS solver");
    DM_REQUIRE(ctx, w_descr >= 0.0 && w_lap >= 0.0 && (w_descr > 0.0 || w_lap > 0.0), "weights must be >= 0 and not both 0");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bA = (size_t)B * k1 * D * 4, bB = (size_t)B * k2 * D * 4;
    int rc = dm_ws_reserve(ctx, dm_project_f16split_ws(B, N1, D, k1, ld1, (int)sizeof(TR)) + dm_project_f16split_ws(B, N2, D, k2, ld2, (int)sizeof(TR)) +
                                    dm_align_up(bA) + dm_align_up(bB) + dm_align_up((size_t)B * 8) + fmap_solve_ws(ctx, B, k1, k2) + 8192);
    if (rc) return rc;
    double* c00 = (double*)dm_ws_take(ctx, (size_t)B * 8);
    if (!c00) return dm_fail(ctx, DM_ENOMEM, "fmap_fit: workspace not reserved");
    // the projections leave their split-K partials where they are when there are at most two of them: the Gram kernel adds them
    // up as it reads (same values as the reduce kernel would store); more chunks are reduced first
    const bool lazy = dm_cdiv(N1, 1024) <= 2 && dm_cdiv(N2, 1024) <= 2;
    float* A = lazy ? nullptr : (float*)dm_ws_take(ctx, bA);
    float* Bm = lazy ? nullptr : (float*)dm_ws_take(ctx, bB);
    if (!lazy && (!A || !Bm)) return dm_fail(ctx, DM_ENOMEM, "fmap_fit: workspace not reserved");
    const float* pA = nullptr; const float* pB = nullptr;
    int nsA = 1, nsB = 1;
    rc = dm_project_f16split_launch<TR>(ctx, B, N1, D, k1, Phi1, ld1, mass1, F1, A, &pA, &nsA);
    if (rc) return rc;
    // (the pinned column's entry comes out of the second projection's maxima pass: one more workgroup per pair, no launch)
    const dm_c00_args<TR> cz{Phi1, (long long)N1 * ld1, Phi2, (long long)N2 * ld2, mass1, mass2, N1, N2, c00};
    rc = dm_project_f16split_launch<TR>(ctx, B, N2, D, k2, Phi2, ld2, mass2, F2, Bm, &pB, &nsB, &cz);
    if (rc) return rc;
    if (lazy) {
        const long long sA = nsA == 2 ? (long long)B * k1 * D : 0, sB = nsB == 2 ? (long long)B * k2 * D : 0;
        KRowsStackedPart opa{pA, pB, sA, sB, k1, k2, D};
        KRowsPart opb{pA, sA, k1, D};
        return fmap_solve_core(ctx, B, k1, k2, D, opa, opb, lam1, lam2, c00, w_descr, w_lap, C, info);
    }
    KRowsStackedF32 opa{A, Bm, k1, k2, D};
    KRowsF32 opb{A, (long long)k1 * D, D, k1, D};
    return fmap_solve_core(ctx, B, k1, k2, D, opa, opb, lam1, lam2, c00, w_descr, w_lap, C, info);
}
extern "C" int dm_fmap_fit(dm_ctx* ctx, int B, int N1, int N2, int D, int k1, int k2, const float* Phi1, int ld1, const float* Phi2,
                           int ld2, const float* mass1, const float* mass2, const void* F1, const void* F2, const double* lam1,
                           const double* lam2, double w_descr, double w_lap, double* C, int32_t* info) {
    return fmap_fit_impl<float>(ctx, B, N1, N2, D, k1, k2, Phi1, ld1, Phi2, ld2, mass1, mass2, F1, F2, lam1, lam2, w_descr, w_lap, C, info);
}
extern "C" int dm_fmap_fit_f64(dm_ctx* ctx, int B, int N1, int N2, int D, int k1, int k2, const double* Phi1, int ld1,
                               const double* Phi2, int ld2, const double* mass1, const double* mass2, const void* F1, const void* F2,
                               const double* lam1, const double* lam2, double w_descr, double w_lap, double* C, int32_t* info) {
    return fmap_fit_impl<double>(ctx, B, N1, N2, D, k1, k2, Phi1, ld1, Phi2, ld2, mass1, mass2, F1, F2, lam1, lam2, w_descr, w_lap, C, info);
}

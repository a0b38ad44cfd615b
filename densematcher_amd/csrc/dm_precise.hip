// Precise (barycentric) vertex-to-surface map on the device (dm_precise_map): FunctionalMapping.get_precise_map.
//
// Reference arithmetic reproduced (oracle/dm_oracle.py: project_pc_to_triangles, precise_map_dense;
// pyFM/functional.py:221-251 -> pyFM/spectral/convert.py:185-229 (use_adj = True) -> pyFM/spectral/projection_utils.py):
//   emb1 = Phi1[:, :k1] (vertices of mesh 1 in the spectral embedding), emb2 = Phi2[:, :k2] C (points to project)
//   lmax_f   = longest edge of face f in the embedding                                     projection_utils.py:118-142
//   Deltamin = distance of the point to its nearest vertex                                 :145-178
//   dmin_f   = distance of the point to the nearest of the face's three vertices           :282-327 (mycdist :181-229)
//   candidates: faces with dmin_f - lmax_f < Deltamin                                      :356
//   projection on every candidate (Eberly's seven regions; the vectorised code's region-4 quirk is kept: it decides
//   which face wins), the closest projection wins, first face index on equal distances     :330-367, :369-998
// One workgroup per point: the point's row of vertex distances sits in LDS, the faces are scanned 256 at a time, the
// candidates (a few dozen) are projected one per wave with the k1-long dot products spread over the lanes.
#include "dm_gemm_f64.h"
#include "dm_internal.h"

static inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

struct OutNTd {
    double* p; long long stride_b; int ld;
    __device__ __forceinline__ void store(int b, int i, int j, double v) const { p[b * stride_b + (long long)i * ld + j] = v; }
};
// dist[i][v] = sqrt(max((-2 xy + |e_v|^2) + |p_i|^2, 0))     (mycdist's operation order)
struct OutDist {
    double* p; long long stride_b; int ld; const double* vs; int N1; const double* ps; int N2;
    __device__ __forceinline__ void store(int b, int i, int j, double xy) const {
        const double d2 = (-2.0 * xy + vs[(long long)b * N1 + j]) + ps[(long long)b * N2 + i];
        p[b * stride_b + (long long)i * ld + j] = sqrt(fmax(d2, 0.0));
    }
};

// E1[b][v][:] = (double) Phi1[b][v][:k1];  one thread per element
template <typename TR>
__global__ __launch_bounds__(256) void widen_rows_kernel(const TR* __restrict__ Phi, int N, int ld, int k, double* __restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (e >= (long long)N * k) return;
    const int v = (int)(e / k), c = (int)(e - (long long)v * k);
    out[(long long)b * N * k + e] = (double)Phi[((long long)b * N + v) * ld + c];
}
// sq[b][r] = (sqrt(sum_c X[b][r][c]^2))^2     (np.linalg.norm(...)**2: the rounding of the square root is part of it)
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const double* __restrict__ X, int N, int k, double* __restrict__ sq) {
    const int b = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    const double* row = X + ((long long)b * N + r) * k;
    double s = 0.0;
    for (int c = 0; c < k; ++c) s += row[c] * row[c];
    const double n = sqrt(s);
    sq[(long long)b * N + r] = n * n;
}
// per face: a = |e1 - e0|^2, b = <e1 - e0, e2 - e0>, c = |e2 - e0|^2, lmax = longest edge     fc (B, nf, 4)
__global__ __launch_bounds__(256) void face_consts_kernel(const double* __restrict__ E1, int N1, int k, const int32_t* __restrict__ faces,
                                                          int nf, double* __restrict__ fc) {
    const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
    if (f >= nf) return;
    const int32_t* fv = faces + ((long long)b * nf + f) * 3;
    const double* p0 = E1 + ((long long)b * N1 + fv[0]) * k;
    const double* p1 = E1 + ((long long)b * N1 + fv[1]) * k;
    const double* p2 = E1 + ((long long)b * N1 + fv[2]) * k;
    double a = 0.0, bb = 0.0, c = 0.0, l12 = 0.0;
    for (int q = 0; q < k; ++q) {
        const double x1 = p1[q] - p0[q], x2 = p2[q] - p0[q], x12 = p2[q] - p1[q];
        a += x1 * x1; bb += x1 * x2; c += x2 * x2; l12 += x12 * x12;
    }
    double* o = fc + ((long long)b * nf + f) * 4;
    o[0] = a; o[1] = bb; o[2] = c;
    o[3] = fmax(fmax(sqrt(a), sqrt(l12)), sqrt(c));
}

// Eberly's regions as in projection_utils.py (oracle/dm_oracle.py: _point_triangle); multi = several candidate faces
__device__ void point_triangle(double a, double b, double c, double d, double e, double f, bool multi, double& so, double& to, double& sq) {
    const double det = a * c - b * b;
    const double s = b * e - c * d;
    const double t = b * d - a * e;
#define PT_RET(S_, T_, Q_) { so = (S_); to = (T_); sq = (Q_); return; }
#define PT_INSIDE(S_, T_) { const double si = (S_), ti = (T_); PT_RET(si, ti, si * (a * si + b * ti + 2.0 * d) + ti * (b * si + c * ti + 2.0 * e) + f) }
    if (s + t <= det) {
        if (s < 0.0) {
            if (t < 0.0) {                                            // region 4
                if (d < 0.0) {
                    if (-d >= a) PT_RET(1.0, 0.0, a + 2.0 * d + f)
                    const double s_ = -d / a;
                    PT_RET(s_, 0.0, multi ? d * s + f : d * s_ + f)
                }
                if (e >= 0.0) PT_RET(0.0, 0.0, f)
                if (-e >= c) PT_RET(0.0, 1.0, c + 2.0 * e + f)
                const double t_ = -e / c;
                PT_RET(0.0, t_, multi ? e * t + f : e * t_ + f)
            }
            if (e >= 0.0) PT_RET(0.0, 0.0, f)                          // region 3
            if (-e >= c) PT_RET(0.0, 1.0, c + 2.0 * e + f)
            const double t_ = -e / c;
            PT_RET(0.0, t_, e * t_ + f)
        }
        if (t < 0.0) {                                                // region 5
            if (d >= 0.0) PT_RET(0.0, 0.0, f)
            if (-d >= a) PT_RET(1.0, 0.0, a + 2.0 * d + f)
            const double s_ = -d / a;
            PT_RET(s_, 0.0, d * s_ + f)
        }
        const double inv = 1.0 / det;                                 // region 0
        PT_INSIDE(s * inv, t * inv)
    }
    if (s < 0.0) {                                                    // region 2
        const double tmp0 = b + d, tmp1 = c + e;
        if (tmp1 > tmp0) {
            const double numer = tmp1 - tmp0, denom = a - 2.0 * b + c;
            if (numer >= denom) PT_RET(1.0, 0.0, a + 2.0 * d + f)
            const double s_ = numer / denom;
            PT_INSIDE(s_, 1.0 - s_)
        }
        if (tmp1 <= 0.0) PT_RET(0.0, 1.0, c + 2.0 * e + f)
        if (e >= 0.0) PT_RET(0.0, 0.0, f)
        const double t_ = -e / c;
        PT_RET(0.0, t_, e * t_ + f)
    }
    if (t < 0.0) {                                                    // region 6
        const double tmp0 = b + e, tmp1 = a + d;
        if (tmp1 > tmp0) {
            const double numer = tmp1 - tmp0, denom = a - 2.0 * b + c;
            if (numer >= denom) PT_RET(0.0, 1.0, c + 2.0 * e + f)
            const double t_ = numer / denom;
            PT_INSIDE(1.0 - t_, t_)
        }
        if (tmp1 <= 0.0) PT_RET(1.0, 0.0, a + 2.0 * d + f)
        if (d >= 0.0) PT_RET(0.0, 0.0, f)
        const double s_ = -d / a;
        PT_RET(s_, 0.0, d * s_ + f)
    }
    const double numer = c + e - b - d;                               // region 1
    if (numer <= 0.0) PT_RET(0.0, 1.0, c + 2.0 * e + f)
    const double denom = a - 2.0 * b + c;
    if (numer >= denom) PT_RET(1.0, 0.0, a + 2.0 * d + f)
    const double s_ = numer / denom;
    PT_INSIDE(s_, 1.0 - s_)
#undef PT_INSIDE
#undef PT_RET
}

constexpr int PM_MAXCAND = 4096;          // candidate faces LISTED per point (a point typically has a few dozen); a point with more
                                          // (far from the surface: a poor C) re-tests every face instead: nothing is dropped

__global__ __launch_bounds__(256) void precise_project_kernel(const double* __restrict__ dist, const double* __restrict__ E1,
                                                              const double* __restrict__ E2, const int32_t* __restrict__ faces,
                                                              const double* __restrict__ fc, int N1, int N2, int k, int nf,
                                                              int32_t* __restrict__ face_match, double* __restrict__ bary,
                                                              double* __restrict__ dense, int32_t* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) double pm_smem[];        // dist row (N1) | point (k) | candidate list
    double* drow = pm_smem;
    double* pt = drow + N1;
    int* cand = reinterpret_cast<int*>(pt + k);
    __shared__ double w_dist[4], w_s[4], w_t[4];
    __shared__ int w_face[4];
    __shared__ int s_ncand, s_nn;
    __shared__ double s_delta;
    const int i = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const double* dr = dist + ((long long)b * N2 + i) * N1;
    for (int v = t; v < N1; v += 256) drow[v] = dr[v];
    for (int q = t; q < k; q += 256) pt[q] = E2[((long long)b * N2 + i) * k + q];
    if (t == 0) s_ncand = 0;
    __syncthreads();
    // nearest vertex (lowest index on equal distances), then its distance computed directly (norm of the difference)
    {
        double bv = DM_INF_F64; int bj = DM_IDX_NONE;
        for (int v = t; v < N1; v += 256) { const double x = drow[v]; if (x < bv) { bv = x; bj = v; } }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off); const int oj = __shfl_xor(bj, off);
            argmin_merge(bv, bj, ov, oj);
        }
        if (lane == 0) { w_dist[wave] = bv; w_face[wave] = bj; }
        __syncthreads();
        if (t == 0) {
            double v0 = w_dist[0]; int j0 = w_face[0];
            for (int w = 1; w < 4; ++w) argmin_merge(v0, j0, w_dist[w], w_face[w]);
            s_nn = j0;
        }
        __syncthreads();
        if (wave == 0) {
            const double* ev = E1 + ((long long)b * N1 + s_nn) * k;
            double acc = 0.0;
            for (int q = lane; q < k; q += 64) { const double x = ev[q] - pt[q]; acc += x * x; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            if (lane == 0) s_delta = sqrt(acc);
        }
        __syncthreads();
    }
    const double Deltamin = s_delta;
    // candidate faces
    for (int f0 = 0; f0 < nf; f0 += 256) {
        const int f = f0 + t;
        if (f < nf) {
            const int32_t* fv = faces + ((long long)b * nf + f) * 3;
            const double dmin = fmin(fmin(drow[fv[0]], drow[fv[1]]), drow[fv[2]]);
            if (dmin - fc[((long long)b * nf + f) * 4 + 3] < Deltamin) {
                const int pos = atomicAdd(&s_ncand, 1);
                if (pos < PM_MAXCAND) cand[pos] = f;
            }
        }
    }
    __syncthreads();
    const int ncand = s_ncand;
    const bool listed = ncand <= PM_MAXCAND;          // else: every face is visited and the candidate test repeated (uniform per wave)
    if (!listed && t == 0) overflow[b] = 1;           // informational: the slow route was taken for some point of this pair
    const bool multi = ncand > 1;
    // project on the candidates: G lanes per candidate (16 for maps up to 16 columns, 32 up to 32, else the whole wave), 64 / G
    // candidates per wave at a time -- every candidate is a chain of dependent gathers (face -> three embedding rows), and with k = 15
    // a wave per candidate left 49 lanes idle through it: 46 ms for 64 pairs (r05).  The sums are the same additions as before: a lane
    // holds the same elements, the butterfly over G lanes is the tail of the one over 64 (whose upper levels added zeros).
    // (The result is the lexicographic minimum of (distance, face index): independent of the order the list was filled in.)
    const int G = k <= 16 ? 16 : (k <= 32 ? 32 : 64), per_wave = 64 / G, sl = lane & (G - 1), grp = lane / G;
    double bd = DM_INF_F64, bs = 0.0, bt = 0.0;
    int bf = DM_IDX_NONE;
    const int nloop = listed ? ncand : nf;
    for (int q0 = wave * per_wave; q0 < nloop; q0 += 4 * per_wave) {
        const int q = q0 + grp;
        bool live = q < nloop;
        int f = 0;
        if (live) {
            f = listed ? cand[q] : q;
            if (!listed) {
                const int32_t* fw = faces + ((long long)b * nf + f) * 3;
                const double dmin = fmin(fmin(drow[fw[0]], drow[fw[1]]), drow[fw[2]]);
                if (!(dmin - fc[((long long)b * nf + f) * 4 + 3] < Deltamin)) live = false;
            }
        }
        double d = 0.0, e = 0.0, ff = 0.0;
        if (live) {
            const int32_t* fv = faces + ((long long)b * nf + f) * 3;
            const double* p0 = E1 + ((long long)b * N1 + fv[0]) * k;
            const double* p1 = E1 + ((long long)b * N1 + fv[1]) * k;
            const double* p2 = E1 + ((long long)b * N1 + fv[2]) * k;
            for (int c = sl; c < k; c += G) {
                const double base = p0[c], diff = base - pt[c];
                d += (p1[c] - base) * diff; e += (p2[c] - base) * diff; ff += diff * diff;
            }
        }
        for (int off = G >> 1; off > 0; off >>= 1) { d += __shfl_xor(d, off); e += __shfl_xor(e, off); ff += __shfl_xor(ff, off); }
        if (live) {
            const double* cst = fc + ((long long)b * nf + f) * 4;
            double s_, t_, sq;
            point_triangle(cst[0], cst[1], cst[2], d, e, ff, multi, s_, t_, sq);
            const double dd = sqrt(fmax(sq, 0.0));
            if (dd < bd || (dd == bd && f < bf)) { bd = dd; bf = f; bs = s_; bt = t_; }
        }
    }
    for (int off = G; off < 64; off <<= 1) {                  // the wave's groups: lexicographic minimum
        const double od = __shfl_xor(bd, off), os = __shfl_xor(bs, off), ot = __shfl_xor(bt, off);
        const int of = __shfl_xor(bf, off);
        if (od < bd || (od == bd && of < bf)) { bd = od; bf = of; bs = os; bt = ot; }
    }
    if (lane == 0) { w_dist[wave] = bd; w_face[wave] = bf; w_s[wave] = bs; w_t[wave] = bt; }
    __syncthreads();
    if (t == 0) {
        double d0 = w_dist[0], s0 = w_s[0], t0 = w_t[0];
        int f0 = w_face[0];
        for (int w = 1; w < 4; ++w)
            if (w_dist[w] < d0 || (w_dist[w] == d0 && w_face[w] < f0)) { d0 = w_dist[w]; f0 = w_face[w]; s0 = w_s[w]; t0 = w_t[w]; }
        if (f0 == DM_IDX_NONE) { f0 = 0; s0 = 0.0; t0 = 0.0; }
        const long long o = (long long)b * N2 + i;
        face_match[o] = f0;
        const double b0 = 1.0 - s0 - t0;
        bary[o * 3] = b0; bary[o * 3 + 1] = s0; bary[o * 3 + 2] = t0;
        if (dense) {                                              // csr_matrix((Sn, (In, Jn))): repeated vertices add up
            const int32_t* fv = faces + ((long long)b * nf + f0) * 3;
            double* row = dense + o * N1;
            row[fv[0]] += b0; row[fv[1]] += s0; row[fv[2]] += t0;
        }
    }
}

template <typename TR>
static int precise_map_impl(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, int nf, const TR* Phi1, int ld1,
                            const TR* Phi2, int ld2, const double* C, const int32_t* faces1, int32_t* face_match,
                            double* bary, double* dense, int32_t* info) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N1 > 0 && N2 > 0 && k1 > 0 && k2 > 0 && nf > 0, "sizes must be positive");
    DM_REQUIRE(ctx, Phi1 && Phi2 && C && faces1 && face_match && bary && info, "null pointer");
    DM_REQUIRE(ctx, ld1 >= k1 && ld2 >= k2, "eigenvector row stride smaller than the map size");
    const size_t lds = ((size_t)N1 + k1) * 8 + (size_t)PM_MAXCAND * 4;
    DM_REQUIRE(ctx, lds <= 160 * 1024 - 1024, "mesh 1 has too many vertices for the in-LDS distance row");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bE1 = (size_t)B * N1 * k1 * 8, bE2 = (size_t)B * N2 * k1 * 8, bD = (size_t)B * N2 * N1 * 8;
    int rc = dm_ws_reserve(ctx, dm_align_up(bE1) + dm_align_up(bE2) + dm_align_up(bD) + dm_align_up((size_t)B * N1 * 8) +
                                    dm_align_up((size_t)B * N2 * 8) + dm_align_up((size_t)B * nf * 4 * 8) + 4096);
    if (rc) return rc;
    double* E1 = (double*)dm_ws_take(ctx, bE1);
    double* E2 = (double*)dm_ws_take(ctx, bE2);
    double* D = (double*)dm_ws_take(ctx, bD);
    double* vs = (double*)dm_ws_take(ctx, (size_t)B * N1 * 8);
    double* ps = (double*)dm_ws_take(ctx, (size_t)B * N2 * 8);
    double* fc = (double*)dm_ws_take(ctx, (size_t)B * nf * 4 * 8);
    int32_t* overflow = info;
    if (!E1 || !E2 || !D || !vs || !ps || !fc) return dm_fail(ctx, DM_ENOMEM, "precise map: workspace not reserved");
    DM_CHECK_HIP(ctx, hipMemsetAsync(overflow, 0, (size_t)B * 4, ctx->stream));
    DM_LAUNCH(ctx, "precise_widen", widen_rows_kernel<TR>, dim3((unsigned)(((long long)N1 * k1 + 255) / 256), B), dim3(256), 0, Phi1, N1, ld1, k1, E1);
    {   // emb2 = Phi2 C   (convert.py:220, use_adj)
        KRows<TR> opa{Phi2, (long long)N2 * ld2, ld2, N2, k2};
        KRowsF64 opb{C, (long long)k2 * k1, k1, k1, k2, 1};
        OutNTd out{E2, (long long)N2 * k1, k1};
        DM_LAUNCH(ctx, "emb2_nt_f64", (gemm_nt_f64<KRows<TR>, KRowsF64, OutNTd>), dim3(dm_cdiv(N2, NT_T) * dm_cdiv(k1, NT_T), 1, B),
                  dim3(256), 0, opa, opb, out, N2, k1, k2);
    }
    DM_LAUNCH(ctx, "precise_sqnorm", row_sqnorm_kernel, dim3(dm_cdiv(N1, 256), B), dim3(256), 0, E1, N1, k1, vs);
    DM_LAUNCH(ctx, "precise_sqnorm", row_sqnorm_kernel, dim3(dm_cdiv(N2, 256), B), dim3(256), 0, E2, N2, k1, ps);
    DM_LAUNCH(ctx, "precise_face_consts", face_consts_kernel, dim3(dm_cdiv(nf, 256), B), dim3(256), 0, E1, N1, k1, faces1, nf, fc);
    {   // distances point -> vertex
        KRowsF64 pa{E2, (long long)N2 * k1, k1, N2, k1, 0};
        KRowsF64 vb{E1, (long long)N1 * k1, k1, N1, k1, 0};
        OutDist out{D, (long long)N2 * N1, N1, vs, N1, ps, N2};
        DM_LAUNCH(ctx, "precise_dist_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutDist>), dim3(dm_cdiv(N2, NT_T) * dm_cdiv(N1, NT_T), 1, B),
                  dim3(256), 0, pa, vb, out, N2, N1, k1);
    }
    if (dense) DM_CHECK_HIP(ctx, hipMemsetAsync(dense, 0, (size_t)B * N2 * N1 * 8, ctx->stream));
    rc = dm_grant_lds(ctx, (const void*)precise_project_kernel, lds);
    if (rc) return rc;
    DM_LAUNCH(ctx, "precise_project", precise_project_kernel, dim3(N2, B), dim3(256), lds, D, E1, E2, faces1, fc, N1, N2, k1, nf,
              face_match, bary, dense, overflow);
    return DM_OK;
}
extern "C" int dm_precise_map(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, int nf, const float* Phi1, int ld1,
                              const float* Phi2, int ld2, const double* C, const int32_t* faces1, int32_t* face_match,
                              double* bary, double* dense, int32_t* info) {
    return precise_map_impl<float>(ctx, B, N1, N2, k1, k2, nf, Phi1, ld1, Phi2, ld2, C, faces1, face_match, bary, dense, info);
}
extern "C" int dm_precise_map_f64(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, int nf, const double* Phi1, int ld1,
                                  const double* Phi2, int ld2, const double* C, const int32_t* faces1, int32_t* face_match,
                                  double* bary, double* dense, int32_t* info) {
    return precise_map_impl<double>(ctx, B, N1, N2, k1, k2, nf, Phi1, ld1, Phi2, ld2, C, faces1, face_match, bary, dense, info);
}

// ---- k nearest neighbours, k > 1 (pyFM/spectral/nn_utils.py:4-38) ----------------------------------------------------
// idx[b][i][r] = index of the r-th nearest row of X[b] to Y[b][i] (ascending distance, lowest index first on equal
// distances), dist = the Euclidean distances.  The (ny x nx) distance matrix is formed on the f64 matrix cores
// (|x|^2 - 2 <x, y> + |y|^2, clipped at 0), one workgroup per query row then extracts the k smallest by k arg-min passes
// over the row held in LDS.  The k = 1 searches of the matching path never take this route (dm_knn_query_f64).
struct OutSqDist {
    double* p; long long stride_b; int ld; const double* xs; int nx; const double* ys; int ny;
    __device__ __forceinline__ void store(int b, int i, int j, double xy) const {
        p[b * stride_b + (long long)i * ld + j] = fmax((xs[(long long)b * nx + j] - 2.0 * xy) + ys[(long long)b * ny + i], 0.0);
    }
};
__global__ __launch_bounds__(256) void row_sumsq_kernel(const double* __restrict__ X, int N, int k, double* __restrict__ sq) {
    const int b = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    const double* row = X + ((long long)b * N + r) * k;
    double s = 0.0;
    for (int c = 0; c < k; ++c) s += row[c] * row[c];
    sq[(long long)b * N + r] = s;
}
// The k nearest are SELECTED on the expanded form |x|^2 - 2 <x, y> + |y|^2 (a matrix product); their distances are then
// re-evaluated directly as |x - y| (what sklearn returns) and the k results put in the order of those exact distances, lowest
// index first on equal ones -- the expanded form is only good to ~1e-16 |x|^2, which can misorder and misreport near-ties.
__global__ __launch_bounds__(256) void topk_rows_kernel(const double* __restrict__ D2, int nx, int ny, int k, int32_t* __restrict__ idx,
                                                        double* __restrict__ dist, const double* __restrict__ X,
                                                        const double* __restrict__ Y, int p) {
    extern __shared__ __attribute__((aligned(16))) double tk_row[];
    __shared__ double w_v[4];
    __shared__ int w_j[4];
    const int i = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const double* dr = D2 + ((long long)b * ny + i) * nx;
    for (int j = t; j < nx; j += 256) tk_row[j] = dr[j];
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        double bv = DM_INF_F64; int bj = DM_IDX_NONE;
        for (int j = t; j < nx; j += 256) { const double x = tk_row[j]; if (x < bv) { bv = x; bj = j; } }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off); const int oj = __shfl_xor(bj, off);
            argmin_merge(bv, bj, ov, oj);
        }
        if (lane == 0) { w_v[wave] = bv; w_j[wave] = bj; }
        __syncthreads();
        if (t == 0) {
            double v0 = w_v[0]; int j0 = w_j[0];
            for (int w = 1; w < 4; ++w) argmin_merge(v0, j0, w_v[w], w_j[w]);
            const long long o = ((long long)b * ny + i) * k + r;
            idx[o] = (j0 == DM_IDX_NONE) ? 0 : j0;
            if (j0 != DM_IDX_NONE) tk_row[j0] = DM_INF_F64;            // taken: out of the next passes
        }
        __syncthreads();
    }
    // exact distances of the k selected rows (one wave per neighbour at a time), kept in tk_row[0 .. k)
    __syncthreads();
    const double* yq = Y + ((long long)b * ny + i) * p;
    int32_t* ir = idx + ((long long)b * ny + i) * k;
    for (int r = wave; r < k; r += 4) {
        const double* xr = X + ((long long)b * nx + ir[r]) * p;
        double acc = 0.0;
        for (int c = lane; c < p; c += 64) { const double dlt = xr[c] - yq[c]; acc = fma(dlt, dlt, acc); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) tk_row[r] = sqrt(acc);
    }
    __syncthreads();
    if (t == 0) {                                                     // insertion sort of the k results by (distance, index)
        for (int a = 1; a < k; ++a) {
            const double dv = tk_row[a]; const int32_t jv = ir[a];
            int q = a - 1;
            while (q >= 0 && (tk_row[q] > dv || (tk_row[q] == dv && ir[q] > jv))) { tk_row[q + 1] = tk_row[q]; ir[q + 1] = ir[q]; --q; }
            tk_row[q + 1] = dv; ir[q + 1] = jv;
        }
        if (dist) for (int a = 0; a < k; ++a) dist[((long long)b * ny + i) * k + a] = tk_row[a];
    }
}
extern "C" int dm_knn_query_topk_f64(dm_ctx* ctx, int B, int nx, int ny, int p, int k, const double* X, const double* Y,
                                     int32_t* idx, double* dist) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && nx > 0 && ny > 0 && p > 0 && k > 0 && k <= nx, "sizes must be positive, k <= nx");
    DM_REQUIRE(ctx, X && Y && idx, "null pointer");
    DM_REQUIRE(ctx, (size_t)nx * 8 <= 150 * 1024, "too many tree points for the in-LDS distance row");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bD = (size_t)B * ny * nx * 8;
    int rc = dm_ws_reserve(ctx, dm_align_up(bD) + dm_align_up((size_t)B * nx * 8) + dm_align_up((size_t)B * ny * 8) + 4096);
    if (rc) return rc;
    double* D2 = (double*)dm_ws_take(ctx, bD);
    double* xs = (double*)dm_ws_take(ctx, (size_t)B * nx * 8);
    double* ys = (double*)dm_ws_take(ctx, (size_t)B * ny * 8);
    if (!D2 || !xs || !ys) return dm_fail(ctx, DM_ENOMEM, "knn top-k: workspace not reserved");
    DM_LAUNCH(ctx, "knn_sumsq", row_sumsq_kernel, dim3(dm_cdiv(nx, 256), B), dim3(256), 0, X, nx, p, xs);
    DM_LAUNCH(ctx, "knn_sumsq", row_sumsq_kernel, dim3(dm_cdiv(ny, 256), B), dim3(256), 0, Y, ny, p, ys);
    KRowsF64 ya{Y, (long long)ny * p, p, ny, p, 0};
    KRowsF64 xb{X, (long long)nx * p, p, nx, p, 0};
    OutSqDist out{D2, (long long)ny * nx, nx, xs, nx, ys, ny};
    DM_LAUNCH(ctx, "knn_dist_nt_f64", (gemm_nt_f64<KRowsF64, KRowsF64, OutSqDist>), dim3(dm_cdiv(ny, NT_T) * dm_cdiv(nx, NT_T), 1, B), dim3(256), 0,
              ya, xb, out, ny, nx, p);
    const size_t lds = (size_t)(nx > k ? nx : k) * 8;
    rc = dm_grant_lds(ctx, (const void*)topk_rows_kernel, lds);
    if (rc) return rc;
    DM_LAUNCH(ctx, "knn_topk", topk_rows_kernel, dim3(ny, B), dim3(256), lds, (const double*)D2, nx, ny, k, idx, dist, X, Y, p);
    return DM_OK;
}

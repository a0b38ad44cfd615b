// Linear assignment on the device (dm_linear_sum_assignment): the Hungarian outputs of compute_surface_map.
//
// Reference call reproduced: scipy.optimize.linear_sum_assignment(mapped_indicator, maximize=True)
// (densematcher/functional_map.py:57,66,78).  The arithmetic is SciPy's (third party, unpinned in the reference's
// setup.py): the shortest-augmenting-path algorithm of D. F. Crouse, IEEE TAES 52(4), 2016.  oracle/dm_oracle.py:
// linear_sum_assignment restates it step for step and is checked against SciPy; this kernel follows the same steps in
// the same floating-point order, so the assignment is IDENTICAL to SciPy's, ties included:
//   rows are augmented in order; each augmentation is a Dijkstra search whose every step scans the `remaining` columns
//   (a list that starts reversed and is compacted by swap-removal), relaxes  r = min_val + cost[i][j] - u[i] - v[j],
//   and picks the smallest tentative cost -- among equals the first in scan order, unless an unassigned column ties, then
//   the last unassigned one; dual variables are updated after each search, then the path is flipped.
//
// One workgroup of 1024 threads per matrix (batch element): the search is sequential in its steps (a step needs the row
// the previous step selected), parallel inside a step (the scan over the remaining columns and the arg-min).  A step
// costs one cost-matrix row from L2 / HBM (latency bound, ~1-2 us); pairs of a batch run concurrently on different CUs.
// The column state (v, tentative costs, path, owners, remaining list) lives in LDS up to 4608 columns, else in global
// scratch.
#include "dm_device.h"
#include "dm_internal.h"

constexpr int LSA_NT = 1024;
constexpr int LSA_LDS_MAX_COLS = 4608;      // 28 bytes per column + reduction scratch within 160 KiB

struct lsa_cand { double val; int sink; int it; };      // sink: 1 when the column is unassigned
// the sequential scan's choice among two candidates (it = position in the scan)
__device__ __forceinline__ lsa_cand lsa_better(const lsa_cand& a, const lsa_cand& b) {
    if (a.it < 0) return b;
    if (b.it < 0) return a;
    if (a.val != b.val) return (a.val < b.val) ? a : b;
    if (a.sink != b.sink) return a.sink ? a : b;
    if (a.sink) return (a.it > b.it) ? a : b;           // every later unassigned column of equal cost replaces the choice
    return (a.it < b.it) ? a : b;                       // else the first in scan order is kept
}

__global__ __launch_bounds__(LSA_NT) void lsa_kernel(const double* __restrict__ costs, int nr, int nc, int negate, int use_lds,
                                                     double* __restrict__ g_f64, int* __restrict__ g_i32,
                                                     int32_t* __restrict__ out_col4row, int32_t* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsa_smem[];
    __shared__ double red_val[16];
    __shared__ int red_sink[16], red_it[16];
    __shared__ int s_state[4];                          // 0: chosen position, 1: sink, 2: current row, 3: remaining count
    __shared__ double s_min;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const double* cost = costs + (long long)b * nr * nc;
    const double sgn = negate ? -1.0 : 1.0;
    // per-matrix state: u (nr), v (nc), spc (nc) doubles; col4row (nr), row4col (nc), path (nc), remaining (nc), sr_list (nr), sc_list (nc) ints
    double* gf = g_f64 + (long long)b * (nr + 2 * (long long)nc);
    int* gi = g_i32 + (long long)b * (2 * (long long)nr + 4 * (long long)nc);
    double* u = gf;
    double* v = use_lds ? reinterpret_cast<double*>(lsa_smem) : gf + nr;
    double* spc = use_lds ? v + nc : gf + nr + nc;
    int* col4row = gi;
    int* sr_list = gi + nr;
    int* row4col = use_lds ? reinterpret_cast<int*>(spc + nc) : gi + 2 * nr;
    int* path = use_lds ? row4col + nc : gi + 2 * nr + nc;
    int* remaining = use_lds ? path + nc : gi + 2 * nr + 2 * nc;
    int* sc_list = gi + 2 * nr + 3 * nc;

    for (int i = t; i < nr; i += LSA_NT) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = t; j < nc; j += LSA_NT) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    __syncthreads();

    for (int cur = 0; cur < nr; ++cur) {
        for (int j = t; j < nc; j += LSA_NT) { spc[j] = DM_INF_F64; remaining[j] = nc - j - 1; }
        if (t == 0) { s_state[1] = -1; s_state[2] = cur; s_state[3] = nc; s_min = 0.0; }
        __syncthreads();
        int n_sr = 0, n_sc = 0;                          // (uniform: every thread counts the same steps)
        while (true) {
            const int i = s_state[2], nrem = s_state[3];
            const double min_val = s_min;
            const double ui = u[i];
            const double* crow = cost + (long long)i * nc;
            lsa_cand best{DM_INF_F64, 0, -1};
            for (int it = t; it < nrem; it += LSA_NT) {
                const int j = remaining[it];
                const double r = ((min_val + sgn * crow[j]) - ui) - v[j];          // SciPy's operation order
                double s = spc[j];
                if (r < s) { path[j] = i; spc[j] = r; s = r; }
                const lsa_cand c{s, row4col[j] == -1 ? 1 : 0, it};
                // a thread's positions ascend: folding them in order is the sequential rule restricted to this thread
                best = lsa_better(best, c);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                lsa_cand o;
                o.val = __shfl_xor(best.val, off);
                o.sink = __shfl_xor(best.sink, off);
                o.it = __shfl_xor(best.it, off);
                best = lsa_better(best, o);
            }
            if (lane == 0) { red_val[wave] = best.val; red_sink[wave] = best.sink; red_it[wave] = best.it; }
            if (t == 0) sr_list[n_sr] = i;
            ++n_sr;
            __syncthreads();
            if (t == 0) {
                lsa_cand w{red_val[0], red_sink[0], red_it[0]};
                for (int q = 1; q < LSA_NT / 64; ++q) w = lsa_better(w, lsa_cand{red_val[q], red_sink[q], red_it[q]});
                // (an infeasible matrix -- all remaining entries infinite -- leaves it = -1: stop with what is assigned)
                if (w.it < 0 || !(w.val < DM_INF_F64)) { s_state[1] = -2; }
                else {
                    const int j = remaining[w.it];
                    s_min = w.val;
                    if (row4col[j] == -1) s_state[1] = j; else s_state[2] = row4col[j];
                    sc_list[n_sc] = j;
                    remaining[w.it] = remaining[nrem - 1];
                    s_state[3] = nrem - 1;
                }
            }
            ++n_sc;
            __syncthreads();
            if (s_state[1] != -1) break;
        }
        const int sink = s_state[1];
        if (sink < 0) {                                  // infeasible (SciPy: ValueError "cost matrix is infeasible"): reported in info
            if (t == 0) atomicMax(&info[b], 1);
            break;
        }
        const double min_val = s_min;
        // dual updates (every visited row / scanned column once: order free)
        for (int q = t; q < n_sr; q += LSA_NT) {
            const int i2 = sr_list[q];
            if (i2 == cur) u[i2] += min_val;
            else u[i2] += min_val - spc[col4row[i2]];
        }
        for (int q = t; q < n_sc; q += LSA_NT) {
            const int j2 = sc_list[q];
            v[j2] -= min_val - spc[j2];
        }
        __syncthreads();
        if (t == 0) {                                    // flip the augmenting path
            int j = sink;
            while (true) {
                const int i2 = path[j];
                row4col[j] = i2;
                const int jn = col4row[i2];
                col4row[i2] = j;
                j = jn;
                if (i2 == cur) break;
            }
        }
        __syncthreads();
    }
    for (int i = t; i < nr; i += LSA_NT) out_col4row[(long long)b * nr + i] = col4row[i];
}

// info[b] = 2 when the matrix holds a NaN or an infinity of the sign SciPy rejects (-inf when minimising, +inf when
// maximising: "matrix contains invalid numeric entries")
__global__ __launch_bounds__(256) void lsa_validate_kernel(const double* __restrict__ cost, long long n, int negate, int32_t* __restrict__ info) {
    const int b = blockIdx.y;
    bool bad = false;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const double x = cost[(long long)b * n + e];
        bad = bad || (x != x) || (negate ? x == DM_INF_F64 : x == -DM_INF_F64);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicMax(&info[b], 2);
}

// out (B, nc, nr) = in (B, nr, nc) transposed
__global__ __launch_bounds__(256) void lsa_transpose_kernel(const double* __restrict__ in, int nr, int nc, double* __restrict__ out) {
    __shared__ double tile[32][33];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int q = ty; q < 32; q += 8) {
        const int r = r0 + q, c = c0 + tx;
        if (r < nr && c < nc) tile[q][tx] = in[((long long)b * nr + r) * nc + c];
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int c = c0 + q, r = r0 + tx;
        if (r < nr && c < nc) out[((long long)b * nc + c) * nr + r] = tile[tx][q];
    }
}
// col_of_row[b][r] = the column whose owner is r (transposed problem solved: owner of column c given), else -1
__global__ __launch_bounds__(256) void lsa_invert_kernel(const int32_t* __restrict__ row_of_col, int nr, int nc, int32_t* __restrict__ col_of_row) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nc) return;
    const int r = row_of_col[(long long)b * nc + c];
    if (r >= 0 && r < nr) col_of_row[(long long)b * nr + r] = c;
}

extern "C" int dm_linear_sum_assignment(dm_ctx* ctx, int B, int nr, int nc, const double* cost, int maximize, int32_t* col_of_row,
                                        int32_t* info) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && nr > 0 && nc > 0, "sizes must be positive");
    DM_REQUIRE(ctx, cost && col_of_row && info, "null pointer");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const bool transposed = nr > nc;                   // SciPy solves the transposed problem when there are more rows than columns
    const int R = transposed ? nc : nr, Cn = transposed ? nr : nc;
    const size_t bT = transposed ? (size_t)B * nr * nc * 8 : 0;
    const size_t bF = (size_t)B * (R + 2 * (size_t)Cn) * 8, bI = (size_t)B * (2 * (size_t)R + 4 * (size_t)Cn) * 4;
    const size_t bO = transposed ? (size_t)B * R * 4 : 0;
    int rc = dm_ws_reserve(ctx, dm_align_up(bT) + dm_align_up(bF) + dm_align_up(bI) + dm_align_up(bO) + 4096);
    if (rc) return rc;
    double* Ct = transposed ? (double*)dm_ws_take(ctx, bT) : nullptr;
    double* gf = (double*)dm_ws_take(ctx, bF);
    int* gi = (int*)dm_ws_take(ctx, bI);
    int32_t* tmp = transposed ? (int32_t*)dm_ws_take(ctx, bO) : nullptr;
    if ((transposed && (!Ct || !tmp)) || !gf || !gi) return dm_fail(ctx, DM_ENOMEM, "assignment: workspace not reserved");
    DM_CHECK_HIP(ctx, hipMemsetAsync(info, 0, (size_t)B * 4, ctx->stream));
    {
        const long long nel = (long long)nr * nc;
        const int gx = (int)((nel + 256 * 16 - 1) / (256 * 16)) < 1024 ? (int)((nel + 256 * 16 - 1) / (256 * 16)) : 1024;
        DM_LAUNCH(ctx, "lsa_validate", lsa_validate_kernel, dim3(gx, B), dim3(256), 0, cost, nel, maximize ? 1 : 0, info);
    }
    if (transposed)
        DM_LAUNCH(ctx, "lsa_transpose", lsa_transpose_kernel, dim3(dm_cdiv(nc, 32), dm_cdiv(nr, 32), B), dim3(256), 0, cost, nr, nc, Ct);
    const int use_lds = Cn <= LSA_LDS_MAX_COLS;
    const size_t lds = use_lds ? (size_t)Cn * 28 + 64 : 0;
    rc = dm_grant_lds(ctx, (const void*)lsa_kernel, lds);
    if (rc) return rc;
    DM_LAUNCH(ctx, "lsa_shortest_augmenting_path", lsa_kernel, dim3(B), dim3(LSA_NT), lds, transposed ? Ct : cost, R, Cn, maximize ? 1 : 0,
              use_lds, gf, gi, transposed ? tmp : col_of_row, info);
    if (transposed) {
        DM_CHECK_HIP(ctx, hipMemsetAsync(col_of_row, 0xFF, (size_t)B * nr * 4, ctx->stream));
        DM_LAUNCH(ctx, "lsa_invert", lsa_invert_kernel, dim3(dm_cdiv(nc, 256), B), dim3(256), 0, tmp, nr, nc, col_of_row);
    }
    return DM_OK;
}

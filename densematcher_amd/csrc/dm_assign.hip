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
#include "dm_indicator_dev.h"

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

// the same rule on separate scalars (a struct passed by value ended up in scratch memory: a memory round trip per merge):
// (v, sk, it, j) <- the better of itself and (ov, osk, oit, oj)
__device__ __forceinline__ void lsa_merge(double& v, int& sk, int& it, int& j, double ov, int osk, int oit, int oj) {
    bool take;
    if (it < 0) take = true;
    else if (oit < 0) take = false;
    else if (v != ov) take = ov < v;
    else if (sk != osk) take = osk != 0;
    else if (sk) take = oit > it;
    else take = oit < it;
    v = take ? ov : v; sk = take ? osk : sk; it = take ? oit : it; j = take ? oj : j;
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

// the same rule with (sink, it, column) folded into ONE integer that orders the candidates of equal cost: an unassigned
// column beats an assigned one, among unassigned ones the LAST in scan order wins, among assigned ones the FIRST:
//   key = ((sink ? 8192 + it : 8191 - it) << 14) | column      (it, column < 8192: nc <= 8192 on this path);  -1 = no candidate
//   better = smaller cost, then larger key  (positions are distinct, so the column bits never decide)
// -- two comparisons per merge instead of five, and three shuffled words per butterfly level instead of five
__device__ __forceinline__ int lsa_key(int sink, int it, int j) { return ((sink ? 8192 + it : 8191 - it) << 14) | j; }
// Reductions inside a wave on the vector ALU (DPP), in TWO phases: the smallest cost first, then the largest key among the
// lanes that hold it -- one v_min_f64 / v_max_i32 per level and no divergent branch (a merged (cost, key) compare costs five
// times that).  Levels: lanes i ^ 1, i ^ 2 (quad permutes), 7 - i within 8 (row_half_mirror), 15 - i within 16 (row_mirror)
// leave every lane of a row of 16 with the row's result; row_bcast:15 into rows 1, 3 and row_bcast:31 into rows 2, 3 carry
// it to lane 63, which a readlane hands to the scalar unit.
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ int lsa_dpp(int x) {
    if constexpr (ROWS == 0xf) return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, false);      // (every lane has a source)
    else return __builtin_amdgcn_update_dpp(x, x, CTRL, ROWS, 0xf, false);                      // rows outside the mask keep x
}
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ double lsa_min_dpp(double v) {                                     // (costs are never NaN)
    return __builtin_fmin(v, __hiloint2double(lsa_dpp<CTRL, ROWS>(__double2hiint(v)), lsa_dpp<CTRL, ROWS>(__double2loint(v))));
}
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ int lsa_max_dpp(int k) { const int o = lsa_dpp<CTRL, ROWS>(k); return o > k ? o : k; }
__device__ __forceinline__ double lsa_wave_min(double v) {          // uniform result
    v = lsa_min_dpp<0xB1>(v); v = lsa_min_dpp<0x4E>(v); v = lsa_min_dpp<0x141>(v); v = lsa_min_dpp<0x140>(v);
    v = lsa_min_dpp<0x142, 0xa>(v); v = lsa_min_dpp<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ int lsa_wave_max(int k) {                // uniform result
    k = lsa_max_dpp<0xB1>(k); k = lsa_max_dpp<0x4E>(k); k = lsa_max_dpp<0x141>(k); k = lsa_max_dpp<0x140>(k);
    k = lsa_max_dpp<0x142, 0xa>(k); k = lsa_max_dpp<0x143, 0xc>(k);
    return __builtin_amdgcn_readlane(k, 63);
}

// Where a matrix's entries come from.  Matrices 0 .. n_lr-1 of a launch are mapped indicators given by their FACTORS (dm_indicator_dev.h:
// E2p (n_lr, nr, KP), P1p (n_lr, nc, KP), a1p (n_lr, nc)): a thread keeps the factor rows of its columns in registers and a cost row is
// KP fused multiply-adds per column from one broadcast row of E2p -- no N x N matrix is read (or needs to exist).  The others are
// dense (nr x nc float64 each, from `dense`).  KP = 0 instantiations know dense matrices only.
struct lsa_src { const double* dense; const double* E2p; const double* P1p; const double* a1p; int n_lr; };

// ---- the same search with the column state in registers ---------------------------------------------------------------------
// Thread t owns the columns t, t + NT, ... (CPT of them, nc <= NT CPT): their dual v, tentative cost, "scanned" flag and
// POSITION IN SCIPY'S `remaining` LIST live in registers, the cost row is read coalesced, and a step needs ONE barrier:
// every wave leaves its best candidate in a double-buffered LDS slot and every thread folds the 16 slots itself.  The
// list is never stored: SciPy removes the chosen entry by moving the last one into its place, so the only column whose
// position changes is the one at position nrem - 1, and its owner sees that in its own registers.  The arithmetic, the
// scan order used for ties and therefore the assignment are those of lsa_kernel (and SciPy).
//   WARM = 1: start from the column minima as duals with every column whose minimum sits in a still free row assigned to it
//   (Jonker-Volgenant's column reduction; a feasible dual pair with tight assigned edges, so the augmentations continue from
//   there to an optimum).  Fewer and shorter searches, but not SciPy's order: the caller accepts the result only when
//   lsa_unique_kernel finds the optimum unique (no slack-free edge outside the assignment), else it reruns with WARM = 0.
// returns true when the warm start stepped aside (nothing written): the caller runs the search in SciPy's order instead
template <int CPT, int WARM, int NT, int KP>
__device__ __forceinline__ bool lsa_reg_body(unsigned char* lsa_smem, const lsa_src& src, int nr, int nc, int negate,
                                             double* __restrict__ g_u, double* __restrict__ g_v, int32_t* __restrict__ out_col4row,
                                             int32_t* __restrict__ info) {
    // LDS: u (nr) doubles | row4col (nc), path (nc), col4row (nr) ints | slots
    double* u = reinterpret_cast<double*>(lsa_smem);
    int* row4col = reinterpret_cast<int*>(u + nr);
    int* path = row4col + nc;
    int* col4row = path + nc;
    __shared__ double sl_val[2][16];
    __shared__ int2 sl_it[2][16];                         // (the candidates' tie keys, lsa_key, and the rows their columns are assigned to)
    __shared__ int s_count[2];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool lr = KP > 0 && b < src.n_lr;               // (uniform) entries from the factors
    const double* cost = lr ? nullptr : src.dense + (long long)(b - src.n_lr) * nr * nc;
    const double* E2b = lr ? src.E2p + (long long)b * nr * (KP > 0 ? KP : 1) : nullptr;
    double psi[CPT][KP > 0 ? KP : 1], a1c[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int j = t + NT * q;
        a1c[q] = (lr && j < nc) ? src.a1p[(long long)b * nc + j] : 0.0;
#pragma unroll
        for (int k = 0; k < (KP > 0 ? KP : 1); ++k) psi[q][k] = (lr && j < nc) ? src.P1p[((long long)b * nc + j) * (KP > 0 ? KP : 1) + k] : 0.0;
    }
    // row i of the matrix, this thread's columns
    auto load_row = [&](int i, double (&cv_)[CPT]) {
        if (KP > 0 && lr) {
            const double* e = E2b + (long long)i * (KP > 0 ? KP : 1);
#pragma unroll
            for (int q = 0; q < CPT; ++q) cv_[q] = ind_value<(KP > 0 ? KP : 1)>(e, psi[q], a1c[q]);
        } else {
            const double* crow = cost + (long long)i * nc;
#pragma unroll
            for (int q = 0; q < CPT; ++q) { const int j = t + NT * q; cv_[q] = crow[j < nc ? j : nc - 1]; }
        }
    };
    const double sgn = negate ? -1.0 : 1.0;
    double v[CPT], spc[CPT];
    int pos[CPT];
    int sc[CPT];
    int r4c[CPT];                                         // row4col of the owned columns (refreshed after every path flip)
#pragma unroll
    for (int q = 0; q < CPT; ++q) { v[q] = 0.0; spc[q] = DM_INF_F64; pos[q] = 0; sc[q] = 0; r4c[q] = -1; }
    for (int i = t; i < nr; i += NT) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = t; j < nc; j += NT) { row4col[j] = -1; path[j] = -1; }
    __syncthreads();
    if (WARM) {
        // v_j = min_i c_ij (lowest i on ties); the column takes that row if no lower column claimed it
        int arg[CPT], cnt[CPT];                           // (cnt: how many entries of the column equal its minimum)
#pragma unroll
        for (int q = 0; q < CPT; ++q) { v[q] = DM_INF_F64; arg[q] = -1; cnt[q] = 0; }
        for (int i = 0; i < nr; ++i) {
            double crow_[CPT];
            load_row(i, crow_);
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const double c = sgn * crow_[q];
                const bool lt = c < v[q];
                cnt[q] = lt ? 1 : cnt[q] + (c == v[q] ? 1 : 0);
                arg[q] = lt ? i : arg[q];
                v[q] = lt ? c : v[q];
            }
        }
        // A matrix full of ties gains nothing from this start: two constant columns can be swapped (the optimum is not unique,
        // the result would be thrown away), and when many column minima are attained more than once the searches from these
        // duals wander through sheets of slack-free edges (the precise map's matrix, three entries per row: 370 ms against
        // 7 ms in SciPy's order).  Leave such a matrix to the exact rerun.
        {
            int tied = 0, flat = 0;
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int j = t + NT * q;
                tied += (j < nc && cnt[q] >= 2) ? 1 : 0; flat += (j < nc && cnt[q] == nr) ? 1 : 0;
            }
            if (t == 0) { s_count[0] = 0; s_count[1] = 0; }
            __syncthreads();
            if (tied) atomicAdd(&s_count[0], tied);
            if (flat) atomicAdd(&s_count[1], flat);
            __syncthreads();
            if (s_count[0] * 8 > nc || (s_count[1] >= 2 && nr >= 2)) return true;          // (uniform)
        }
        // claim: col4row[i] = the lowest column whose minimum is in row i (LDS atomics), then the winners record themselves
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int j = t + NT * q;
            if (j < nc && arg[q] >= 0 && v[q] < DM_INF_F64) atomicMin(reinterpret_cast<unsigned int*>(&col4row[arg[q]]), (unsigned int)j);
        }
        __syncthreads();                                  // (col4row was -1 = 0xffffffff: any column index is smaller as unsigned)
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int j = t + NT * q;
            if (j < nc && arg[q] >= 0 && col4row[arg[q]] == j) row4col[j] = arg[q];
            if (j < nc && !(v[q] < DM_INF_F64)) v[q] = 0.0;     // a column without a finite entry: no usable minimum
        }
        __syncthreads();
    }
    for (int cur = 0; cur < nr; ++cur) {
        if (WARM && col4row[cur] != -1) continue;         // (uniform) assigned by the column reduction
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int j = t + NT * q;
            spc[q] = DM_INF_F64; sc[q] = 0; pos[q] = nc - 1 - j;          // remaining[it] = nc - it - 1
            r4c[q] = j < nc ? row4col[j] : -1;
        }
        int i = cur, nrem = nc, sink = -1, step = 0;
        double min_val = 0.0;
        bool infeasible = false;
        // the tie key of column j at list position p is kb + p * ks (lsa_key): both fixed for the search.  After a warm start
        // the order is free and kb is just the column with a "free" flag above it.
        int kb[CPT], ks[CPT], pth[CPT];
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int j = t + NT * q;
            const bool fr = r4c[q] == -1;
            kb[q] = WARM ? ((fr ? 1 << 14 : 0) | j) : (((fr ? 8192 : 8191) << 14) | j);
            ks[q] = fr ? (1 << 14) : -(1 << 14); pth[q] = -1;
        }
        double ui = u[i];
        double cv[CPT];
        load_row(i, cv);
        while (true) {
            // relax the open columns (no branch: a select per column; the predecessor stays in a register until the search ends)
            double cand[CPT];
            const double mu = min_val - ui;
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int j = t + NT * q;
                const bool open = (j < nc) & (sc[q] == 0);
                const double r = WARM ? (sgn * cv[q] + mu) - v[q] : ((min_val + sgn * cv[q]) - ui) - v[q];     // (WARM = 0: SciPy's operation order)
                const bool better = open & (r < spc[q]);
                spc[q] = better ? r : spc[q];
                pth[q] = better ? i : pth[q];
                cand[q] = open ? spc[q] : DM_INF_F64;
            }
            // the wave's smallest tentative cost, then WHICH column: SciPy's order needs the largest key among the columns at
            // that cost (one lane almost always: a ballot finds it, the butterfly runs only on a real tie); after a warm start
            // the order is free (the result is accepted only when the optimum is unique) and the first lane's column is taken
            double tv = cand[0];
#pragma unroll
            for (int q = 1; q < CPT; ++q) tv = __builtin_fmin(tv, cand[q]);
            const double wmin = lsa_wave_min(tv);
            const bool feas = wmin < DM_INF_F64;
            int tk = -1, trow = -1;
#pragma unroll
            for (int q = CPT - 1; q >= 0; --q) {
                const int key = WARM ? kb[q] : kb[q] + pos[q] * ks[q];
                const bool eq = (cand[q] == wmin) & feas;
                const bool take = eq & (key > tk);         // (warm start: a free column first, any order otherwise)
                tk = take ? key : tk; trow = take ? r4c[q] : trow;
            }
            int wsel = -1, wrow = -1;
            {
                const unsigned long long have = __builtin_amdgcn_ballot_w64(tk >= 0);
                if (have) {
                    int src = (int)__builtin_ctzll(have);
                    if (WARM) {
                        const unsigned long long fr = __builtin_amdgcn_ballot_w64(tk >= (1 << 14));
                        if (fr) src = (int)__builtin_ctzll(fr);
                    } else if ((have & (have - 1)) != 0) {
                        const int best = lsa_wave_max(tk);
                        src = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(tk == best));
                    }
                    wsel = __builtin_amdgcn_readlane(tk, src); wrow = __builtin_amdgcn_readlane(trow, src);
                }
            }
            const int par = step & 1;
            if (lane == 0) { sl_val[par][wave] = wmin; sl_it[par][wave] = make_int2(wsel, wrow); }
            __syncthreads();
            // every wave folds the NT / 64 wave results itself: slot (lane mod NT / 64), butterfly over those lanes
            const double sv = sl_val[par][lane & (NT / 64 - 1)];
            const int2 sk = sl_it[par][lane & (NT / 64 - 1)];
            double wv = lsa_min_dpp<0xB1>(sv); wv = lsa_min_dpp<0x4E>(wv);
            if (NT > 256) wv = lsa_min_dpp<0x141>(wv);
            if (NT > 512) wv = lsa_min_dpp<0x140>(wv);
            wv = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(wv)), __builtin_amdgcn_readfirstlane(__double2loint(wv)));
            int wkey = -1, inext = -1;
            {
                const bool mine = (sv == wv) & (sk.x >= 0);
                const unsigned long long have = __builtin_amdgcn_ballot_w64(mine) & ((1ull << (NT / 64)) - 1);
                if (have) {
                    int src = (int)__builtin_ctzll(have);
                    if (WARM) {
                        const unsigned long long fr = __builtin_amdgcn_ballot_w64(mine & (sk.x >= (1 << 14))) & ((1ull << (NT / 64)) - 1);
                        if (fr) src = (int)__builtin_ctzll(fr);
                    } else if ((have & (have - 1)) != 0) {
                        int m = mine ? sk.x : -1;
                        m = lsa_max_dpp<0xB1>(m); m = lsa_max_dpp<0x4E>(m);
                        if (NT > 256) m = lsa_max_dpp<0x141>(m);
                        if (NT > 512) m = lsa_max_dpp<0x140>(m);
                        src = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(mine & (sk.x == m)));
                    }
                    wkey = __builtin_amdgcn_readlane(sk.x, src); inext = __builtin_amdgcn_readlane(sk.y, src);
                }
            }
            int wj, wsk, wit;
            if (WARM) { wj = wkey < 0 ? -1 : (wkey & 16383); wsk = wkey < 0 ? 0 : (wkey >> 14); wit = wkey < 0 ? -1 : 0; }
            else {
                const int wK = wkey >> 14;
                wj = wkey < 0 ? -1 : (wkey & 16383);
                wsk = (wkey >= 0 && wK >= 8192) ? 1 : 0;
                wit = wkey < 0 ? -1 : (wsk ? wK - 8192 : 8191 - wK);
            }
            ++step;
            if (wit < 0 || !(wv < DM_INF_F64)) { infeasible = true; break; }
            min_val = wv;
            if (!wsk) {                                    // the next row's loads go out before the list is updated
                i = inext;
                load_row(i, cv);
                ui = u[i];
            }
            // remove the chosen column from the list: the column at the last position takes its place
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int j = t + NT * q;
                const bool hit = j == wj;
                if (!WARM) {
                    const bool moved = (sc[q] == 0) & !hit & (pos[q] == nrem - 1);
                    pos[q] = moved ? wit : pos[q];
                }
                sc[q] = hit ? 1 : sc[q];
            }
            --nrem;
            if (wsk) { sink = wj; break; }
        }
        if (infeasible) {                                 // (uniform) SciPy: ValueError "cost matrix is infeasible"
            if (t == 0) atomicMax(&info[b], 1);
            break;
        }
        // dual updates: every scanned column once; the row it was assigned to (if any) is a visited row
        if (t == 0) u[cur] += min_val;
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            if (sc[q]) {
                path[t + NT * q] = pth[q];                // (the flip walks scanned columns only)
                const double dlt = min_val - spc[q];
                if (r4c[q] != -1) u[r4c[q]] += dlt;       // (distinct rows: one column each)
                v[q] -= dlt;
            }
        }
        __syncthreads();
        if (t == 0) {                                     // flip the augmenting path
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
    for (int i = t; i < nr; i += NT) {
        out_col4row[(long long)b * nr + i] = col4row[i];
        if (g_u) g_u[(long long)b * nr + i] = u[i];
    }
    if (g_v) {
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int j = t + NT * q;
            if (j < nc) g_v[(long long)b * nc + j] = v[q];
        }
    }
    return false;
}
// run_if (the rerun in SciPy's order): only the matrices flagged 1 = "the warm start's optimum may not be unique".
// skip_warm (the warm-start launch): a matrix the warm start steps aside from is searched in SciPy's order right here, by the
// same workgroup (other matrices of the batch are still in their warm searches: no reason to wait for the rerun launch), and
// flagged 2 = "done in order": the uniqueness check cannot lower that and the rerun passes it by.
template <int CPT, int WARM, int NT, int KP>
__global__ __launch_bounds__(NT) void lsa_reg_kernel(const lsa_src src, int nr, int nc, int negate,
                                                         double* __restrict__ g_u, double* __restrict__ g_v,
                                                         int32_t* __restrict__ out_col4row, int32_t* __restrict__ info,
                                                         const int32_t* __restrict__ run_if, int32_t* __restrict__ skip_warm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsa_smem[];
    if (run_if && run_if[blockIdx.x] != 1) return;
    const bool aside = lsa_reg_body<CPT, WARM, NT, KP>(lsa_smem, src, nr, nc, negate, g_u, g_v, out_col4row, info);
    if (WARM && aside) {                                  // (uniform)
        __syncthreads();
        lsa_reg_body<CPT, 0, NT, KP>(lsa_smem, src, nr, nc, negate, g_u, g_v, out_col4row, info);
        if (threadIdx.x == 0 && skip_warm) skip_warm[blockIdx.x] = 2;
    }
}

// ---- is the optimum unique? ---------------------------------------------------------------------------------------------------
// Given optimal duals (u, v) and the assignment, another assignment of the same cost exists exactly when the "equality graph"
// has an alternating cycle: with r(j) the row assigned to column j, the directed graph on the rows with an edge i -> r(j) for
// every edge (i, j) OUTSIDE the assignment without slack (c_ij - u_i - v_j = 0) has a directed cycle.  (Slack-free edges by
// themselves are the rule, not the exception: the assignment polytope is degenerate, every search tree leaves them behind.)
// A slack-free edge into an UNASSIGNED column (rectangular problems) is reported as a possible tie without further analysis.
//   lsa_rowofcol_kernel  r(j)
//   lsa_tight_kernel     adjacency bit matrix adj[i][i'] (nr x ceil(nr / 32) words per matrix), zeroed by the caller
//   lsa_acyclic_kernel   one workgroup per matrix peels rows of in-degree zero (Kahn); rows left over = a cycle: tie[b] = 1
__global__ __launch_bounds__(256) void lsa_rowofcol_kernel(const int32_t* __restrict__ col4row, int nr, int nc, int32_t* __restrict__ row4col) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nr) return;
    const int j = col4row[(long long)b * nr + i];
    if (j >= 0 && j < nc) row4col[(long long)b * nc + j] = i;
}
__global__ __launch_bounds__(256) void lsa_tight_kernel(const double* __restrict__ costs, int nr, int nc, int negate, const double* __restrict__ g_u,
                                                        const double* __restrict__ g_v, const int32_t* __restrict__ col4row,
                                                        const int32_t* __restrict__ row4col, unsigned int* __restrict__ adj, int nw,
                                                        int32_t* __restrict__ tie) {
    const int b = blockIdx.y;
    const double sgn = negate ? -1.0 : 1.0;
    const double* cost = costs + (long long)b * nr * nc;
    bool freecol = false;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)nr * nc; e += (long long)gridDim.x * 256) {
        const int i = (int)(e / nc), j = (int)(e - (long long)i * nc);
        if (col4row[(long long)b * nr + i] == j) continue;
        const double c = sgn * cost[e], ui = g_u[(long long)b * nr + i], vj = g_v[(long long)b * nc + j];
        const double slack = (c - ui) - vj;
        // (relative to the magnitudes that formed it: a few hundred ulps count as "no slack")
        if (!(slack > 1e-13 * (fabs(c) + fabs(ui) + fabs(vj)))) {
            const int r = row4col[(long long)b * nc + j];
            if (r < 0) freecol = true;
            else atomicOr(&adj[((long long)b * nr + i) * nw + (r >> 5)], 1u << (r & 31));
        }
    }
    if (__any(freecol) && (threadIdx.x & 63) == 0) atomicMax(&tie[b], 1);
}
// the same for matrices given by their factors: a thread owns a column (its factor row in registers) and sweeps 64 rows
template <int KP>
__global__ __launch_bounds__(256) void lsa_tight_lr_kernel(const double* __restrict__ E2p, const double* __restrict__ P1p, const double* __restrict__ a1p,
                                                           int nr, int nc, int negate, const double* __restrict__ g_u, const double* __restrict__ g_v,
                                                           const int32_t* __restrict__ col4row, const int32_t* __restrict__ row4col,
                                                           unsigned int* __restrict__ adj, int nw, int32_t* __restrict__ tie) {
    const int b = blockIdx.z, j = blockIdx.x * 256 + threadIdx.x, i0 = blockIdx.y * 64;
    const double sgn = negate ? -1.0 : 1.0;
    bool freecol = false;
    if (j < nc) {
        double p[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) p[k] = P1p[((long long)b * nc + j) * KP + k];
        const double a1 = a1p[(long long)b * nc + j], vj = g_v[(long long)b * nc + j];
        const int r = row4col[(long long)b * nc + j];
        const int i1 = min(nr, i0 + 64);
        for (int i = i0; i < i1; ++i) {
            if (col4row[(long long)b * nr + i] == j) continue;
            const double c = sgn * ind_value<KP>(E2p + ((long long)b * nr + i) * KP, p, a1), ui = g_u[(long long)b * nr + i];
            const double slack = (c - ui) - vj;
            if (!(slack > 1e-13 * (fabs(c) + fabs(ui) + fabs(vj)))) {
                if (r < 0) freecol = true;
                else atomicOr(&adj[((long long)b * nr + i) * nw + (r >> 5)], 1u << (r & 31));
            }
        }
    }
    if (__any(freecol) && (threadIdx.x & 63) == 0) atomicMax(&tie[b], 1);
}
// IN_LDS: the in-degrees and the front live in the LDS (2 nr ints of dynamic LDS): a round of the peel is two scans and three barriers,
// and the equality graph of a fitted map's indicator is hundreds of rounds deep -- with the counters in global memory every round paid
// device-memory atomics and round trips (192 matrices of 2048 rows: 5.7 ms, as long as its slowest matrix; r05).
template <bool IN_LDS>
__global__ __launch_bounds__(1024) void lsa_acyclic_kernel(const unsigned int* __restrict__ adj, int nr, int nw, int* __restrict__ indeg_ws,
                                                           int32_t* __restrict__ tie) {
    extern __shared__ int acy_sh[];
    __shared__ int s_removed, s_front;
    const int b = blockIdx.x, t = threadIdx.x;
    const unsigned int* A = adj + (long long)b * nr * nw;
    int* indeg = IN_LDS ? acy_sh : indeg_ws + (long long)b * 2 * nr;       // in-degree, then -1 once removed
    int* front = indeg + nr;                              // the rows removed in the current round
    for (int i = t; i < nr; i += 1024) indeg[i] = 0;
    if (t == 0) s_removed = 0;
    __syncthreads();
    for (int i = t; i < nr; i += 1024)
        for (int w = 0; w < nw; ++w) {
            unsigned int m = A[(long long)i * nw + w];
            while (m) { const int r = (w << 5) + __ffs((int)m) - 1; m &= m - 1; atomicAdd(&indeg[r], 1); }
        }
    __syncthreads();
    for (int round = 0; round <= nr; ++round) {
        if (t == 0) s_front = 0;
        __syncthreads();
        for (int i = t; i < nr; i += 1024)
            if (indeg[i] == 0) { indeg[i] = -1; front[atomicAdd(&s_front, 1)] = i; }
        __syncthreads();
        const int nfront = s_front;
        if (nfront == 0) break;
        if (t == 0) s_removed += nfront;
        for (int q = t; q < nfront * nw; q += 1024) {      // the out-edges of the removed rows
            const int i = front[q / nw], w = q % nw;
            unsigned int m = A[(long long)i * nw + w];
            while (m) { const int r = (w << 5) + __ffs((int)m) - 1; m &= m - 1; atomicSub(&indeg[r], 1); }
        }
        __syncthreads();
    }
    if (t == 0 && s_removed < nr) atomicMax(&tie[b], 1);   // rows that never reach in-degree zero sit on or behind a cycle
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

static size_t lsa_ws_bytes(int B, int nr, int nc) {
    const bool transposed = nr > nc;
    const int R = transposed ? nc : nr, Cn = transposed ? nr : nc;
    const size_t bT = transposed ? (size_t)B * nr * nc * 8 : 0;
    const size_t bF = (size_t)B * (R + 2 * (size_t)Cn) * 8, bI = (size_t)B * (2 * (size_t)R + 4 * (size_t)Cn) * 4;
    const size_t bO = transposed ? (size_t)B * R * 4 : 0;
    return dm_align_up(bT) + dm_align_up(bF) + dm_align_up(bI) + dm_align_up(bO) + dm_align_up((size_t)B * 4) +
           dm_align_up((size_t)B * R * dm_cdiv(R, 32) * 4) + dm_align_up((size_t)B * Cn * 4) + dm_align_up((size_t)B * 2 * R * 4) + 8192;
}

// B matrices: the first n_lr from factors (KP = 16 | 32), the others dense.  The workspace is reserved by the caller (lsa_ws_bytes).
static int lsa_run(dm_ctx* ctx, int B, int nr, int nc, const double* dense, int n_lr, int KP, const double* E2p, const double* P1p,
                   const double* a1p, int maximize, int32_t* col_of_row, int32_t* info) {
    const int n_dense = B - n_lr;
    const bool transposed = nr > nc;                   // SciPy solves the transposed problem when there are more rows than columns
    if (n_lr > 0 && transposed) return dm_fail(ctx, DM_EINVAL, "assignment from factors: needs nr <= nc");
    const int R = transposed ? nc : nr, Cn = transposed ? nr : nc;
    const size_t bT = transposed ? (size_t)B * nr * nc * 8 : 0;
    const size_t bF = (size_t)B * (R + 2 * (size_t)Cn) * 8, bI = (size_t)B * (2 * (size_t)R + 4 * (size_t)Cn) * 4;
    const size_t bO = transposed ? (size_t)B * R * 4 : 0;
    int rc;
    double* Ct = transposed ? (double*)dm_ws_take(ctx, bT) : nullptr;
    double* gf = (double*)dm_ws_take(ctx, bF);
    int* gi = (int*)dm_ws_take(ctx, bI);
    int32_t* tmp = transposed ? (int32_t*)dm_ws_take(ctx, bO) : nullptr;
    if ((transposed && (!Ct || !tmp)) || !gf || !gi) return dm_fail(ctx, DM_ENOMEM, "assignment: workspace not reserved");
    if (n_dense > 0) {
        const long long nel = (long long)nr * nc;
        const int gx = (int)((nel + 256 * 16 - 1) / (256 * 16)) < 1024 ? (int)((nel + 256 * 16 - 1) / (256 * 16)) : 1024;
        DM_LAUNCH(ctx, "lsa_validate", lsa_validate_kernel, dim3(gx, n_dense), dim3(256), 0, dense, nel, maximize ? 1 : 0, info + n_lr);
    }
    if (transposed)
        DM_LAUNCH(ctx, "lsa_transpose", lsa_transpose_kernel, dim3(dm_cdiv(nc, 32), dm_cdiv(nr, 32), B), dim3(256), 0, dense, nr, nc, Ct);
    // register-resident search (columns owned by threads) when the columns fit 8 per thread and the row state fits the LDS
    const size_t lds_reg = (size_t)R * 8 + ((size_t)2 * Cn + R) * 4 + 64;
    if (ctx->opt_lsa_reg && Cn <= 8 * LSA_NT && lds_reg <= 150 * 1024) {
        const lsa_src src{transposed ? Ct : dense, E2p, P1p, a1p, n_lr};
        int32_t* outp = transposed ? tmp : col_of_row;
        int32_t* tie = (int32_t*)dm_ws_take(ctx, (size_t)B * 4);
        if (!tie) return dm_fail(ctx, DM_ENOMEM, "assignment: workspace not reserved");
        double* gu = gf;
        double* gv = gf + (size_t)B * R;
        // 512 threads with twice the columns each up to 4096 columns (a step is a barrier, a 16-slot merge and one cost row: fewer
        // waves make the barrier and the merge cheaper; 0.61 -> 0.54 s on the notebook's 2048 x 2048 indicator, 256 threads: 0.64 s)
        const int nt = Cn <= 8 * 512 ? 512 : LSA_NT;
        const int per = dm_cdiv(Cn, nt);
        const int cpt = per <= 1 ? 1 : (per <= 2 ? 2 : (per <= 4 ? 4 : 8));
        if (n_lr > 0 && (cpt > 4 || nt != 512)) return dm_fail(ctx, DM_EINVAL, "assignment from factors: at most 2048 columns");
#define LSA_REG(CPT_, WARM_, RUNIF_, NT_, KP_)                                                                         \
        {                                                                                                              \
            rc = dm_grant_lds(ctx, (const void*)lsa_reg_kernel<CPT_, WARM_, NT_, KP_>, lds_reg);                       \
            if (rc) return rc;                                                                                         \
            DM_LAUNCH(ctx, (WARM_ ? "lsa_shortest_augmenting_path" : (RUNIF_ ? "lsa_rerun_in_order" : "lsa_in_order")), (lsa_reg_kernel<CPT_, WARM_, NT_, KP_>), dim3(B), dim3(NT_), lds_reg, src, R, Cn, \
                      maximize ? 1 : 0, gu, gv, outp, info, RUNIF_, WARM_ ? tie : (int32_t*)nullptr);                                                   \
        }
#define LSA_REG_NT(WARM_, RUNIF_, NT_)                                                                                 \
        if (cpt == 1) LSA_REG(1, WARM_, RUNIF_, NT_, 0) else if (cpt == 2) LSA_REG(2, WARM_, RUNIF_, NT_, 0)           \
        else if (cpt == 4) LSA_REG(4, WARM_, RUNIF_, NT_, 0) else LSA_REG(8, WARM_, RUNIF_, NT_, 0)
#define LSA_REG_LR(WARM_, RUNIF_)                                                                                      \
        if (KP == 16) { if (cpt == 1) LSA_REG(1, WARM_, RUNIF_, 512, 16) else if (cpt == 2) LSA_REG(2, WARM_, RUNIF_, 512, 16) else LSA_REG(4, WARM_, RUNIF_, 512, 16) } \
        else { if (cpt == 1) LSA_REG(1, WARM_, RUNIF_, 512, 32) else if (cpt == 2) LSA_REG(2, WARM_, RUNIF_, 512, 32) else LSA_REG(4, WARM_, RUNIF_, 512, 32) }
#define LSA_REG_CPT(WARM_, RUNIF_)                                                                                     \
        if (n_lr > 0) { LSA_REG_LR(WARM_, RUNIF_) }                                                                    \
        else if (nt == 512) { LSA_REG_NT(WARM_, RUNIF_, 512) } else { LSA_REG_NT(WARM_, RUNIF_, 1024) }
        if (ctx->opt_lsa_reg >= 2 && R == Cn) {
            // column-reduction start (not SciPy's order; square problems only: the duals of columns that stay unassigned would have
            // to be zero), accepted per matrix when its optimum is provably unique, else redone exactly.  Measured and dropped:
            // Jonker-Volgenant's augmenting row reduction in front of the searches (two sweeps): 0.61 -> 0.87 s on the notebook's
            // indicator matrix (near-ties make its evictions ping-pong), 0.19 -> 0.40 s on a random matrix.
            DM_CHECK_HIP(ctx, hipMemsetAsync(tie, 0, (size_t)B * 4, ctx->stream));
            LSA_REG_CPT(1, (const int32_t*)nullptr)
            const long long nel = (long long)R * Cn;
            const int gx = (int)((nel + 256 * 16 - 1) / (256 * 16)) < 2048 ? (int)((nel + 256 * 16 - 1) / (256 * 16)) : 2048;
            const int nw = dm_cdiv(R, 32);
            unsigned int* adj = (unsigned int*)dm_ws_take(ctx, (size_t)B * R * nw * 4);
            int32_t* r4c = (int32_t*)dm_ws_take(ctx, (size_t)B * Cn * 4);
            int* indeg = (int*)dm_ws_take(ctx, (size_t)B * 2 * R * 4);
            if (!adj || !r4c || !indeg) return dm_fail(ctx, DM_ENOMEM, "assignment: workspace not reserved");
            DM_CHECK_HIP(ctx, hipMemsetAsync(adj, 0, (size_t)B * R * nw * 4, ctx->stream));
            DM_CHECK_HIP(ctx, hipMemsetAsync(r4c, 0xFF, (size_t)B * Cn * 4, ctx->stream));
            DM_LAUNCH(ctx, "lsa_unique_rowofcol", lsa_rowofcol_kernel, dim3(dm_cdiv(R, 256), B), dim3(256), 0, (const int32_t*)outp, R, Cn, r4c);
            if (n_lr > 0) {
                const dim3 gl(dm_cdiv(Cn, 256), dm_cdiv(R, 64), n_lr);
                if (KP == 16) DM_LAUNCH(ctx, "lsa_unique_tight", lsa_tight_lr_kernel<16>, gl, dim3(256), 0, E2p, P1p, a1p, R, Cn, maximize ? 1 : 0, (const double*)gu,
                                        (const double*)gv, (const int32_t*)outp, (const int32_t*)r4c, adj, nw, tie);
                else DM_LAUNCH(ctx, "lsa_unique_tight", lsa_tight_lr_kernel<32>, gl, dim3(256), 0, E2p, P1p, a1p, R, Cn, maximize ? 1 : 0, (const double*)gu,
                               (const double*)gv, (const int32_t*)outp, (const int32_t*)r4c, adj, nw, tie);
            }
            if (n_dense > 0)
                DM_LAUNCH(ctx, "lsa_unique_tight", lsa_tight_kernel, dim3(gx, n_dense), dim3(256), 0, src.dense, R, Cn, maximize ? 1 : 0, (const double*)(gu + (size_t)n_lr * R),
                          (const double*)(gv + (size_t)n_lr * Cn), (const int32_t*)(outp + (size_t)n_lr * R), (const int32_t*)(r4c + (size_t)n_lr * Cn),
                          adj + (size_t)n_lr * R * nw, nw, tie + n_lr);
            if ((size_t)R * 8 <= 60 * 1024) DM_LAUNCH(ctx, "lsa_unique_acyclic", lsa_acyclic_kernel<true>, dim3(B), dim3(1024), (size_t)R * 8, (const unsigned int*)adj, R, nw, indeg, tie);
            else DM_LAUNCH(ctx, "lsa_unique_acyclic", lsa_acyclic_kernel<false>, dim3(B), dim3(1024), 0, (const unsigned int*)adj, R, nw, indeg, tie);
            LSA_REG_CPT(0, (const int32_t*)tie)
        } else {
            LSA_REG_CPT(0, (const int32_t*)nullptr)
        }
#undef LSA_REG_CPT
#undef LSA_REG_LR
#undef LSA_REG_NT
#undef LSA_REG
        if (transposed) {
            DM_CHECK_HIP(ctx, hipMemsetAsync(col_of_row, 0xFF, (size_t)B * nr * 4, ctx->stream));
            DM_LAUNCH(ctx, "lsa_invert", lsa_invert_kernel, dim3(dm_cdiv(nc, 256), B), dim3(256), 0, tmp, nr, nc, col_of_row);
        }
        return DM_OK;
    }
    if (n_lr > 0) return dm_fail(ctx, DM_EINVAL, "assignment from factors: sizes outside the register-resident search");
    const int use_lds = Cn <= LSA_LDS_MAX_COLS;
    const size_t lds = use_lds ? (size_t)Cn * 28 + 64 : 0;
    rc = dm_grant_lds(ctx, (const void*)lsa_kernel, lds);
    if (rc) return rc;
    DM_LAUNCH(ctx, "lsa_shortest_augmenting_path", lsa_kernel, dim3(B), dim3(LSA_NT), lds, transposed ? Ct : dense, R, Cn, maximize ? 1 : 0,
              use_lds, gf, gi, transposed ? tmp : col_of_row, info);
    if (transposed) {
        DM_CHECK_HIP(ctx, hipMemsetAsync(col_of_row, 0xFF, (size_t)B * nr * 4, ctx->stream));
        DM_LAUNCH(ctx, "lsa_invert", lsa_invert_kernel, dim3(dm_cdiv(nc, 256), B), dim3(256), 0, tmp, nr, nc, col_of_row);
    }
    return DM_OK;
}

extern "C" int dm_linear_sum_assignment(dm_ctx* ctx, int B, int nr, int nc, const double* cost, int maximize, int32_t* col_of_row,
                                        int32_t* info) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && nr > 0 && nc > 0, "sizes must be positive");
    DM_REQUIRE(ctx, cost && col_of_row && info, "null pointer");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = dm_ws_reserve(ctx, lsa_ws_bytes(B, nr, nc));
    if (rc) return rc;
    DM_CHECK_HIP(ctx, hipMemsetAsync(info, 0, (size_t)B * 4, ctx->stream));
    return lsa_run(ctx, B, nr, nc, cost, 0, 0, nullptr, nullptr, nullptr, maximize, col_of_row, info);
}

// 1 when dm_lsa_indicator takes these sizes (else the caller forms the dense indicators and calls dm_linear_sum_assignment)
extern "C" int dm_lsa_indicator_ok(dm_ctx* ctx, int N1, int N2, int k1, int k2) {
    if (!ctx || !ctx->opt_lsa_reg) return 0;
    return (k1 >= 1 && k2 >= 1 && k1 <= 32 && k2 <= 32 && N2 <= N1 && N1 <= 2048 && N2 >= 1) ? 1 : 0;
}

// The assignments of n_ind mapped indicators (Phi2 C Phi1^T diag(a1), N2 x N1 each, given by their factors) and of n_dense dense
// N2 x N1 matrices in ONE launch (a workgroup per matrix): col_of_row / info hold the indicators' results first, then the dense ones'.
template <typename TR>
static int lsa_indicator_impl(dm_ctx* ctx, int n_ind, int N1, int N2, int k1, int k2, const TR* Phi1, int ld1, const TR* Phi2, int ld2,
                              const TR* mass1, const double* C, int n_dense, const double* dense, int maximize, int32_t* col_of_row, int32_t* info) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, n_ind > 0 && n_dense >= 0 && N1 > 0 && N2 > 0, "sizes must be positive");
    DM_REQUIRE(ctx, Phi1 && Phi2 && mass1 && C && col_of_row && info && (n_dense == 0 || dense), "null pointer");
    DM_REQUIRE(ctx, ld1 >= k1 && ld2 >= k2, "eigenvector row stride smaller than the map size");
    DM_REQUIRE(ctx, dm_lsa_indicator_ok(ctx, N1, N2, k1, k2), "lsa_indicator: maps up to 32 x 32, N2 <= N1 <= 2048");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const int B = n_ind + n_dense, KP = k1 <= 16 ? 16 : 32;
    const size_t bE2 = (size_t)n_ind * N2 * KP * 8, bP1 = (size_t)n_ind * N1 * KP * 8, bA = (size_t)n_ind * N1 * 8;
    int rc = dm_ws_reserve(ctx, dm_align_up(bE2) + dm_align_up(bP1) + dm_align_up(bA) + lsa_ws_bytes(B, N2, N1) + 4096);
    if (rc) return rc;
    double* E2p = (double*)dm_ws_take(ctx, bE2);
    double* P1p = (double*)dm_ws_take(ctx, bP1);
    double* a1p = (double*)dm_ws_take(ctx, bA);
    if (!E2p || !P1p || !a1p) return dm_fail(ctx, DM_ENOMEM, "lsa_indicator: workspace not reserved");
    DM_CHECK_HIP(ctx, hipMemsetAsync(info, 0, (size_t)B * 4, ctx->stream));
    DM_LAUNCH(ctx, "indicator_e2", ind_e2_kernel<TR>, dim3(dm_cdiv(N2, 256), n_ind), dim3(256), 0, Phi2, ld2, N2, k1, k2, C, KP, E2p, info);
    DM_LAUNCH(ctx, "indicator_p1", ind_p1_kernel<TR>, dim3(dm_cdiv(N1, 256), n_ind), dim3(256), 0, Phi1, ld1, mass1, N1, k1, KP, P1p, a1p, info);
    return lsa_run(ctx, B, N2, N1, dense, n_ind, KP, E2p, P1p, a1p, maximize, col_of_row, info);
}
extern "C" int dm_lsa_indicator(dm_ctx* ctx, int n_ind, int N1, int N2, int k1, int k2, const float* Phi1, int ld1, const float* Phi2, int ld2,
                                const float* mass1, const double* C, int n_dense, const double* dense, int maximize, int32_t* col_of_row, int32_t* info) {
    return lsa_indicator_impl<float>(ctx, n_ind, N1, N2, k1, k2, Phi1, ld1, Phi2, ld2, mass1, C, n_dense, dense, maximize, col_of_row, info);
}
extern "C" int dm_lsa_indicator_f64(dm_ctx* ctx, int n_ind, int N1, int N2, int k1, int k2, const double* Phi1, int ld1, const double* Phi2, int ld2,
                                    const double* mass1, const double* C, int n_dense, const double* dense, int maximize, int32_t* col_of_row, int32_t* info) {
    return lsa_indicator_impl<double>(ctx, n_ind, N1, N2, k1, k2, Phi1, ld1, Phi2, ld2, mass1, C, n_dense, dense, maximize, col_of_row, info);
}

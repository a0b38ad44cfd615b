// Internal declarations shared by the libdensematch translation units.
// gfx950 (MI355X / CDNA4) only: wave64, f16/f32/f64 MFMA, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "densematch.h"

// (abi N: bumped whenever an existing entry point changes what it reads or writes -- abi 5: dm_fmap_energy_grad / dm_fmap_fit_steps read TEN
//  weights (w_area, w_conformal appended in round 4); dm_eigenbasis warm_start = 2; dm_laplacian_*, dm_fmap_fit_fused added)
#define DM_VERSION_STRING "densematch 0.5.0 (gfx950, abi 5)"

struct dm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // scratch arena: one allocation, bump-pointer per call, grown lazily
    char* ws = nullptr;
    size_t ws_bytes = 0;
    size_t ws_off = 0;

    // kernel timing
    std::string prof_name;
    std::vector<hipEvent_t> prof_events;   // pairs (start, stop)
    std::vector<const char*> prof_names;   // name of the launch each pair brackets (string literals of DM_LAUNCH)
    size_t prof_used = 0;                  // events used so far

    // largest dynamic-LDS size already granted to each kernel on this device (hipFuncSetAttribute is per device)
    std::unordered_map<const void*, size_t> lds_granted;

    // dm_set_option: selection between equivalent (always exact) code paths; tests use it to cover the non-default ones
    int opt_simnn_pipe = 1;      // 0: every similarity tile goes through the bounds-checked register-staged kernel
    int opt_knn_split = 1;       // 0: knn21 (ZoomOut, ICP, knn_query) on the float64 G kernel instead of the fp16 split
    int opt_solve_packed = 0;    // 1: the packed-storage solver for every system size it supports
    int opt_proj_onepass = 1;    // 1: the fp16-split projection reads the basis once (running scale per workgroup); 0: maxima pass + fp32 copy + r03 tile kernel
    int opt_fit_mfma = 1;        // fp32 element loop of dm_fmap_fit_fused, maps up to 16 x 16: the two products of an entry on v_mfma_f32_16x16x4_f32 (1) or on the packed vector FMA (0); both agree with the oracle to the same 1e-7
    int opt_fit_f32 = 0;         // 1: dm_fmap_fit_fused runs its element loop in fp32 (the reference's precision), 0: float64
    int opt_simnn_prio = 0;      // 1: the tile kernels raise their wave priority around the matrix instructions (s_setprio)
    int opt_simnn_big = 0;       // 1: the tile kernels run four waves of 128 x 128 (accumulators in AGPRs) instead of eight of 128 x 64
    int opt_simnn_band = 4;      // tile rows per band of the similarity kernels' tile order (0: row-major); 4 measured best at
                                 // N = 8192 (tools/simnn_band_sweep.py: 8.93 / 8.62 / 8.44 / 8.57 / 8.82 ms for 0 / 2 / 4 / 8 / 16), neutral at N = 2048
    int opt_lsa_reg = 2;         // linear assignment: 0 the LDS-state kernel, 1 the register-state kernel in SciPy's order, 2 the same
                                 // from a column-reduction start, kept where the optimum is provably unique, else redone in order
    int opt_solve_pcg = 1;       // 1: systems of order 65 .. 199 by the batched preconditioned conjugate-gradient iteration (dm_pcg.h), the direct solver as its fall-back; 0: direct
    int opt_solve_reg = 1;       // 0: the LDS-resident blocked solver also where the register-resident one (n <= 128) would run
    int opt_p2p_split = 2;       // four maps: 0 the float64 G kernel, 1 two passes of the two-key fp16 tile kernel, 2 one pass reducing in both directions (3: 4-wave shape)
    int opt_simnn_persist = 1;   // 0: one workgroup per similarity tile instead of one persistent workgroup per CU
    int opt_p2pfm_direct = 1;    // 0: p2p_to_FM on the LDS-staged 64 x 64 tile kernel with split-K partials + a reduce launch
    int opt_simnn1_wt = 4;       // the biased-key search of the fused ZoomOut iteration: 4 = 8 waves, 256 x 256 tiles, one workgroup per CU; 2 = 4 waves, 128 x 256, two per CU
    int opt_zoomout_fused = 1;   // 0: ZoomOut as six launches per iteration (embedding, row build, search, merge, exact, p2p_to_FM [+ reduce])
    int opt_energy_keep_gram = 0;  // 1: dm_fmap_energy_grad keeps P = A A^T, Q = B A^T of its FIRST call and reuses them while A, B
                                   // (pointers and sizes) stay the same: the caller promises not to change their contents (the L-BFGS
                                   // driver: the projected descriptors are fixed during a fit).  Setting the option again drops them.
    double* gram_keep = nullptr;   // (its own allocation: the workspace arena is recycled by every call)
    size_t gram_keep_bytes = 0;
    const void* gram_key_ptr[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int gram_key_dim[6] = {0, 0, 0, 0, 0, 0};
    bool gram_valid = false;
    bool gram_sums_valid = false;  // the basis sums p, s2 behind P, Q in the same block (written by the first evaluation that needs them)
    int n_cu = 0;                // multiProcessorCount of the device
    int32_t* pinned_words = nullptr;   // 64 ints of page-locked host memory + an event: small device -> host reads that must not stall the
    hipEvent_t pinned_event = nullptr; // launch queue (dm_pinned_words; the ICP's "have all polar iterations converged?")
    // Per-basis statistics kept between calls (dm_fm_to_p2p: the per-64-vertex maxima of |Phi2|, from which the power-of-two scale of
    // the target rows follows -- a property of the MESH, not of the map).  Keyed on (pointer, sizes), two buffers per
    // entry: a call reads the maxima the previous call on this basis left and writes its own into the other buffer.  The hint only
    // saves a pass over the basis: a call whose data turn out to have another scale than the hint's (a tensor rewritten in place, or
    // another one at the same address) re-evaluates that pair's rows exactly -- slower, never wrong.
    struct basis_stat { const void* ptr = nullptr; int B = 0, N = 0, k = 0, ld = 0, esz = 0, n = 0; double* buf[2] = {nullptr, nullptr}; int cur = 0; bool valid = false; };
    basis_stat stats[4];
    int stats_next = 0;
    int opt_basis_stats = 1;     // 0: no hints, every call takes its own pass over the basis for the target rows (fs_build_rows)
    const int32_t* last_flag_counts = nullptr;   // device: the four queue counters (64 ints apart) of the last tile pass with its own merge; null: none
    int last_flag_sets = 0;
};

// Experiment knobs (ablation variants that may produce WRONG results) exist only in a -DDM_EXPERIMENTS build, where
// they are read from the environment; the product library never calls getenv.
#ifdef DM_EXPERIMENTS
int dm_knob(const char* env_name, int dflt);
#else
static inline int dm_knob(const char*, int dflt) { return dflt; }
#endif

int dm_fail(dm_ctx* ctx, int code, const char* fmt, ...);
// the context's 64 page-locked words and their event, allocated at first use
int dm_pinned_words(dm_ctx* ctx, int32_t** words, hipEvent_t* ev);
// allow `func` to be launched with `bytes` of dynamic LDS (> 64 KiB needs an explicit opt-in); remembered per context
int dm_grant_lds(dm_ctx* ctx, const void* func, size_t bytes);

#define DM_CHECK_HIP(ctx, expr)                                                        \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return dm_fail(ctx, DM_EHIP, "%s failed: %s (%s:%d)", #expr,               \
                           hipGetErrorString(_e), __FILE__, __LINE__);                 \
    } while (0)

#define DM_REQUIRE(ctx, cond, msg)                                                     \
    do {                                                                               \
        if (!(cond)) return dm_fail(ctx, DM_EINVAL, "%s: requirement failed: %s (%s)", \
                                    __func__, #cond, msg);                             \
    } while (0)

// ---- workspace arena -------------------------------------------------------
// A call first declares the total it needs (dm_ws_reserve: may reallocate, which
// synchronises the stream so that no in-flight kernel still reads the old
// block), then carves aligned pieces with dm_ws_take.
int dm_ws_reserve(dm_ctx* ctx, size_t total_bytes);
void* dm_ws_take(dm_ctx* ctx, size_t bytes);
static inline size_t dm_align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- launch bookkeeping -----------------------------------------------------
// DM_LAUNCH(ctx, "name", kernel, grid, block, shmem, args...) launches on the
// ctx stream, brackets with events when `name` is being profiled, and returns
// DM_EHIP from the enclosing function on a launch error.
int dm_prof_begin(dm_ctx* ctx, const char* name);
int dm_prof_end(dm_ctx* ctx, int token);

#define DM_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                          \
    do {                                                                               \
        int _tok = dm_prof_begin(ctx, name);                                           \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);    \
        hipError_t _le = hipGetLastError();                                            \
        if (_le != hipSuccess)                                                         \
            return dm_fail(ctx, DM_EHIP, "launch of %s failed: %s", name,              \
                           hipGetErrorString(_le));                                    \
        dm_prof_end(ctx, _tok);                                                        \
    } while (0)

static inline __host__ __device__ int dm_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- internal building blocks (each in its own .hip) -------------------------
// K-major float64 copy of the first k columns of Phi:  out[b][c][i] = Phi[b][i][c]
// (c < k), zero for padded entries; out is (B, kpad, Npad).
// amax (nullable, zeroed by the caller): (B, DM_NCH) partial maxima of |Phi[:, :k]| (bit patterns of non-negative doubles)
constexpr int DM_NCH = 32;
// Real arrays at the ABI (eigenvectors Phi, lumped masses) come as fp32 or fp64 (the *_f64 entry points: the reference's
// own dtype, pyFM/mesh/trimesh.py:118); the launchers below are templates over TR = float | double, instantiated for both
// in their .hip.  Masses are always float64 inside the library (dm_widen_mass converts an fp32 vector once per call).
template <typename TR>
int dm_launch_phiT(dm_ctx* ctx, int B, int N, int k, const TR* Phi, int ld,
                   double* out, int kpad, int Npad, double* amax = nullptr);
// mass as float64: the pointer itself for TR = double, a converted copy in `buf` (n doubles of workspace) for float
int dm_widen_mass(dm_ctx* ctx, long long n, const float* mass, double* buf, const double** out);
int dm_widen_mass(dm_ctx* ctx, long long n, const double* mass, double* buf, const double** out);

// embT[b][r][j] = sum_m Cm[b][r][m] * Phi[b][j][m]   (r < kr, m < km), K-major f64
// (B, krpad, Npad); nrm[b][j] = sum_r embT[b][r][j]^2 (nullable).  Only entries (r < kr, j < N)
// are written; zero_first clears the whole buffer before (padding must read as 0).
// Cm is (B, kr, km) f64 with row stride ldc; if transC, Cm[b][m][r] is read instead.  embT may be null (norms only).
template <typename TR>
int dm_launch_embed(dm_ctx* ctx, int B, int N, int kr, int km, const TR* Phi, int ld,
                    const double* Cm, int ldc, long long strideC, int transC,
                    double* embT, int krpad, int Npad, double* nrm, int zero_first,
                    double* amax_part = nullptr,    // amax_part: (B, ceil(Npad / DM_EMB_COLS)) max |embT| per block of columns (nullable)
                    double* amax_in_part = nullptr,   // same shape: max |Phi[:, :km]| over the block's vertices (nullable)
                    const struct dm_embed_fx* fx = nullptr);
constexpr int DM_EMB_COLS = 64;
// The basis rows a norms-only embedding streams anyway can leave it as SPLIT fp16 rows of the tile kernels (what fs_build_rows_kernel
// writes in a pass of its own), when the power-of-two scale is known up front: `hint` = the per-block maxima a previous call on the
// same basis left (dm_ctx::basis_stat).  amax_copy (nullable): a second copy of amax_in_part, the next call's hint.
struct dm_embed_fx { _Float16* F; int D, rows_out; const double* hint /* (B) hinted max |Phi| per pair */; int n_hint; double* amax_copy; };
dm_ctx::basis_stat* dm_stat_entry(dm_ctx* ctx, const void* ptr, int B, int N, int k, int ld, int esz, int n);

// Fused G = A^T B tile kernel with the arg-reductions (see dm_p2p.hip).
struct dm_gred_args {
    int B, N2, N1;                    // G is N2 x N1
    int Kloop;                        // contraction depth actually traversed (multiple of 16, <= Kpad)
    int Ktrue = 0;                    // rows of BT that can be non-zero (<= Kloop; 0 = unknown, use Kloop)
    const double* AT; int N2pad;      // (B, Kpad, N2pad)  rows = Phi2^T
    const double* BT; int N1pad;      // (B, Kpad, N1pad)  rows = emb1^T
    int Kpad;
    const double* n1;                 // (B, N1pad) |emb1_j|^2          (knn21)
    const double* n2;                 // (B, N2pad) |Phi2_i C|^2        (knn12)
    const double* mass1;              // (B, N1)                        (ind21, ind12)
    int32_t* knn21; int32_t* knn12; int32_t* ind21; int32_t* ind12;   // any nullable
};
int dm_launch_gred(dm_ctx* ctx, const dm_gred_args& a);
size_t dm_gred_ws_bytes(int B, int N2, int N1);

// fp16 tile kernel + merge of the feature-similarity NN, reusable as a first pass (dm_simnn.hip)
// Rows queued for exact re-evaluation, and what prunes it: the tile pass's own per-partial top-2.  Partial q of row i covers the
// candidates [q pw, (q + 1) pw) and holds (best fp32 score, its index, second-best score).  A candidate whose fp32 score
// reaches the row's threshold is either the best of its partial or not above the partial's second-best score, so
//   second >= thr  -> every candidate of the partial is re-scored;   else best >= thr -> only the best one's block of 32.
// (Round 2 kept a separate plane of maxima per block of 32 candidates for this, N^2 / 32 floats per key: 40 % of the HBM writes
//  of the four-map pass and four more VALU instructions per accumulator block in its epilogue.)
struct dm_simnn_queue {
    const float* pb; const int32_t* pj; const float* ps;      // (B, nparts, Npad)
    int nparts; int pw; int Npad;
    const int32_t* flag_count; const int32_t* flag_list; const float* flag_thr;
};
// does block `sb` (32 candidates) of row i (pair b) have to be re-scored?  (device code of the exact kernels)
#ifdef __HIPCC__
__device__ __forceinline__ bool dm_simnn_keep(const float* __restrict__ pb, const int32_t* __restrict__ pj, const float* __restrict__ ps,
                                              int nparts, int pw, int Npad, int b, int i, int sb, float thr) {
    const int q = (sb * 32) / pw;
    if (q >= nparts) return false;
    const long long o = ((long long)b * nparts + q) * Npad + i;
    if (ps[o] >= thr) return true;
    return pb[o] >= thr && (pj[o] >> 5) == sb;
}
#endif
// Second reduction of the same fp16 products in one pass (the four maps of dm_fm_to_p2p, dm_knnsplit.hip).  A pass with key
// sets reads SPLIT rows: per 16 contraction indices [16 high halves | 16 low halves] of both operands, from which the kernel
// forms hx.hy + hx.ly + lx.hy (three k-steps per 32-halves stage); D = 32 ceil(K / 16).
//   key A = score + bias[j] (-> nn21 / q of dm_simnn_core),  key B = score * scale[j], or the plain score when scale is null
struct dm_simnn_cols {                // both directions in one pass: two more reductions, per SOURCE row over the targets
    const float* biasT;               // (B, N2) key A' = score + biasT[i]; key B' = score
    const float* tau_add;             // (B) max_i |biasT_i|
    int32_t* nn_a; int32_t* nn_b;     // (B, N1) arg-max of key A' / key B'
    dm_simnn_queue* q_a; dm_simnn_queue* q_b;
    const double* zero_b = nullptr;   // (B, N1) nullable: nn_b[j] = 0 where zero_b[j] == 0 (a zero indicator column: first index)
};
struct dm_simnn_dual {
    const float* bias;                // (B, N1)
    const float* scale;               // (B, N1), nullable
    const float* tau_add;             // (B) max_j |bias_j|: the biased key is bounded relative to |t| max|s| + tau_add
    const float* tau_mul;             // (B) max_j scale_j (nullable): the scaled key relative to |t| max|s| tau_mul
    int32_t* nn_b;                    // (B, N2) arg-max of key B
    dm_simnn_queue* q_b;              // its queue of ambiguous rows
    const dm_simnn_cols* cols = nullptr;
    bool padded = false;              // Ftgt / Fsrc hold pad256(N2) / pad256(N1) rows per pair (zero rows behind the real ones) and
                                      // bias / scale / biasT the same number of entries: any N2, N1 (edge tiles mask the padding)
    bool single = false;              // key A alone on split rows (ZoomOut's search): scale, nn_b, q_b, cols unused
};
// optional: the caller owns the control block and / or the merge of a tile pass (the fused ZoomOut iteration, dm_zoomfuse.hip)
struct dm_simnn_ext {
    void* ctl = nullptr;              // a ZEROED block of dm_simnn_ctl_bytes(B) (per-pair maxima, queue counters): no memset per call
    bool skip_merge = false;          // the tile pass only; what a merge needs comes back below
    const float* pb = nullptr; const int32_t* pj = nullptr; const float* ps = nullptr; int nparts = 0, pw = 0, N2pad = 0;
    const float* tnorm2 = nullptr; const unsigned int* smax2 = nullptr; float tau_scale = 0.f;
};
size_t dm_simnn_ctl_bytes(int B);
size_t dm_simnn_ws_bytes(int B, int N2, int N1, int dual = 0);
bool dm_simnn_dual_ok(const dm_ctx* ctx, int N2, int N1, int D, bool padded = false);
int dm_simnn_core(dm_ctx* ctx, int B, int N2, int N1, int D, const _Float16* Ftgt, int ldT, const _Float16* Fsrc, int ldS,
                  float rel_extra,
                  const int32_t* force_flag, int32_t* nn21, float* best, float* margin, dm_simnn_queue* q,
                  const dm_simnn_dual* dual = nullptr, dm_simnn_ext* ext = nullptr);

// knn21 alone (ZoomOut, ICP, knn_query): fp16-split first pass on the fp16 matrix cores + exact float64 re-evaluation of
// the ambiguous rows (dm_knnsplit.hip).  The target side (rows of AT) is prepared once for the largest contraction depth
// kf of a call; every search then passes the current depth in a.Ktrue and nS partial maxima of |BT| per pair in amaxS
// (what colnorm_kernel / dm_launch_embed emit).  Only AT, BT, n1, knn21 of dm_gred_args are used.
struct dm_knn_split_state {
    _Float16* Ft = nullptr; int ldT = 0; double* amaxT = nullptr; int kf = 0;
    bool enabled = false;             // false (DM_KNN_SPLIT=0): dm_launch_knn21 runs the float64 G kernel instead
};
size_t dm_knn_split_prep_bytes(int B, int N2, int kf);
size_t dm_knn_split_ws_bytes(int B, int N2, int N1, int kf);
int dm_knn_split_prepare(dm_ctx* ctx, int B, int N2, int N2pad, int Kpad, int kf, const double* AT, dm_knn_split_state* st);
int dm_launch_knn21(dm_ctx* ctx, const dm_gred_args& a, const dm_knn_split_state& st, const double* amaxS, int nS);   // nS maxima per pair

// all four maps of dm_fm_to_p2p: two passes of the two-key fp16 tile kernel + exact float64 re-evaluation (dm_knnsplit.hip)
bool dm_fm_split_ok(const dm_ctx* ctx, int N2, int N1, int K);
size_t dm_fm_split_ws_bytes(int B, int N2, int N1, int K);
size_t dm_fm_split_zero_bytes(int B);      // block the caller zeroes: the per-pair bounds (max |bias|, max mass) the pass accumulates with atomicMax
// pre (nullable): the target rows Fx taken from the arena by the caller (the FIRST dm_fm_split_ws_bytes piece) and -- built = true --
// already written by the second embedding with the scale of `hint`; the pass then checks the hint against the maxima the embedding
// measured and re-evaluates a pair exactly where they give different scales
struct dm_fm_split_pre { _Float16* Fx; bool built; const double* hint /* (B) pair maxima the rows were scaled with */; double* pair_out /* (B) nullable: this call's pair maxima */; };
template <typename TR>
int dm_launch_fm_split(dm_ctx* ctx, const dm_gred_args& a, const double* amaxS, int nS, const double* amaxT, int nT, void* zeroed,
                       const TR* Phi2, int ld2, const dm_fm_split_pre* pre = nullptr);
_Float16* dm_fm_split_take_fx(dm_ctx* ctx, int B, int N2, int K, int* D_out, int* rows_out);

template <typename TR>
int dm_fm_split_build_rows(dm_ctx* ctx, int B, int N, int K, const TR* Phi, int ld, const double* amaxT, int nT, int D, _Float16* F, int rows_out);

// C[b] = Phi2[:, :k2]^T (mass2 * Phi1[p21, :k1]) (dm_zoomout.hip)
template <typename TR>
int dm_launch_p2p_to_fm(dm_ctx* ctx, int B, int N1, int N2, int k1, int k2, const int32_t* p21,
                        const TR* Phi1, int ld1, const TR* Phi2, int ld2, const double* mass2,
                        double* C, int ldc, long long strideC,
                        const double* Xs = nullptr, int ldx = 0);   // Xs: mass2 * Phi2 as dm_p2pfm_prescale builds it (a caller that
                                                                    // converts many maps on the same target builds it once); null: built per call
size_t dm_p2pfm_ws_bytes(int B, int N2, int k1, int k2);
size_t dm_p2pfm_xs_bytes(int B, int N2, int k2);
template <typename TR>
int dm_p2pfm_prescale(dm_ctx* ctx, int B, int N2, int k2, const TR* Phi2, int ld2, const double* mass2, double* Xs);

// dm_fmap_c00: sign(Phi1[0,0] Phi2[0,0]) sqrt(area2 / area1) per pair (pyFM/functional.py:654-658).  One workgroup of 256
// threads per pair; also run as an extra workgroup of a projection's maxima pass (dm_fmap_fit), same arithmetic.
template <typename TR>
struct dm_c00_args {
    const TR* Phi1; long long s1; const TR* Phi2; long long s2; const TR* mass1; const TR* mass2; int N1, N2; double* c00;
};
#ifdef __HIPCC__
template <typename TR>
__device__ __forceinline__ void dm_c00_body(const dm_c00_args<TR>& z, int b, int t, double (&red)[2][4]) {
    double a1 = 0.0, a2 = 0.0;               // (threads beyond 256 of a larger workgroup only take part in the barrier)
    for (int i = t; i < z.N1 && t < 256; i += 256) a1 += (double)z.mass1[(long long)b * z.N1 + i];
    for (int i = t; i < z.N2 && t < 256; i += 256) a2 += (double)z.mass2[(long long)b * z.N2 + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a1 += __shfl_xor(a1, off);
        a2 += __shfl_xor(a2, off);
    }
    if ((t & 63) == 0 && t < 256) { red[0][t >> 6] = a1; red[1][t >> 6] = a2; }
    __syncthreads();
    if (t == 0) {
        const double area1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const double area2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const double pr = (double)z.Phi1[b * z.s1] * (double)z.Phi2[b * z.s2];
        const double sgn = (pr > 0.0) ? 1.0 : ((pr < 0.0) ? -1.0 : 0.0);      // np.sign
        z.c00[b] = sgn * sqrt(area2 / area1);
    }
}
#endif

// fp16 split-operand MFMA projection (dm_project.hip); F must be fp16
size_t dm_project_f16split_ws(int B, int N, int D, int k, int ld, int real_bytes);
// cz (nullable): the maxima pass of this projection also computes the pinned entries c00 of the pairs (one more workgroup each)
template <typename TR>
int dm_project_f16split_launch(dm_ctx* ctx, int B, int N, int D, int k, const TR* Phi, int ld, const TR* mass, const void* F,
                               float* Ared, const float** partial_out, int* nsplit_out, const dm_c00_args<TR>* cz = nullptr);
template <typename TR>
int dm_project_f16split(dm_ctx* ctx, int B, int N, int D, int k, const TR* Phi, int ld, const TR* mass,
                        const void* F, float* Ared);

// Laplace-Beltrami operators of triangle meshes: dm_laplacian_ell (device assembly) and dm_tufted_cover (host, the intrinsic
// Delaunay cover of the "robust" Laplacian).
//
// Reference computation replaced: what TriMesh.process(robust = ...) assembles before its eigensolve
// (pyFM/mesh/trimesh.py:440-482):
//   robust = False   laplacian.cotangent_weights + laplacian.dia_area_mat (pyFM/mesh/laplacian.py:88-140, 5-40): cotangent stiffness
//                    matrix W (1/2 cot = 1/2 cos / sqrt(1 - cos^2) of the angle opposite each edge) and lumped masses (a third of
//                    the incident triangle areas);
//   robust = True    robust_laplacian.mesh_laplacian (external C++ wheel, trimesh.py:465-470; always taken by
//                    FunctionalMapping.preprocess, functional.py:294-295): the tufted intrinsic-Delaunay Laplacian of Sharp & Crane
//                    (SGP 2020) -- mollified edge lengths, the tufted double cover, intrinsic edge flips until every cover edge is
//                    Delaunay, cotangent weights from the intrinsic lengths, times 1/2.
// Split of the work: the flips are a sequential, pointer-chasing algorithm on a few thousand triangles -- host C++ here as in the
// wheel (dm_tufted_cover: no device, no context, thread safe: the Python layer runs one call per mesh on a thread pool) -- and
// everything that is arithmetic on all triangles at once runs on the device for the whole batch of meshes (dm_laplacian_ell):
// per-triangle cotangents / areas from vertex positions (reference formula) or from intrinsic lengths (Heron), accumulation per
// vertex in a fixed order, the symmetric scaling A^-1/2 W A^-1/2 and the ELL layout dm_eigenbasis reads.  (Round 4 did all of it
// with SciPy sparse matrices per mesh on the host: 0.93 s of a 64-pair compute_surface_map_batch call.)
#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <vector>

#include "dm_internal.h"

// =====================================================================================================================
// host: tufted cover + intrinsic Delaunay flips
// =====================================================================================================================
namespace {
struct cover {
    int nt;
    int32_t* T; double* L;                  // (nt, 3): corner vertices; side s runs from corner s to corner s + 1
    std::vector<int32_t> gt, gs;            // glue: the (triangle, side) across side s
    double cot_opp(int t, int s) const {    // cotangent of the angle opposite side s (Heron)
        const double a = L[3 * t + s], b = L[3 * t + (s + 1) % 3], c = L[3 * t + (s + 2) % 3];
        const double sp = 0.5 * (a + b + c);
        const double ar = std::sqrt(std::max(sp * (sp - a) * (sp - b) * (sp - c), 0.0));
        return (b * b + c * c - a * a) / (4.0 * std::max(ar, 1e-300));
    }
    void link(int ta, int sa, int tb, int sb) { gt[3 * ta + sa] = tb; gs[3 * ta + sa] = sb; gt[3 * tb + sb] = ta; gs[3 * tb + sb] = sa; }
};
}  // namespace

extern "C" int dm_tufted_cover(int n, int nf, const double* verts, const int32_t* faces, double mollify_factor, int32_t* T, double* L,
                               int32_t* info /* [flips, converged] */, double* mollify_eps) {
    if (n <= 0 || nf <= 0 || !verts || !faces || !T || !L) return DM_EINVAL;
    for (int e = 0; e < 3 * nf; ++e)
        if (faces[e] < 0 || faces[e] >= n) return DM_EINVAL;
    const int nt = 2 * nf;
    // ---- 1. intrinsic lengths, mollified: the smallest eps added to every length such that every triangle satisfies the triangle
    //         inequality with margin delta = mollify_factor * mean length
    std::vector<double> l0((size_t)3 * nf);
    double mean = 0.0;
    for (int f = 0; f < nf; ++f)
        for (int s = 0; s < 3; ++s) {
            const double* p = verts + 3 * (size_t)faces[3 * f + s];
            const double* q = verts + 3 * (size_t)faces[3 * f + (s + 1) % 3];
            const double d = std::sqrt((q[0] - p[0]) * (q[0] - p[0]) + (q[1] - p[1]) * (q[1] - p[1]) + (q[2] - p[2]) * (q[2] - p[2]));
            l0[3 * (size_t)f + s] = d;
            mean += d;
        }
    mean /= (double)(3 * (size_t)nf);
    const double delta = mollify_factor * mean;
    double eps = 0.0;
    for (int f = 0; f < nf; ++f) {
        const double a = l0[3 * (size_t)f], b = l0[3 * (size_t)f + 1], c = l0[3 * (size_t)f + 2];
        eps = std::max(eps, std::max(delta + a - b - c, std::max(delta + b - c - a, delta + c - a - b)));
    }
    if (mollify_eps) *mollify_eps = eps;
    // ---- 2. the cover: triangles 0 .. nf-1 front copies, nf .. 2 nf-1 back copies (a, c, b) with sides (2, 1, 0) of the front
    for (int f = 0; f < nf; ++f) {
        const int32_t a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
        T[3 * f] = a; T[3 * f + 1] = b; T[3 * f + 2] = c;
        T[3 * (f + nf)] = a; T[3 * (f + nf) + 1] = c; T[3 * (f + nf) + 2] = b;
        for (int s = 0; s < 3; ++s) {
            L[3 * (size_t)f + s] = l0[3 * (size_t)f + s] + eps;
            L[3 * (size_t)(f + nf) + s] = l0[3 * (size_t)f + (2 - s)] + eps;
        }
    }
    cover cv;
    cv.nt = nt; cv.T = T; cv.L = L;
    cv.gt.assign((size_t)3 * nt, -1); cv.gs.assign((size_t)3 * nt, -1);
    // ---- gluing: around an undirected edge with incident faces f_0 .. f_{m-1} (in face order) the copy of f_i that runs hi -> lo is
    //      glued to the copy of f_{i+1} that runs lo -> hi (cyclically)
    {
        // (edge key, halfedge index) packed into one integer and sorted: the incident faces of an edge in face order
        std::vector<unsigned long long> ko((size_t)3 * nf);
        std::vector<long long> key((size_t)3 * nf);
        std::vector<int> order((size_t)3 * nf);
        const bool pack = (double)n * (double)n * 3.0 * (double)nf < 1.8e19;
        for (int h = 0; h < 3 * nf; ++h) {
            const int f = h / 3, s = h % 3;
            const long long a = faces[3 * f + s], b = faces[3 * f + (s + 1) % 3];
            key[h] = std::min(a, b) * (long long)n + std::max(a, b);
            order[h] = h;
            if (pack) ko[h] = (unsigned long long)key[h] * (unsigned long long)(3 * (size_t)nf) + (unsigned long long)h;
        }
        if (pack) {
            std::sort(ko.begin(), ko.end());
            for (size_t q = 0; q < ko.size(); ++q) order[q] = (int)(ko[q] % (unsigned long long)(3 * (size_t)nf));
        } else {
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return key[x] < key[y]; });
        }
        size_t g0 = 0;
        while (g0 < order.size()) {
            size_t g1 = g0 + 1;
            while (g1 < order.size() && key[order[g1]] == key[order[g0]]) ++g1;
            auto sides = [&](int h, int& pt, int& ps, int& mt, int& ms) {      // plus: the copy running lo -> hi; minus: hi -> lo
                const int f = h / 3, s = h % 3;
                const bool fwd = faces[3 * f + s] <= faces[3 * f + (s + 1) % 3];
                if (fwd) { pt = f; ps = s; mt = f + nf; ms = 2 - s; }
                else { pt = f + nf; ps = 2 - s; mt = f; ms = s; }
            };
            for (size_t i = g0; i < g1; ++i) {
                const size_t nx = (i + 1 < g1) ? i + 1 : g0;
                int pt, ps, mt, ms, pt2, ps2, mt2, ms2;
                sides(order[i], pt, ps, mt, ms);
                sides(order[nx], pt2, ps2, mt2, ms2);
                cv.gt[3 * (size_t)mt + ms] = pt2; cv.gs[3 * (size_t)mt + ms] = ps2;
                cv.gt[3 * (size_t)pt2 + ps2] = mt; cv.gs[3 * (size_t)pt2 + ps2] = ms;
            }
            g0 = g1;
        }
    }
    // ---- 3. intrinsic Delaunay flips, one edge at a time from a stack (the intrinsic Delaunay triangulation is unique: the order
    //         of the flips does not matter)
    std::vector<std::pair<int, int>> stack;
    for (int t = 0; t < nt; ++t)
        for (int s = 0; s < 3; ++s)
            if (cv.cot_opp(t, s) + cv.cot_opp(cv.gt[3 * (size_t)t + s], cv.gs[3 * (size_t)t + s]) < -1e-12) stack.emplace_back(t, s);
    long long flips = 0;
    const long long limit = 20ll * nt;
    while (!stack.empty() && flips < limit) {
        const int t = stack.back().first, s = stack.back().second;
        stack.pop_back();
        const int t2 = cv.gt[3 * (size_t)t + s], s2 = cv.gs[3 * (size_t)t + s];
        if (t2 == t) continue;                                         // an edge glued to its own face cannot be flipped
        if (cv.cot_opp(t, s) + cv.cot_opp(t2, s2) >= -1e-12) continue;
        // triangle t: i -> j (side s), j -> k, k -> i;  triangle t2: j -> i (side s2), i -> m, m -> j
        const int s_1 = (s + 1) % 3, s_2 = (s + 2) % 3, q_1 = (s2 + 1) % 3, q_2 = (s2 + 2) % 3;
        const int32_t i = T[3 * t + s], j = T[3 * t + s_1], k = T[3 * t + s_2], m = T[3 * t2 + q_2];
        const double lij = L[3 * (size_t)t + s], ljk = L[3 * (size_t)t + s_1], lki = L[3 * (size_t)t + s_2];
        const double lim = L[3 * (size_t)t2 + q_1], lmj = L[3 * (size_t)t2 + q_2];
        // unfold the two triangles in the plane: i = (0, 0), j = (lij, 0), k above the axis, m below
        const double xk = (lki * lki - ljk * ljk + lij * lij) / (2.0 * lij);
        const double yk = std::sqrt(std::max(lki * lki - xk * xk, 0.0));
        const double xm = (lim * lim - lmj * lmj + lij * lij) / (2.0 * lij);
        const double ym = -std::sqrt(std::max(lim * lim - xm * xm, 0.0));
        const double lkm = std::hypot(xk - xm, yk - ym);
        if (!(lkm > 0.0)) continue;
        // neighbours across the four outer sides (one of them may be a side of the two triangles themselves: translate it)
        auto nb = [&](int tt, int ss) {
            std::pair<int, int> g(cv.gt[3 * (size_t)tt + ss], cv.gs[3 * (size_t)tt + ss]);
            if (g.first == t && g.second == s_2) return std::make_pair(t, 0);
            if (g.first == t2 && g.second == q_1) return std::make_pair(t, 1);
            if (g.first == t2 && g.second == q_2) return std::make_pair(t2, 0);
            if (g.first == t && g.second == s_1) return std::make_pair(t2, 1);
            return g;
        };
        const auto g_jk = nb(t, s_1), g_ki = nb(t, s_2), g_im = nb(t2, q_1), g_mj = nb(t2, q_2);
        // new triangles: t = (k, i, m): k->i, i->m, m->k;   t2 = (m, j, k): m->j, j->k, k->m
        T[3 * t] = k; T[3 * t + 1] = i; T[3 * t + 2] = m;
        L[3 * (size_t)t] = lki; L[3 * (size_t)t + 1] = lim; L[3 * (size_t)t + 2] = lkm;
        T[3 * t2] = m; T[3 * t2 + 1] = j; T[3 * t2 + 2] = k;
        L[3 * (size_t)t2] = lmj; L[3 * (size_t)t2 + 1] = ljk; L[3 * (size_t)t2 + 2] = lkm;
        cv.link(t, 0, g_ki.first, g_ki.second);
        cv.link(t, 1, g_im.first, g_im.second);
        cv.link(t2, 0, g_mj.first, g_mj.second);
        cv.link(t2, 1, g_jk.first, g_jk.second);
        cv.link(t, 2, t2, 2);
        ++flips;
        stack.emplace_back(t, 0); stack.emplace_back(t, 1); stack.emplace_back(t2, 0); stack.emplace_back(t2, 1);
    }
    if (info) { info[0] = (int32_t)std::min<long long>(flips, 0x7fffffff); info[1] = stack.empty() ? 1 : 0; }
    return DM_OK;
}

// the same for `count` meshes on `n_threads` host threads (<= 0: one per hardware thread, at most one per mesh)
extern "C" int dm_tufted_cover_batch(int count, const int32_t* n, const int32_t* nf, const double* const* verts, const int32_t* const* faces,
                                     double mollify_factor, int32_t* const* T, double* const* L, int32_t* info /* count x 2 */,
                                     double* mollify_eps /* count, nullable */, int n_threads) {
    if (count <= 0 || !n || !nf || !verts || !faces || !T || !L || !info) return DM_EINVAL;
    int nthr = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    nthr = std::max(1, std::min(nthr, count));
    std::atomic<int> next(0), status(DM_OK);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= count) break;
            const int rc = dm_tufted_cover(n[i], nf[i], verts[i], faces[i], mollify_factor, T[i], L[i], info + 2 * i, mollify_eps ? mollify_eps + i : nullptr);
            if (rc != DM_OK) status.store(rc);
        }
    };
    if (nthr == 1) work();
    else {
        std::vector<std::thread> pool;
        for (int q = 0; q < nthr; ++q) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    return status.load();
}

// =====================================================================================================================
// device: per-vertex accumulation of cotangent weights and lumped masses -> rows -> scaled ELL
// =====================================================================================================================
constexpr int LAP_MAXROW = 64;        // entries a vertex's row may hold (its neighbours + the diagonal); more: DM_EINVAL

struct lap_args {
    int B, N, nt;                     // meshes, vertices per mesh, triangles per mesh
    const int32_t* tri;               // (B, nt, 3)
    const double* len;                // (B, nt, 3) intrinsic side lengths, or null: from verts
    const double* verts;              // (B, N, 3)
    double scale;                     // 1, or 1/2 for a double cover
    const int32_t* nv;                // (B) vertices of each mesh (<= N), or null: N.  Meshes of fewer vertices / triangles ride along padded:
                                      // triangles (-1, -1, -1) are skipped, vertices >= nv[b] become decoupled rows (below)
};
__device__ __forceinline__ int lap_nv(const lap_args& a, int b) { return a.nv ? a.nv[b] : a.N; }

// count[b][v] = corners of vertex v
__global__ __launch_bounds__(256) void lap_count_kernel(lap_args a, int32_t* __restrict__ count, int32_t* __restrict__ bad) {
    const int b = blockIdx.y;
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= 3 * a.nt) return;
    const int v = a.tri[(long long)b * 3 * a.nt + h];
    if (v < 0) return;                                       // padding triangle
    if (v >= lap_nv(a, b)) { atomicOr(bad, 1); return; }
    atomicAdd(count + (long long)b * a.N + v, 1);
}
// exclusive prefix sums of a mesh's counts (one workgroup per mesh); cursor = offsets
__global__ __launch_bounds__(1024) void lap_scan_kernel(int N, const int32_t* __restrict__ count, int32_t* __restrict__ offset, int32_t* __restrict__ cursor) {
    __shared__ int part[1024];
    const int b = blockIdx.x, t = threadIdx.x;
    const int per = (N + 1023) / 1024, i0 = t * per, i1 = min(N, i0 + per);
    int s = 0;
    for (int i = i0; i < i1; ++i) s += count[(long long)b * N + i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = i0; i < i1; ++i) {
        offset[(long long)b * (N + 1) + i] = run; cursor[(long long)b * N + i] = run;
        run += count[(long long)b * N + i];
    }
    if (t == 1023) offset[(long long)b * (N + 1) + N] = part[1023];
}
// corner lists (unordered: the row kernel sorts each vertex's few corners)
__global__ __launch_bounds__(256) void lap_scatter_kernel(lap_args a, int32_t* __restrict__ cursor, int32_t* __restrict__ list) {
    const int b = blockIdx.y;
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= 3 * a.nt) return;
    const int v = a.tri[(long long)b * 3 * a.nt + h];
    if (v < 0 || v >= lap_nv(a, b)) return;
    const int pos = atomicAdd(cursor + (long long)b * a.N + v, 1);
    list[(long long)b * 3 * a.nt + pos] = h;
}

// the two weights a triangle gives the row of its corner s (vertex a; the others b = corner s + 1, c = corner s + 2):
//   w_ab = scale 1/2 cot(angle at c), w_ac = scale 1/2 cot(angle at b), and the mass share scale area / 3
__device__ __forceinline__ void lap_corner_weights(const lap_args& a, int b_, int t, int s, int& vb, int& vc, double& w_ab, double& w_ac, double& m) {
    const int32_t* T = a.tri + ((long long)b_ * a.nt + t) * 3;
    const int s1 = (s + 1) % 3, s2 = (s + 2) % 3;
    vb = T[s1]; vc = T[s2];
    if (a.len) {
        // intrinsic: side s = a -> b, side s1 = b -> c, side s2 = c -> a; Heron's area, cot = (sum of the adjacent squares - opposite^2) / (4 area)
        const double* L = a.len + ((long long)b_ * a.nt + t) * 3;
        const double lab = L[s], lbc = L[s1], lca = L[s2];
        const double sp = 0.5 * (lab + lbc + lca);
        const double ar = sqrt(fmax(sp * (sp - lab) * (sp - lbc) * (sp - lca), 0.0));
        const double den = 4.0 * fmax(ar, 1e-300);
        w_ab = a.scale * 0.5 * ((lbc * lbc + lca * lca - lab * lab) / den);      // angle at c is opposite side a -> b
        w_ac = a.scale * 0.5 * ((lab * lab + lbc * lbc - lca * lca) / den);      // angle at b is opposite side c -> a
        m = a.scale * ar / 3.0;
    } else {
        // extrinsic, the reference's arithmetic (laplacian.py:118-134): cos of the angle from the normalised edge vectors, cot = cos / sqrt(1 - cos^2)
        const double* V = a.verts + (long long)b_ * a.N * 3;
        const double* pa = V + 3 * (long long)T[s];
        const double* pb = V + 3 * (long long)vb;
        const double* pc = V + 3 * (long long)vc;
        const double ab[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
        const double ac[3] = {pc[0] - pa[0], pc[1] - pa[1], pc[2] - pa[2]};
        const double bc[3] = {pc[0] - pb[0], pc[1] - pb[1], pc[2] - pb[2]};
        const double lab = sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]);
        const double lac = sqrt(ac[0] * ac[0] + ac[1] * ac[1] + ac[2] * ac[2]);
        const double lbc = sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]);
        const double cos_c = (ac[0] * bc[0] + ac[1] * bc[1] + ac[2] * bc[2]) / (lac * lbc);          // angle at c: between c -> a and c -> b
        const double cos_b = -(ab[0] * bc[0] + ab[1] * bc[1] + ab[2] * bc[2]) / (lab * lbc);         // angle at b: between b -> a and b -> c
        w_ab = a.scale * 0.5 * cos_c / sqrt(1.0 - cos_c * cos_c);
        w_ac = a.scale * 0.5 * cos_b / sqrt(1.0 - cos_b * cos_b);
        const double cx = ab[1] * ac[2] - ab[2] * ac[1], cy = ab[2] * ac[0] - ab[0] * ac[2], cz = ab[0] * ac[1] - ab[1] * ac[0];
        m = a.scale * 0.5 * sqrt(cx * cx + cy * cy + cz * cz) / 3.0;
    }
}

// One thread per vertex: its corners in ascending (triangle, corner) order, their weights accumulated into the row's entries in
// that order (a fixed order of additions: the same mesh gives the same bits), the diagonal = the sum of the weights, entries sorted
// by column.  Rows leave as (row length, LAP_MAXROW x (col, val)); maxlen[b] = the longest row of the mesh.
__global__ __launch_bounds__(128) void lap_rows_kernel(lap_args a, const int32_t* __restrict__ offset, int32_t* __restrict__ list,
                                                       int32_t* __restrict__ rowlen, int32_t* __restrict__ rowcol, double* __restrict__ rowval,
                                                       double* __restrict__ mass, int32_t* __restrict__ maxlen, int32_t* __restrict__ bad) {
    const int b = blockIdx.y;
    const int v = blockIdx.x * 128 + threadIdx.x;
    if (v >= a.N) return;
    const int o0 = offset[(long long)b * (a.N + 1) + v], o1 = offset[(long long)b * (a.N + 1) + v + 1];
    int32_t* lst = list + (long long)b * 3 * a.nt;
    for (int i = o0 + 1; i < o1; ++i) {                      // insertion sort of the (few) corners
        const int x = lst[i];
        int j = i - 1;
        while (j >= o0 && lst[j] > x) { lst[j + 1] = lst[j]; --j; }
        lst[j + 1] = x;
    }
    int32_t* rc = rowcol + ((long long)b * a.N + v) * LAP_MAXROW;
    double* rv = rowval + ((long long)b * a.N + v) * LAP_MAXROW;
    int nr = 1;
    rc[0] = v; rv[0] = 0.0;
    double m = 0.0, diag = 0.0;
    bool overflow = false;
    for (int i = o0; i < o1; ++i) {
        const int h = lst[i], t = h / 3, s = h - 3 * t;
        int vb, vc; double w_ab, w_ac, mm;
        lap_corner_weights(a, b, t, s, vb, vc, w_ab, w_ac, mm);
        m += mm;
        diag += w_ab; diag += w_ac;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int col = e ? vc : vb;
            const double w = e ? w_ac : w_ab;
            if (col == v) { diag -= w; continue; }           // (a degenerate face that repeats a vertex contributes nothing to the row)
            int q = 1;
            while (q < nr && rc[q] != col) ++q;
            if (q == nr) {
                if (nr == LAP_MAXROW) { overflow = true; continue; }
                rc[nr] = col; rv[nr] = 0.0; ++nr;
            }
            rv[q] -= w;
        }
    }
    rv[0] = diag;
    for (int i = 1; i < nr; ++i) {                           // entries by column (the diagonal keeps its place among them)
        const int c = rc[i]; const double x = rv[i];
        int j = i - 1;
        while (j >= 0 && rc[j] > c) { rc[j + 1] = rc[j]; rv[j + 1] = rv[j]; --j; }
        rc[j + 1] = c; rv[j + 1] = x;
    }
    const bool pad = v >= lap_nv(a, b);                      // a padding vertex: a decoupled row, unit mass (lap_pad_kernel sets its entry)
    rowlen[(long long)b * a.N + v] = nr;
    mass[(long long)b * a.N + v] = pad ? 1.0 : m;
    atomicMax(maxlen + b, nr);
    if (overflow) atomicOr(bad, 2);
    if (!pad && !(m > 0.0)) atomicOr(bad, 4);
}

// ELL of L = A^-1/2 W A^-1/2 with the masses rounded to fp32 first (dm_eigenbasis solves the problem of the rounded masses and
// divides the eigenvectors by their square roots), W and mass (fp64, unrounded) as the caller's TriMesh wants them
__global__ __launch_bounds__(256) void lap_fill_kernel(int N, int nnz, const int32_t* __restrict__ rowlen, const int32_t* __restrict__ rowcol,
                                                       const double* __restrict__ rowval, const double* __restrict__ mass,
                                                       int32_t* __restrict__ ell_cols, double* __restrict__ ell_vals, double* __restrict__ w_vals,
                                                       float* __restrict__ mass32) {
    const int b = blockIdx.y;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)N * nnz) return;
    const int v = (int)(e / nnz), q = (int)(e - (long long)v * nnz);
    const int nr = rowlen[(long long)b * N + v];
    const long long o = ((long long)b * N + v) * nnz + q;
    if (q == 0) mass32[(long long)b * N + v] = (float)mass[(long long)b * N + v];
    if (q < nr) {
        const int c = rowcol[((long long)b * N + v) * LAP_MAXROW + q];
        const double w = rowval[((long long)b * N + v) * LAP_MAXROW + q];
        const double mv = (double)(float)mass[(long long)b * N + v], mc = (double)(float)mass[(long long)b * N + c];
        ell_cols[o] = c;
        ell_vals[o] = w / (sqrt(mv) * sqrt(mc));
        if (w_vals) w_vals[o] = w;
    } else {
        ell_cols[o] = v; ell_vals[o] = 0.0;
        if (w_vals) w_vals[o] = 0.0;
    }
}

// Padding vertices (v >= nv[b]) get the single entry L_vv = the Gershgorin bound max_i sum_q |L_iq| of their mesh's own operator: an
// eigenvalue at the upper end of the interval the eigensolver's filter damps, never inside the wanted part of the spectrum.
__global__ __launch_bounds__(256) void lap_pad_kernel(int N, int nnz, const int32_t* __restrict__ nv, double* __restrict__ ell_vals) {
    __shared__ double sh[256];
    const int b = blockIdx.x, t = threadIdx.x, nb = nv[b];
    if (nb >= N) return;
    double mx = 0.0;
    for (int i = t; i < nb; i += 256) {
        const double* vr = ell_vals + ((long long)b * N + i) * nnz;
        double s = 0.0;
        for (int q = 0; q < nnz; ++q) s += fabs(vr[q]);
        mx = fmax(mx, s);
    }
    sh[t] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (t < off) sh[t] = fmax(sh[t], sh[t + off]); __syncthreads(); }
    const double g = sh[0];
    for (int i = nb + t; i < N; i += 256) ell_vals[((long long)b * N + i) * nnz] = g;
}

extern "C" size_t dm_laplacian_rows_bytes(int B, int N, int nt) {
    return dm_align_up((size_t)B * N * 4) * 3 + dm_align_up((size_t)B * (N + 1) * 4) + dm_align_up((size_t)B * 3 * nt * 4) +
           dm_align_up((size_t)B * N * LAP_MAXROW * 4) + dm_align_up((size_t)B * N * LAP_MAXROW * 8) + dm_align_up((size_t)B * N * 8) +
           dm_align_up((size_t)B * 4) + 4096;
}

// Stage 1: rows of W and masses in `rows` (dm_laplacian_rows_bytes of device memory the caller owns); max_row (host) = the longest row
// over the batch (the ELL width stage 2 needs).  Synchronises the stream once (that number).
extern "C" int dm_laplacian_rows(dm_ctx* ctx, int B, int N, int nt, const int32_t* tri, const double* len, const double* verts, double scale,
                                 const int32_t* n_verts /*nullable*/, void* rows, int* max_row) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N > 0 && nt > 0 && B <= 65535, "sizes must be positive, B <= 65535");
    DM_REQUIRE(ctx, tri && (len || verts) && rows && max_row, "null pointer");
    DM_REQUIRE(ctx, scale > 0.0, "scale must be positive");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    char* p = (char*)rows;
    auto take = [&](size_t bytes) { char* q = p; p += dm_align_up(bytes); return q; };
    int32_t* count = (int32_t*)take((size_t)B * N * 4);
    int32_t* cursor = (int32_t*)take((size_t)B * N * 4);
    int32_t* rowlen = (int32_t*)take((size_t)B * N * 4);
    int32_t* offset = (int32_t*)take((size_t)B * (N + 1) * 4);
    int32_t* list = (int32_t*)take((size_t)B * 3 * nt * 4);
    int32_t* rowcol = (int32_t*)take((size_t)B * N * LAP_MAXROW * 4);
    double* rowval = (double*)take((size_t)B * N * LAP_MAXROW * 8);
    double* mass = (double*)take((size_t)B * N * 8);
    int32_t* maxlen = (int32_t*)take((size_t)(B + 1) * 4);
    int32_t* bad = maxlen + B;
    (void)rowlen; (void)rowcol; (void)rowval; (void)mass;
    lap_args a{B, N, nt, tri, len, verts, scale, n_verts};
    DM_CHECK_HIP(ctx, hipMemsetAsync(count, 0, (size_t)B * N * 4, ctx->stream));
    DM_CHECK_HIP(ctx, hipMemsetAsync(maxlen, 0, (size_t)(B + 1) * 4, ctx->stream));
    const dim3 gh(dm_cdiv(3 * nt, 256), B);
    DM_LAUNCH(ctx, "lap_count", lap_count_kernel, gh, dim3(256), 0, a, count, bad);
    DM_LAUNCH(ctx, "lap_scan", lap_scan_kernel, dim3(B), dim3(1024), 0, N, (const int32_t*)count, offset, cursor);
    DM_LAUNCH(ctx, "lap_scatter", lap_scatter_kernel, gh, dim3(256), 0, a, cursor, list);
    DM_LAUNCH(ctx, "lap_rows", lap_rows_kernel, dim3(dm_cdiv(N, 128), B), dim3(128), 0, a, (const int32_t*)offset, list, rowlen, rowcol, rowval, mass, maxlen, bad);
    std::vector<int32_t> h((size_t)B + 1);
    DM_CHECK_HIP(ctx, hipMemcpyAsync(h.data(), maxlen, (size_t)(B + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    DM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h[B] & 1) return dm_fail(ctx, DM_EINVAL, "laplacian: face indices must lie in [0, N)");
    if (h[B] & 2) return dm_fail(ctx, DM_EINVAL, "laplacian: a vertex has more than %d neighbours", LAP_MAXROW - 1);
    if (h[B] & 4) return dm_fail(ctx, DM_EINVAL, "laplacian: vertices with zero lumped mass (isolated vertices or degenerate faces): clean the mesh first");
    int mx = 1;
    for (int b = 0; b < B; ++b) mx = std::max(mx, (int)h[b]);
    *max_row = mx;
    return DM_OK;
}

// Stage 2: the ELL operands of dm_eigenbasis -- ell_cols / ell_vals (B, N, nnz) with nnz >= the max_row of stage 1, mass32 (B, N) -- and,
// optionally, the unscaled entries w_vals (B, N, nnz) of W and the fp64 masses mass64 (B, N) (what TriMesh.W / TriMesh.A hold).
extern "C" int dm_laplacian_ell(dm_ctx* ctx, int B, int N, int nt, const void* rows, int nnz, const int32_t* n_verts /*nullable*/, int32_t* ell_cols,
                                double* ell_vals, float* mass32, double* w_vals /*nullable*/, double* mass64 /*nullable*/) {
    if (!ctx) return DM_EINVAL;
    DM_REQUIRE(ctx, B > 0 && N > 0 && nt > 0 && nnz > 0 && nnz <= LAP_MAXROW && B <= 65535, "sizes");
    DM_REQUIRE(ctx, rows && ell_cols && ell_vals && mass32, "null pointer");
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const char* p = (const char*)rows;
    auto take = [&](size_t bytes) { const char* q = p; p += dm_align_up(bytes); return q; };
    take((size_t)B * N * 4); take((size_t)B * N * 4);
    const int32_t* rowlen = (const int32_t*)take((size_t)B * N * 4);
    take((size_t)B * (N + 1) * 4); take((size_t)B * 3 * nt * 4);
    const int32_t* rowcol = (const int32_t*)take((size_t)B * N * LAP_MAXROW * 4);
    const double* rowval = (const double*)take((size_t)B * N * LAP_MAXROW * 8);
    const double* mass = (const double*)take((size_t)B * N * 8);
    const long long per = (long long)N * nnz;
    DM_LAUNCH(ctx, "lap_fill_ell", lap_fill_kernel, dim3((unsigned)((per + 255) / 256), B), dim3(256), 0, N, nnz, rowlen, rowcol, rowval, mass, ell_cols,
              ell_vals, w_vals, mass32);
    if (n_verts) DM_LAUNCH(ctx, "lap_pad", lap_pad_kernel, dim3(B), dim3(256), 0, N, nnz, n_verts, ell_vals);
    if (mass64) DM_CHECK_HIP(ctx, hipMemcpyAsync(mass64, mass, (size_t)B * N * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return DM_OK;
}

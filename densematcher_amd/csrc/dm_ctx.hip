// Context, error reporting, scratch arena and per-kernel event timing.
#include <stdarg.h>

#include "dm_internal.h"

int dm_fail(dm_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

// reason of the last context-less failure (dm_create), returned by dm_last_error(NULL)
static thread_local std::string g_create_err = "null context";

static int create_fail(const char* what, hipError_t e) {
    char buf[512];
    snprintf(buf, sizeof(buf), "dm_create: %s%s%s", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
    g_create_err = buf;
    return DM_EHIP;
}

extern "C" int dm_create(int device, void* hip_stream, dm_ctx** out) {
    if (!out) return DM_EINVAL;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess) return create_fail("hipGetDeviceCount failed", e);
    if (ndev <= 0 || device < 0 || device >= ndev) return create_fail("no such HIP device", hipSuccess);
    e = hipSetDevice(device);
    if (e != hipSuccess) return create_fail("hipSetDevice failed", e);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return create_fail("hipGetDeviceProperties failed", e);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {                 // kernels exist for gfx950 only
        char buf[300];
        snprintf(buf, sizeof(buf), "device is %s, libdensematch is built for gfx950 only", prop.gcnArchName);
        return create_fail(buf, hipSuccess);
    }
    dm_ctx* ctx = new dm_ctx();
    ctx->n_cu = prop.multiProcessorCount;
    ctx->device = device;
    ctx->stream = (hipStream_t)hip_stream;
    *out = ctx;
    return DM_OK;
}

extern "C" int dm_destroy(dm_ctx* ctx) {
    if (!ctx) return DM_EINVAL;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->gram_keep) (void)hipFree(ctx->gram_keep);
    for (auto& e : ctx->stats) { if (e.buf[0]) (void)hipFree(e.buf[0]); }
    if (ctx->pinned_words) (void)hipHostFree(ctx->pinned_words);
    if (ctx->pinned_event) (void)hipEventDestroy(ctx->pinned_event);
    delete ctx;
    return DM_OK;
}

int dm_pinned_words(dm_ctx* ctx, int32_t** words, hipEvent_t* ev) {
    if (!ctx->pinned_words) DM_CHECK_HIP(ctx, hipHostMalloc((void**)&ctx->pinned_words, 64 * sizeof(int32_t), hipHostMallocDefault));
    if (!ctx->pinned_event) DM_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->pinned_event, hipEventDisableTiming));
    *words = ctx->pinned_words; *ev = ctx->pinned_event;
    return DM_OK;
}

#ifdef DM_EXPERIMENTS
#include <stdlib.h>
int dm_knob(const char* env_name, int dflt) {
    const char* e = getenv(env_name);
    return e ? atoi(e) : dflt;
}
#endif

extern "C" int dm_set_option(dm_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return DM_EINVAL;
    const std::string n = name;
    if (n == "simnn_pipe") ctx->opt_simnn_pipe = value;
    else if (n == "knn_split") ctx->opt_knn_split = value;
    else if (n == "solve_packed") ctx->opt_solve_packed = value;
    else if (n == "solve_reg") ctx->opt_solve_reg = value;
    else if (n == "simnn_band") ctx->opt_simnn_band = value;
    else if (n == "simnn_big") ctx->opt_simnn_big = value;
    else if (n == "simnn_prio") ctx->opt_simnn_prio = value;
    else if (n == "fit_f32") ctx->opt_fit_f32 = value;
    else if (n == "fit_mfma") ctx->opt_fit_mfma = value;
    else if (n == "solve_pcg") ctx->opt_solve_pcg = value;
    else if (n == "basis_stats") { ctx->opt_basis_stats = value; for (auto& e : ctx->stats) e.valid = false; }
    else if (n == "lsa_reg") ctx->opt_lsa_reg = value;
    else if (n == "p2p_split") ctx->opt_p2p_split = value;
    else if (n == "simnn_persist") ctx->opt_simnn_persist = value;
    else if (n == "p2pfm_direct") ctx->opt_p2pfm_direct = value;
    else if (n == "zoomout_fused") ctx->opt_zoomout_fused = value;
    else if (n == "simnn1_wt") ctx->opt_simnn1_wt = value;
    else if (n == "proj_onepass") ctx->opt_proj_onepass = value;
    else if (n == "energy_keep_gram") { ctx->opt_energy_keep_gram = value; ctx->gram_valid = false; }
    else return dm_fail(ctx, DM_EINVAL, "dm_set_option: unknown option '%s'", name);
    return DM_OK;
}

extern "C" const char* dm_last_error(const dm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
extern "C" const char* dm_version(void) { return DM_VERSION_STRING; }
extern "C" size_t dm_workspace_bytes(const dm_ctx* ctx) { return ctx ? ctx->ws_bytes : 0; }

// the statistics entry of a basis (dm_ctx::basis_stat), created on first sight; the oldest entry makes room.  Null when the two small
// buffers cannot be allocated (the caller then runs without hints).
dm_ctx::basis_stat* dm_stat_entry(dm_ctx* ctx, const void* ptr, int B, int N, int k, int ld, int esz, int n) {
    for (auto& e : ctx->stats)
        if (e.ptr == ptr && e.B == B && e.N == N && e.k == k && e.ld == ld && e.esz == esz && e.n == n && e.buf[0]) return &e;
    dm_ctx::basis_stat& e = ctx->stats[ctx->stats_next];
    ctx->stats_next = (ctx->stats_next + 1) % 4;
    const size_t need = (size_t)B * n * 8;
    if (!e.buf[0] || (size_t)e.B * e.n * 8 < need) {
        if (e.buf[0]) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(e.buf[0]); e.buf[0] = e.buf[1] = nullptr; }
        void* p = nullptr;
        if (hipMalloc(&p, 2 * dm_align_up(need)) != hipSuccess) { e.ptr = nullptr; return nullptr; }
        e.buf[0] = (double*)p;
        e.buf[1] = (double*)((char*)p + dm_align_up(need));
    } else {
        e.buf[1] = (double*)((char*)e.buf[0] + dm_align_up(need));
    }
    e.ptr = ptr; e.B = B; e.N = N; e.k = k; e.ld = ld; e.esz = esz; e.n = n; e.cur = 0; e.valid = false;
    return &e;
}

int dm_ws_reserve(dm_ctx* ctx, size_t total_bytes) {
    ctx->ws_off = 0;
    // (the queue counters dm_last_requeued_rows reads live in the arena of the call that wrote them: a new call recycles -- or, growing,
    //  frees -- that memory, so the pointer dies here; only the call that sets it again at its end makes the diagnostic valid.  ADVICE r04)
    ctx->last_flag_counts = nullptr;
    ctx->last_flag_sets = 0;
    total_bytes = dm_align_up(total_bytes + 4096, 1 << 20);
    if (total_bytes <= ctx->ws_bytes) return DM_OK;
    DM_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->ws) {
        DM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // nothing in flight may still use the old block
        DM_CHECK_HIP(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    void* p = nullptr;
    if (hipMalloc(&p, total_bytes) != hipSuccess)
        return dm_fail(ctx, DM_ENOMEM, "workspace allocation of %zu bytes failed", total_bytes);
    ctx->ws = (char*)p;
    ctx->ws_bytes = total_bytes;
    return DM_OK;
}

void* dm_ws_take(dm_ctx* ctx, size_t bytes) {
    size_t off = dm_align_up(ctx->ws_off);
    if (off + bytes > ctx->ws_bytes) return nullptr;   // callers reserve exactly what they take
    ctx->ws_off = off + bytes;
    return ctx->ws + off;
}

// ---- per-kernel timing -------------------------------------------------------
extern "C" int dm_profile_kernel(dm_ctx* ctx, const char* name) {
    if (!ctx) return DM_EINVAL;
    DM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->prof_name = name ? name : "";
    ctx->prof_used = 0;
    return DM_OK;
}

int dm_prof_begin(dm_ctx* ctx, const char* name) {
    if (ctx->prof_name.empty() || (ctx->prof_name != "*" && ctx->prof_name != name)) return -1;
    if (ctx->prof_used + 2 > ctx->prof_events.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return -1;
            ctx->prof_events.push_back(e);
        }
    }
    int tok = (int)ctx->prof_used;
    if (ctx->prof_names.size() < ctx->prof_events.size() / 2) ctx->prof_names.resize(ctx->prof_events.size() / 2);
    ctx->prof_names[tok / 2] = name;
    (void)hipEventRecord(ctx->prof_events[tok], ctx->stream);
    ctx->prof_used += 2;
    return tok;
}

int dm_prof_end(dm_ctx* ctx, int token) {
    if (token < 0) return 0;
    (void)hipEventRecord(ctx->prof_events[token + 1], ctx->stream);
    return 0;
}

extern "C" int dm_profile_read(dm_ctx* ctx, int* launches, double* total_ms) {
    if (!ctx || !launches || !total_ms) return DM_EINVAL;
    DM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    int n = 0;
    for (size_t i = 0; i + 1 < ctx->prof_used; i += 2) {
        float ms = 0.f;
        DM_CHECK_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_events[i], ctx->prof_events[i + 1]));
        tot += ms;
        ++n;
    }
    *launches = n;
    *total_ms = tot;
    ctx->prof_used = 0;
    return DM_OK;
}

// every launch since dm_profile_kernel(ctx, "*") (or the one named kernel), aggregated by name in order of first launch
extern "C" int dm_profile_report(dm_ctx* ctx, char* buf, size_t cap) {
    if (!ctx || !buf || cap == 0) return DM_EINVAL;
    DM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<std::string> names;
    std::vector<int> count;
    std::vector<double> total;
    for (size_t i = 0; i + 1 < ctx->prof_used; i += 2) {
        float ms = 0.f;
        DM_CHECK_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_events[i], ctx->prof_events[i + 1]));
        const std::string n = ctx->prof_names[i / 2] ? ctx->prof_names[i / 2] : "?";
        size_t q = 0;
        while (q < names.size() && names[q] != n) ++q;
        if (q == names.size()) { names.push_back(n); count.push_back(0); total.push_back(0.0); }
        count[q] += 1;
        total[q] += ms;
    }
    std::string out;
    for (size_t q = 0; q < names.size(); ++q) {
        char line[256];
        snprintf(line, sizeof(line), "%s\t%d\t%.6f\n", names[q].c_str(), count[q], total[q]);
        out += line;
    }
    if (out.size() + 1 > cap) return dm_fail(ctx, DM_EINVAL, "dm_profile_report: buffer of %zu bytes too small (%zu needed)", cap, out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    ctx->prof_used = 0;
    return DM_OK;
}

int dm_grant_lds(dm_ctx* ctx, const void* func, size_t bytes) {
    size_t& have = ctx->lds_granted[func];
    if (bytes > have) {
        DM_CHECK_HIP(ctx, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    return DM_OK;
}

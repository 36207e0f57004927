// svx_api.hip -- the C ABI (include/svx.h): context, scratch arena, per-op entry points and the
// fused batch pipeline that restates dp_utils.vecalign() (svecalign/vecalign/dp_utils.py:381-537)
// as a fixed sequence of batched kernel launches with no host round trip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "svx_common.h"

static char g_err[512] = "";

int svx_fail(svx_ctx* ctx, int code, const char* fmt, ...) {
    char* dst = ctx ? ctx->err : g_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

static const char* kStageNames[] = {"pyr0",       "pyrN",     "pyr_aux",      "knob_scoresN", "knob",      "dense_costs",
                                    "dense_dp",   "path",     "band_costs0",  "band_costsN",  "band_dp0",  "band_dpN",
                                    "traceback",  "setup",    "total",        "host_plan",    "host_launch", "knob_sort", "knob_scores0", "pyr1", "tiles"};
enum {
    S_PYR0 = 0, S_PYRN, S_PYR_AUX, S_KNOB_SCORES, S_KNOB, S_DENSE_COSTS, S_DENSE_DP, S_PATH, S_BAND_COSTS0, S_BAND_COSTSN,
    S_BAND_DP0, S_BAND_DPN, S_TRACEBACK, S_SETUP, S_TOTAL, S_HOST_PLAN, S_HOST_LAUNCH, S_KNOB_SORT, S_KNOB_SCORES0, S_PYR1, S_TILES, S_COUNT
};

struct StageRec {
    int stage;
    hipEvent_t a, b;
};

struct svx_ctx_ext : svx_ctx {
    double ms[S_COUNT];
    int launches[S_COUNT];
    std::vector<StageRec> recs;
    // profiling mode 2: the event pairs of every call wait here, unread, until a stage time is asked for
    std::vector<StageRec> pending;
    double acc_ms[S_COUNT];
    long long acc_launches[S_COUNT];
    std::vector<SvxPairDev> host;  // descriptors of the last batch (kept alive for the async upload)
    // sub-batches of one svx_align_batch call run on separate streams so that the latency-bound serial
    // kernels (DP, traceback) of one sub-batch overlap the streaming kernels of the other
    int n_streams;
    hipStream_t aux[3];
    hipEvent_t fork_ev, join_ev[3], stag_ev[3];
    bool aux_ready;
    // side stream: small latency-bound kernels that do not depend on the streaming passes (the sample sort)
    hipStream_t side;
    hipEvent_t side_fork, side_join;
    bool side_ready;
    int last_ntypes, last_band;  // of the last svx_align_batch call (svx_debug_level)
};

static inline svx_ctx_ext* X(svx_ctx* c) { return static_cast<svx_ctx_ext*>(c); }

struct StageScope {
    svx_ctx_ext* c;
    StageRec r;
    bool on;
    StageScope(svx_ctx* ctx, int stage) : c(X(ctx)), on(ctx->profiling != 0) {
        r.stage = stage;
        if (on) {
            (void)hipEventCreate(&r.a);
            (void)hipEventCreate(&r.b);
            (void)hipEventRecord(r.a, c->stream);
        }
    }
    ~StageScope() {
        if (on) {
            (void)hipEventRecord(r.b, c->stream);
            c->recs.push_back(r);
        }
        c->launches[r.stage]++;
    }
};

static void drop_pending(svx_ctx_ext* c) {
    for (auto& r : c->pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    c->pending.clear();
}

extern "C" {

const char* svx_version(void) { return "svx 0.1 (gfx950)"; }

int svx_create(int device_id, svx_ctx** out) {
    if (!out) return svx_fail(nullptr, SVX_ERR_ARG, "svx_create: out is NULL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return svx_fail(nullptr, SVX_ERR_HIP, "svx_create: no HIP device (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev) return svx_fail(nullptr, SVX_ERR_ARG, "svx_create: device %d of %d", device_id, ndev);
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return svx_fail(nullptr, SVX_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    svx_ctx_ext* c = new svx_ctx_ext();
    c->device = device_id;
    c->stream = nullptr;
    c->arena = nullptr;
    c->arena_bytes = 0;
    c->arena_used = 0;
    c->err[0] = 0;
    c->profiling = 0;
    for (int i = 0; i < S_COUNT; i++) { c->ms[i] = -1.0; c->launches[i] = 0; }
    c->n_streams = 1;
    c->last_ntypes = 0;
    c->last_band = 0;
    c->aux_ready = false;
    c->side_ready = false;
    *out = c;
    return SVX_OK;
}

int svx_destroy(svx_ctx* ctx) {
    if (!ctx) return SVX_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->arena) (void)hipFree(ctx->arena);
    drop_pending(X(ctx));
    if (X(ctx)->side_ready) {
        (void)hipStreamDestroy(X(ctx)->side);
        (void)hipEventDestroy(X(ctx)->side_fork);
        (void)hipEventDestroy(X(ctx)->side_join);
    }
    if (X(ctx)->aux_ready) {
        for (int i = 0; i < 3; i++) { (void)hipStreamDestroy(X(ctx)->aux[i]); (void)hipEventDestroy(X(ctx)->join_ev[i]); (void)hipEventDestroy(X(ctx)->stag_ev[i]); }
        (void)hipEventDestroy(X(ctx)->fork_ev);
    }
    delete X(ctx);
    return SVX_OK;
}

int svx_set_stream(svx_ctx* ctx, void* hip_stream) {
    if (!ctx) return SVX_ERR_ARG;
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return SVX_OK;
}

int svx_synchronize(svx_ctx* ctx) {
    if (!ctx) return SVX_ERR_ARG;
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVX_OK;
}

const char* svx_last_error(const svx_ctx* ctx) { return ctx ? ctx->err : g_err; }

int64_t svx_scratch_bytes(const svx_ctx* ctx) { return ctx ? (int64_t)ctx->arena_bytes : 0; }

int svx_set_streams(svx_ctx* ctx, int n) {
    if (!ctx || n < 1 || n > 4) return SVX_ERR_ARG;
    X(ctx)->n_streams = n;
    return SVX_OK;
}

// mode 2: fold the unread event pairs of the calls so far into the running totals (one stream synchronisation)
static void fold_pending(svx_ctx_ext* c) {
    if (c->pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto& r : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c->acc_ms[r.stage] += ms; }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    c->pending.clear();
}

int svx_set_profiling(svx_ctx* ctx, int on) {
    if (!ctx) return SVX_ERR_ARG;
    svx_ctx_ext* c = X(ctx);
    drop_pending(c);
    ctx->profiling = on;
    if (on == 2)
        for (int i = 0; i < S_COUNT; i++) { c->acc_ms[i] = 0.0; c->acc_launches[i] = 0; }
    return SVX_OK;
}

static int stage_index(const char* name) {
    for (int i = 0; i < S_COUNT; i++)
        if (strcmp(name, kStageNames[i]) == 0) return i;
    return -1;
}

double svx_stage_ms(svx_ctx* ctx, const char* stage) {
    if (!ctx || !stage) return -1.0;
    int i = stage_index(stage);
    if (i < 0) return -1.0;
    if (ctx->profiling == 2) {
        fold_pending(X(ctx));
        return X(ctx)->acc_ms[i];
    }
    return X(ctx)->ms[i];
}

int svx_stage_launches(svx_ctx* ctx, const char* stage) {
    if (!ctx || !stage) return -1;
    int i = stage_index(stage);
    if (i < 0) return -1;
    if (ctx->profiling == 2) return (int)X(ctx)->acc_launches[i];
    return X(ctx)->launches[i];
}

}  // extern "C"

// ------------------------------------------------------------------------------ arena
static int arena_reserve(svx_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->arena_bytes) return SVX_OK;
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->arena) SVX_HIP(ctx, hipFree(ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_bytes = 0;
    size_t want = bytes + bytes / 16 + (1 << 20);
    hipError_t e = hipMalloc(&ctx->arena, want);
    if (e != hipSuccess) return svx_fail(ctx, SVX_ERR_NOMEM, "scratch arena: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    ctx->arena_bytes = want;
    return SVX_OK;
}

struct Bump;
template <class T>
static inline void set_off(T*& p, Bump& b, size_t bytes);

struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

template <class T>
static inline void set_off(T*& p, Bump& b, size_t bytes) {
    p = reinterpret_cast<T*>(b.take(bytes) + 1);
}

static int make_types(svx_ctx* ctx, const int32_t* types, int T, SvxTypes* out) {
    if (T < 0 || T > SVX_MAX_TYPES) return svx_fail(ctx, SVX_ERR_ARG, "%d alignment types (max %d)", T, SVX_MAX_TYPES);
    out->n = T;
    out->maxstep = 1;
    for (int t = 0; t < T; t++) {
        int x = types[2 * t], y = types[2 * t + 1];
        if (x < 1 || y < 1 || x > 100 || y > 100) return svx_fail(ctx, SVX_ERR_ARG, "alignment type (%d,%d): sizes must be >= 1", x, y);
        out->x[t] = (int8_t)x;
        out->y[t] = (int8_t)y;
        if (x + y > out->maxstep) out->maxstep = x + y;
    }
    out->x[T] = 0; out->y[T] = 1;
    out->x[T + 1] = 1; out->y[T + 1] = 0;
    return SVX_OK;
}

static int check_dim(svx_ctx* ctx, int d) {
    if (d <= 0 || d % 8 != 0 || d > SVX_MAX_DIM)
        return svx_fail(ctx, SVX_ERR_ARG, "embedding dimension %d: must be a positive multiple of 8, at most %d", d, SVX_MAX_DIM);
    return SVX_OK;
}

#define NEED(ctx, cond, ...) \
    do { if (!(cond)) return svx_fail(ctx, SVX_ERR_ARG, __VA_ARGS__); } while (0)

extern "C" {

// ------------------------------------------------------------------------------ per-op entry points
int svx_dense_costs(svx_ctx* ctx, const float* vecs0, int k0, int s0, const float* vecs1, int k1, int s1, int d,
                    const float* norm0, const float* norm1, int offset0, int offset1, float* costs) {
    NEED(ctx, ctx && vecs0 && vecs1 && norm0 && norm1 && costs, "svx_dense_costs: null argument");
    NEED(ctx, k0 > offset0 && k1 > offset1 && offset0 >= 0 && offset1 >= 0, "svx_dense_costs: offsets (%d,%d) outside %d/%d layers", offset0, offset1, k0, k1);
    int rc = check_dim(ctx, d);
    if (rc) return rc;
    return svxl_dense_costs(ctx, vecs0 + (size_t)offset0 * s0 * d, s0, vecs1 + (size_t)offset1 * s1 * d, s1, d,
                            norm0 + (size_t)offset0 * s0, norm1 + (size_t)offset1 * s1, offset0 + 1, offset1 + 1, costs);
}

int svx_dense_dp(svx_ctx* ctx, const float* cost, int s0, int s1, float pen, double* csum, int32_t* bp) {
    NEED(ctx, ctx && bp && (cost || s0 == 0 || s1 == 0), "svx_dense_dp: null argument");
    NEED(ctx, s0 >= 0 && s1 >= 0, "svx_dense_dp: negative size");
    return svxl_dense_dp(ctx, cost, s0, s1, pen, csum, bp);
}

int svx_score_path(svx_ctx* ctx, const int32_t* xx, const int32_t* yy, int64_t n, const float* norm1, const float* norm2,
                   const float* vecs1, int rows1, const float* vecs2, int rows2, int d, float* out) {
    NEED(ctx, ctx && xx && yy && norm1 && norm2 && vecs1 && vecs2 && out, "svx_score_path: null argument");
    NEED(ctx, rows1 > 0 && rows2 > 0, "svx_score_path: empty vecs");
    int rc = check_dim(ctx, d);
    if (rc) return rc;
    return svxl_score_path(ctx, xx, yy, n, norm1, norm2, vecs1, rows1, vecs2, rows2, d, out);
}

int svx_sparse_costs(svx_ctx* ctx, const float* vecs0, int k0, int xsize, const float* vecs1, int k1, int ysize, int d,
                     const float* norms0, const float* norms1, const int32_t* path, int A, const int32_t* types_host, int T,
                     int width_over2, float* costs, int32_t* b_offset) {
    NEED(ctx, ctx && vecs0 && vecs1 && norms0 && norms1 && path && b_offset && (costs || T == 0), "svx_sparse_costs: null argument");
    NEED(ctx, width_over2 >= 1 && A >= 0, "svx_sparse_costs: bad width/path length");
    int rc = check_dim(ctx, d);
    if (rc) return rc;
    SvxTypes ty;
    rc = make_types(ctx, types_host, T, &ty);
    if (rc) return rc;
    int mx = 0, my = 0;
    for (int t = 0; t < T; t++) { if (ty.x[t] > mx) mx = ty.x[t]; if (ty.y[t] > my) my = ty.y[t]; }
    if (mx > k0) return svx_fail(ctx, SVX_ERR_OVERLAPS, "%d x overlaps requrested (via alignment_types), but vecs0 only has %d", mx, k0);
    if (my > k1) return svx_fail(ctx, SVX_ERR_OVERLAPS, "%d y overlaps requrested (via alignment_types), but vecs1 only has %d", my, k1);
    rc = arena_reserve(ctx, 256);
    if (rc) return rc;
    int* status = reinterpret_cast<int*>(ctx->arena);
    SVX_HIP(ctx, hipMemsetAsync(status, 0, sizeof(int), ctx->stream));
    rc = svxl_band_costs(ctx, vecs0, k0, xsize, vecs1, k1, ysize, d, SVX_F32, nullptr, nullptr, norms0, norms1, path, A, ty,
                         width_over2, costs, b_offset, status);
    if (rc) return rc;
    int hs = 0;
    SVX_HIP(ctx, hipMemcpyAsync(&hs, status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hs != 0) return svx_fail(ctx, hs, "search path is not a unit-step lattice path starting at (0,0)");
    return SVX_OK;
}

int svx_sparse_dp(svx_ctx* ctx, const float* costs, const int32_t* b_offset_in, int A, int B, const int32_t* types_host,
                  int T, double del_penalty, int x_in_size, int y_in_size, double* csum, int32_t* xp, int32_t* yp,
                  int32_t* b_offset_out) {
    NEED(ctx, ctx && b_offset_in && csum && xp && yp && b_offset_out && (costs || T == 0), "svx_sparse_dp: null argument");
    NEED(ctx, A >= 1 && B >= 1, "svx_sparse_dp: empty cost band");
    SvxTypes ty;
    int rc = make_types(ctx, types_host, T, &ty);
    if (rc) return rc;
    return svxl_sparse_dp(ctx, costs, b_offset_in, A, B, ty, del_penalty, x_in_size, y_in_size, csum, xp, yp, b_offset_out);
}

int svx_make_norm1(svx_ctx* ctx, float* vecs, int64_t rows, int d) {
    NEED(ctx, ctx && (vecs || rows == 0), "svx_make_norm1: null argument");
    int rc = check_dim(ctx, d);
    if (rc) return rc;
    return svxl_make_norm1(ctx, vecs, rows, d);
}

int svx_downsample(svx_ctx* ctx, const float* vecs, int k, int n, int d, float* half) {
    NEED(ctx, ctx && vecs && (half || n < 2), "svx_downsample: null argument");
    int rc = check_dim(ctx, d);
    if (rc) return rc;
    const int h = n / 2;
    if (h <= 0 || k <= 0) return SVX_OK;
    const int nblk = (h + SVX_PYR_SLOTS - 1) / SVX_PYR_SLOTS;
    Bump b;
    size_t o_part = b.take((size_t)k * nblk * d * sizeof(float));
    size_t o_mean = b.take((size_t)k * d * sizeof(float));
    rc = arena_reserve(ctx, b.off);
    if (rc) return rc;
    float* part = reinterpret_cast<float*>(ctx->arena + o_part);
    float* mean = reinterpret_cast<float*>(ctx->arena + o_mean);
    if ((rc = svxl_pairsum(ctx, vecs, k, n, d, half, part, nblk))) return rc;
    if ((rc = svxl_colmean_plain(ctx, part, k, nblk, d, h, mean))) return rc;
    return svxl_sub_mean(ctx, half, k, h, d, mean);
}

int svx_compute_norms(svx_ctx* ctx, const float* vecs0, int k0, int n0, const float* vecs1, int k1, int n1, int d,
                      const int32_t* idx, int samples_per_overlap, float* norms0) {
    NEED(ctx, ctx && vecs0 && vecs1 && idx && norms0, "svx_compute_norms: null argument");
    NEED(ctx, n1 > 0 && samples_per_overlap > 0 && k1 > 0, "svx_compute_norms: nothing to sample");
    int rc = check_dim(ctx, d);
    if (rc) return rc;
    rc = arena_reserve(ctx, (size_t)d * sizeof(float) + 256);
    if (rc) return rc;
    float* rbar = reinterpret_cast<float*>(ctx->arena);
    if ((rc = svxl_sample_mean_plain(ctx, vecs1, k1, n1, d, idx, samples_per_overlap, rbar))) return rc;
    return svxl_norms_from_rbar(ctx, vecs0, (int64_t)k0 * n0, d, rbar, norms0);
}

int svx_del_penalty(svx_ctx* ctx, const float* scores, int64_t n, double frac, double* del_penalty) {
    NEED(ctx, ctx && scores && del_penalty && n > 0, "svx_del_penalty: null/empty argument");
    return svxl_del_penalty(ctx, scores, n, frac, del_penalty);
}

int svx_dense_traceback(svx_ctx* ctx, const int32_t* bp, int s0, int s1, int32_t* align, int32_t* count) {
    NEED(ctx, ctx && bp && align && count, "svx_dense_traceback: null argument");
    return svxl_dense_traceback(ctx, bp, s0, s1, align, count);
}

int svx_sparse_traceback(svx_ctx* ctx, const double* csum, const int32_t* xp, const int32_t* yp, const int32_t* b_offset_out,
                         int a_out, int B, int xsize, int ysize, int32_t* align, double* scores, int32_t* count) {
    NEED(ctx, ctx && csum && xp && yp && b_offset_out && align && scores && count, "svx_sparse_traceback: null argument");
    return svxl_sparse_traceback(ctx, csum, xp, yp, b_offset_out, a_out, B, xsize, ysize, align, scores, count);
}

int svx_search_path(svx_ctx* ctx, const int32_t* align, const int32_t* n_align, int upsample, int size0, int size1,
                    int32_t* path, int32_t* path_len) {
    NEED(ctx, ctx && align && n_align && path && path_len, "svx_search_path: null argument");
    return svxl_search_path(ctx, align, n_align, upsample, size0, size1, path, size0 + size1 + 4, path_len);
}

int svx_gather_rows(svx_ctx* ctx, const void* table, int64_t n_rows, int d, int dtype, const int32_t* idx, int64_t n_out, void* out) {
    NEED(ctx, ctx && table && idx && out, "svx_gather_rows: null argument");
    NEED(ctx, dtype == SVX_F32 || dtype == SVX_F16 || dtype == SVX_BF16, "svx_gather_rows: unknown dtype %d", dtype);
    int rc = check_dim(ctx, d);
    if (rc) return rc;
    return svxl_gather_rows(ctx, table, n_rows, d * (dtype == SVX_F32 ? 4 : 2), dtype, idx, n_out, out);
}

int svx_num_levels(int n, int m, int max_size_full_dp) {
    long long s0 = n, s1 = m, lim = (long long)max_size_full_dp * max_size_full_dp;
    int depth = 0;
    while (s0 * s1 > lim) {
        depth++;
        s0 /= 2;
        s1 /= 2;
    }
    return depth;
}

int64_t svx_knob_count(int n_l, int m_l, int costs_sample_size) {
    long long p = (long long)n_l * m_l;
    return p < costs_sample_size ? p : costs_sample_size;
}

}  // extern "C"

// ------------------------------------------------------------------------------ fused batch
__global__ void k_init_batch(const SvxPairDev* pairs) {
    const SvxPairDev& P = pairs[blockIdx.x];
    if (threadIdx.x != 0) return;
    *P.status = 0;
    for (int l = 0; l <= P.L; l++) {
        if (P.lev[l].n_align) *P.lev[l].n_align = 0;
        if (P.lev[l].path_len) *P.lev[l].path_len = 0;
    }
    if (P.straight) {
        // the straight search path is the slant of ONE alignment block covering both documents (dp_utils.py:177-225)
        int* r = P.lev[0].align;
        r[0] = 0; r[1] = P.lev[0].n[0]; r[2] = 0; r[3] = P.lev[0].n[1];
        *P.lev[0].n_align = 1;
    }
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

extern "C" int svx_copy_to_host(svx_ctx* ctx, void* dst_host, const void* src_device, int64_t bytes) {
    NEED(ctx, ctx && (bytes == 0 || (dst_host && src_device)) && bytes >= 0, "svx_copy_to_host: bad argument");
    if (bytes == 0) return SVX_OK;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    SVX_HIP(ctx, hipMemcpyAsync(dst_host, src_device, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVX_OK;
}

extern "C" int svx_debug_level(svx_ctx* ctx, int pair, int level, svx_level_view* out) {
    NEED(ctx, ctx && out, "svx_debug_level: null argument");
    svx_ctx_ext* cx = X(ctx);
    NEED(ctx, pair >= 0 && pair < (int)cx->host.size(), "svx_debug_level: pair %d of the last batch (%d pairs)", pair, (int)cx->host.size());
    const SvxPairDev& P = cx->host[pair];
    NEED(ctx, level >= 0 && level <= P.L, "svx_debug_level: level %d (the pair has %d)", level, P.L + 1);
    const SvxLevel& Lv = P.lev[level];
    memset(out, 0, sizeof(*out));
    out->size0 = Lv.n[0]; out->size1 = Lv.n[1];
    out->k0 = P.K[0]; out->k1 = P.K[1];
    out->n_types = level == 0 ? cx->last_ntypes : 1;
    out->band = cx->last_band;
    out->n0 = Lv.nrm[0]; out->n1 = Lv.nrm[1];
    out->del_penalty = Lv.pen;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (Lv.n_align) SVX_HIP(ctx, hipMemcpy(&out->n_align, Lv.n_align, sizeof(int), hipMemcpyDeviceToHost));
    out->alignments = Lv.align;
    const bool refined = (level < P.L) || (P.L == 0);
    if (refined) {
        SVX_HIP(ctx, hipMemcpy(&out->path_len, Lv.path_len, sizeof(int), hipMemcpyDeviceToHost));
        out->searchpath = Lv.path;
        out->a_b_costs = Lv.costs;
        out->b_offset = Lv.boff;
        out->a_b_csum = Lv.csum;
        out->a_b_bp = Lv.bpk;
        out->a_b_xp = Lv.xp;
        out->a_b_yp = Lv.yp;
        out->new_b_offset = Lv.boff_out;
        out->alignment_scores = Lv.scores;
    }
    if (level == P.L && !P.straight) {  // the dense stage (dp_utils.py:465-473)
        out->costs_1to1 = P.dcost;
        out->x_y_tb_diag = P.dbp;
    }
    out->knob_scores = Lv.kscore;
    out->n_knob = Lv.kn;
    if (level >= 1) {
        out->v0_l0 = Lv.P[0];
        out->v1_l0 = Lv.P[1];
    }
    return SVX_OK;
}

extern "C" int svx_align_batch(svx_ctx* ctx, const svx_align_params* prm, const svx_pair* pairs, int n_pairs) {
    NEED(ctx, ctx && prm && (pairs || n_pairs == 0), "svx_align_batch: null argument");
    if (n_pairs <= 0) return SVX_OK;
    const auto t_enter = std::chrono::steady_clock::now();
    svx_ctx_ext* cx = X(ctx);
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = check_dim(ctx, prm->d);
    if (rc) return rc;
    const int d = prm->d, dtype = prm->dtype;
    NEED(ctx, dtype == SVX_F32 || dtype == SVX_F16 || dtype == SVX_BF16, "svx_align_batch: unknown dtype %d", dtype);
    NEED(ctx, prm->max_size_full_dp >= 1, "svx_align_batch: max_size_full_dp must be >= 1");
    NEED(ctx, prm->costs_sample_size >= 1, "svx_align_batch: costs_sample_size must be >= 1");
    NEED(ctx, prm->num_samps_for_norm >= 0, "svx_align_batch: num_samps_for_norm must be >= 0");
    SvxTypes tfinal, t11;
    if ((rc = make_types(ctx, prm->types, prm->n_types, &tfinal))) return rc;
    const int32_t one_one[2] = {1, 1};
    make_types(ctx, one_one, 1, &t11);
    int W = prm->width_over2 < 3 ? 3 : prm->width_over2;  // dp_utils.py:391-393
    const int B = 2 * W;
    int mx = 0, my = 0;
    for (int t = 0; t < tfinal.n; t++) { if (tfinal.x[t] > mx) mx = tfinal.x[t]; if (tfinal.y[t] > my) my = tfinal.y[t]; }
    const bool packable = mx <= 15 && my <= 15;  // back-pointers fit 4 bits each
    NEED(ctx, prm->search_mode == SVX_SEARCH_COARSE_TO_FINE || prm->search_mode == SVX_SEARCH_STRAIGHT, "svx_align_batch: unknown search_mode %d", prm->search_mode);
    const bool straight = prm->search_mode == SVX_SEARCH_STRAIGHT;
    cx->last_ntypes = tfinal.n;
    cx->last_band = B;
    // straight search with a band wider than the one-workgroup DP kernel takes: wavefront of tiles (svx_tiles.hip)
    const bool tiles = straight && B > 64;
    if (tiles) {
        NEED(ctx, packable && tfinal.n >= 1 && svxl_band_tiles_ok(tfinal),
             "wide straight band: the tile kernel takes 1..16 alignment types on <= 12 overlap layers with steps <= 8");
    }

    // ---- plan: level sizes, scratch layout, launch extents
    std::vector<SvxPairDev>& host = cx->host;
    host.assign(n_pairs, SvxPairDev());
    Bump bump;
    const size_t o_desc = bump.take((size_t)n_pairs * sizeof(SvxPairDev));
    size_t o_tickets = 0;  // wide straight bands: ticket table [tile anti-diagonals x pairs + 1] and the ticket counter (below)
    int maxL = 0, max_ksum = 0, max_kn = 0, max_ds0 = 0, max_ds1 = 0, max_n0 = 0;
    int max_nblk[SVX_MAX_LEVELS] = {0}, max_A[SVX_MAX_LEVELS] = {0};
    bool any_L0 = false;
#define OFF(ptr_field, bytes) set_off(ptr_field, bump, bytes)
    // offsets are stored +1 so that a null pointer stays distinguishable; patched below
    for (int p = 0; p < n_pairs; p++) {
        const svx_pair& in = pairs[p];
        SvxPairDev& P = host[p];
        memset(&P, 0, sizeof(P));
        NEED(ctx, in.vecs0 && in.vecs1 && in.align && in.scores && in.info && in.knob_idx, "pair %d: null pointer", p);
        NEED(ctx, in.n >= 1 && in.m >= 1 && in.k0 >= 1 && in.k1 >= 1, "pair %d: empty document or no overlap layers", p);
        if (mx > in.k0) return svx_fail(ctx, SVX_ERR_OVERLAPS, "%d x overlaps requrested (via alignment_types), but vecs0 only has %d", mx, in.k0);
        if (my > in.k1) return svx_fail(ctx, SVX_ERR_OVERLAPS, "%d y overlaps requrested (via alignment_types), but vecs1 only has %d", my, in.k1);
        const int L = straight ? 0 : svx_num_levels(in.n, in.m, prm->max_size_full_dp);
        P.straight = straight ? 1 : 0;
        NEED(ctx, L < SVX_MAX_LEVELS, "pair %d: %d pyramid levels (max %d)", p, L + 1, SVX_MAX_LEVELS);
        P.v[0] = in.vecs0; P.v[1] = in.vecs1;
        P.K[0] = in.k0; P.K[1] = in.k1;
        P.L = L; P.d = d;
        P.norm_override[0] = in.norms0 != nullptr;
        P.norm_override[1] = in.norms1 != nullptr;
        P.status = in.info + 1;
        if (L > maxL) maxL = L;
        if (in.n > max_n0) max_n0 = in.n;
        if (L == 0) any_L0 = true;
        if (in.k0 + in.k1 > max_ksum) max_ksum = in.k0 + in.k1;
        const int S_from[2] = {in.k0 > 0 ? ceil_div(prm->num_samps_for_norm, in.k0) : 0,
                               in.k1 > 0 ? ceil_div(prm->num_samps_for_norm, in.k1) : 0};
        const int32_t* nidx = in.norm_idx;
        const int32_t* kidx = in.knob_idx;
        for (int l = 0; l <= L; l++) {
            SvxLevel& Lv = P.lev[l];
            const int K[2] = {in.k0, in.k1};
            for (int s = 0; s < 2; s++) {
                Lv.n[s] = (s == 0 ? in.n : in.m) >> l;
                Lv.nblk[s] = ceil_div((Lv.n[s] + 1) / 2, SVX_PYR_SLOTS);
                if (Lv.nblk[s] > max_nblk[l]) max_nblk[l] = Lv.nblk[s];
            }
            // sampled indices in the reference's draw order: into side 1 (for n0), then into side 0 (for n1)
            for (int s = 1; s >= 0; s--) {
                const bool skip = (l == 0) && P.norm_override[1 - s];  // the side whose norms use these samples
                Lv.S[s] = skip ? 0 : S_from[s];
                Lv.sidx[s] = nullptr;
                if (Lv.S[s] > 0) {
                    NEED(ctx, nidx != nullptr, "pair %d: norm_idx is NULL", p);
                    Lv.sidx[s] = nidx;
                    nidx += (size_t)K[s] * Lv.S[s];
                }
            }
            for (int s = 0; s < 2; s++) {
                if (l >= 1) {
                    Lv.npart[s] = P.lev[l - 1].nblk[s];
                    // level 1 keeps only its normalised layer 0: its rows are re-formed from the inputs (k_pyramid MODE 2)
                    OFF(Lv.P[s], (size_t)(l == 1 ? 1 : K[s]) * Lv.n[s] * d * sizeof(float));
                    OFF(Lv.part[s], (size_t)K[s] * Lv.npart[s] * d * sizeof(float));
                    OFF(Lv.mean[s], (size_t)K[s] * d * sizeof(float));
                } else {
                    OFF(Lv.inv[s], (size_t)K[s] * Lv.n[s] * sizeof(float));
                }
                OFF(Lv.rbar[s], (size_t)d * sizeof(float));
                if (l == 0 && P.norm_override[s]) Lv.nrm[s] = const_cast<float*>(s == 0 ? in.norms0 : in.norms1);
                else OFF(Lv.nrm[s], (size_t)K[s] * Lv.n[s] * sizeof(float));
            }
            const int64_t kn = svx_knob_count(Lv.n[0], Lv.n[1], prm->costs_sample_size);
            Lv.kn = (int)kn;
            Lv.kx = kidx;
            Lv.ky = kidx + kn;
            kidx += 2 * kn;
            if (Lv.kn > max_kn) max_kn = Lv.kn;
            OFF(Lv.kscore, (size_t)kn * sizeof(float));
            OFF(Lv.korder, (size_t)kn * sizeof(int));
            OFF(Lv.kys, (size_t)kn * sizeof(int));
            OFF(Lv.kstart, (size_t)(Lv.n[0] + 1) * sizeof(int));
            if (in.del_pen) Lv.pen = in.del_pen + l;
            else OFF(Lv.pen, sizeof(double));
            const bool refined = (l < L) || (L == 0);
            const int rows_cap = Lv.n[0] + Lv.n[1] + 2;
            if (refined) {
                const int cap = Lv.n[0] + Lv.n[1] + 4;
                const int T = (l == 0) ? tfinal.n : 1;
                Lv.path_cap = cap;
                if (cap > max_A[l]) max_A[l] = cap;
                OFF(Lv.path, (size_t)cap * 2 * sizeof(int));
                OFF(Lv.path_len, sizeof(int));
                OFF(Lv.cstart, (size_t)(cap / 16 + 3) * sizeof(int));  // a chunk holds at least 17 path points
                OFF(Lv.nchunks, sizeof(int));
                if (!tiles) OFF(Lv.costs, (size_t)(T > 0 ? T : 1) * cap * B * sizeof(float));  // (the tile sweep keeps costs in LDS)
                OFF(Lv.boff, (size_t)cap * sizeof(int));
                OFF(Lv.csum, (size_t)(cap + 2) * B * sizeof(double));
                if (packable) {
                    OFF(Lv.bpk, (size_t)(cap + 2) * B);
                } else {
                    OFF(Lv.xp, (size_t)(cap + 2) * B * sizeof(int));
                    OFF(Lv.yp, (size_t)(cap + 2) * B * sizeof(int));
                }
                OFF(Lv.boff_out, (size_t)(cap + 2) * sizeof(int));
            }
            if (l == 0) {
                Lv.align = in.align;
                Lv.scores = in.scores;
                Lv.n_align = in.info;
            } else {
                OFF(Lv.align, (size_t)rows_cap * 4 * sizeof(int));
                OFF(Lv.scores, (size_t)rows_cap * sizeof(double));
                OFF(Lv.n_align, sizeof(int));
            }
        }
        const SvxLevel& top = P.lev[L];
        if (!straight) {
            if (top.n[0] > max_ds0) max_ds0 = top.n[0];
            if (top.n[1] > max_ds1) max_ds1 = top.n[1];
            OFF(P.dcost, (size_t)top.n[0] * top.n[1] * sizeof(float));
            OFF(P.ddot, (size_t)top.n[0] * top.n[1] * sizeof(float));
            OFF(P.dbp, (size_t)(top.n[0] + top.n[1] + 1) * (top.n[0] + 1) * sizeof(int));  // (anti-diagonal, row) layout
        }
        if (tiles) {
            const long long TI = in.n / 32 + 1, TJ = in.m / 32 + 1;  // tiles of 32 x 32 nodes over (n + 1) x (m + 1)
            P.t_nd = (int)(TI + TJ - 1);
            long long tcap = (long long)P.t_nd * ((B + 64) / 32 + 2);
            if (tcap > TI * TJ) tcap = TI * TJ;
            NEED(ctx, tcap < (1ll << 30), "pair %d: too many tiles", p);
            P.t_cap = (int)tcap;
            OFF(P.t_lo, (size_t)P.t_nd * sizeof(int));
            OFF(P.t_cnt, (size_t)P.t_nd * sizeof(int));
            OFF(P.t_pref, (size_t)(P.t_nd + 1) * sizeof(int));
            OFF(P.t_flag, (size_t)P.t_cap * sizeof(int));
        }
    }
#undef OFF
    int max_tnd = 0;
    for (int p = 0; p < n_pairs; p++)
        if (host[p].t_nd > max_tnd) max_tnd = host[p].t_nd;
    if (tiles) o_tickets = bump.take(((size_t)(max_tnd + 2) * n_pairs + 2) * sizeof(int));
    if ((rc = arena_reserve(ctx, bump.off))) return rc;
    // patch offsets (+1) into device pointers
    char* base = ctx->arena;
    const char* ulo = reinterpret_cast<const char*>(1);
    const char* uhi = reinterpret_cast<const char*>(bump.off + 1);
    auto patch = [&](auto& ptr) {
        const char* v = reinterpret_cast<const char*>(ptr);
        if (v >= ulo && v < uhi) ptr = reinterpret_cast<typename std::remove_reference<decltype(ptr)>::type>(base + (v - ulo));
    };
    for (int p = 0; p < n_pairs; p++) {
        SvxPairDev& P = host[p];
        const svx_pair& in = pairs[p];
        for (int l = 0; l <= P.L; l++) {
            SvxLevel& Lv = P.lev[l];
            for (int s = 0; s < 2; s++) {
                patch(Lv.P[s]); patch(Lv.part[s]); patch(Lv.mean[s]); patch(Lv.rbar[s]); patch(Lv.inv[s]);
                if (!(l == 0 && P.norm_override[s])) patch(Lv.nrm[s]);
            }
            patch(Lv.kscore); patch(Lv.korder); patch(Lv.kys); patch(Lv.kstart);
            if (!in.del_pen) patch(Lv.pen);
            patch(Lv.path); patch(Lv.path_len); patch(Lv.cstart); patch(Lv.nchunks); patch(Lv.costs); patch(Lv.boff); patch(Lv.csum); patch(Lv.xp); patch(Lv.yp); patch(Lv.bpk);
            patch(Lv.boff_out);
            if (l > 0) { patch(Lv.align); patch(Lv.scores); patch(Lv.n_align); }
        }
        patch(P.dcost);
        patch(P.ddot);
        patch(P.dbp);
        patch(P.t_lo); patch(P.t_cnt); patch(P.t_pref); patch(P.t_flag);
    }
    SvxPairDev* dpairs = reinterpret_cast<SvxPairDev*>(base + o_desc);
    hipStream_t st = ctx->stream;
    const auto t_plan = std::chrono::steady_clock::now();

    // ---- run
    for (int i = 0; i < S_COUNT; i++) { cx->ms[i] = 0.0; cx->launches[i] = 0; }
    for (auto& r : cx->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    cx->recs.clear();
    if (!cx->side_ready) {
        int prio_lo = 0, prio_hi = 0;
        SVX_HIP(ctx, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));  // (lowest, highest): the side work only fills gaps
        SVX_HIP(ctx, hipStreamCreateWithPriority(&cx->side, hipStreamNonBlocking, prio_lo));
        SVX_HIP(ctx, hipEventCreateWithFlags(&cx->side_fork, hipEventDisableTiming));
        SVX_HIP(ctx, hipEventCreateWithFlags(&cx->side_join, hipEventDisableTiming));
        cx->side_ready = true;
    }
    auto run_stages = [&](SvxPairDev* dp, int np, hipEvent_t streamed_ev) -> int {
        int rc2;
        const bool use_side = cx->n_streams <= 1;  // (sub-batch streams already overlap their stages)
        {
            // the counting sort of the sampled (x, y) pairs only needs the descriptors: it runs beside the pyramid
            hipStream_t main_stream = ctx->stream;
            if (use_side) {
                SVX_HIP(ctx, hipEventRecord(cx->side_fork, main_stream));
                SVX_HIP(ctx, hipStreamWaitEvent(cx->side, cx->side_fork, 0));
                ctx->stream = cx->side;
            }
            {
                StageScope sc(ctx, S_KNOB_SORT);
                rc2 = svxl_knob_scores(ctx, dp, np, maxL, max_kn, max_n0, dtype, d, 0);
            }
            ctx->stream = main_stream;
            if (rc2) return rc2;
            if (use_side) SVX_HIP(ctx, hipEventRecord(cx->side_join, cx->side));
        }
        for (int l = 0; l <= maxL; l++) {
            // pyramid: the streaming pass of level l (S_PYR0 / S_PYRN) and its small helpers
            {
                StageScope sc(ctx, S_PYR_AUX);
                if ((rc2 = svxl_pyramid_level(ctx, dp, np, l, dtype, d, max_nblk[l], max_ksum, 0))) return rc2;
            }
            StageScope sc(ctx, l == 0 ? S_PYR0 : (l == 1 ? S_PYR1 : S_PYRN));
            if ((rc2 = svxl_pyramid_level(ctx, dp, np, l, dtype, d, max_nblk[l], max_ksum, 1))) return rc2;
        }
        if (!straight) {
            // coarsest level: the dense 1-1 cost matrix, and with it the dot products its sampled scores need
            StageScope sc(ctx, S_DENSE_COSTS);
            if ((rc2 = svxl_dense_costs_batch(ctx, dp, np, max_ds0, max_ds1, dtype, d))) return rc2;
            if ((rc2 = svxl_knob_from_dots(ctx, dp, np, max_kn))) return rc2;
        }
        if (use_side) SVX_HIP(ctx, hipStreamWaitEvent(ctx->stream, cx->side_join, 0));
        if (maxL >= 1) {
            StageScope sc(ctx, S_KNOB_SCORES);
            if ((rc2 = svxl_knob_scores(ctx, dp, np, maxL, max_kn, max_n0, dtype, d, 1))) return rc2;
        }
        {
            StageScope sc(ctx, S_KNOB_SCORES0);
            if ((rc2 = svxl_knob_scores(ctx, dp, np, maxL, max_kn, max_n0, dtype, d, 2))) return rc2;
        }
        // (the next sub-batch starts its streaming passes here: they overlap this one's serial DP stages)
        if (streamed_ev) SVX_HIP(ctx, hipEventRecord(streamed_ev, ctx->stream));
        {
            StageScope sc(ctx, S_KNOB);
            if ((rc2 = svxl_del_penalty_batch(ctx, dp, np, maxL + 1, prm->del_percentile_frac))) return rc2;
        }
        if (!straight) {
            StageScope sc(ctx, S_DENSE_DP);
            if ((rc2 = svxl_dense_stage_batch(ctx, dp, np, max_ds0))) return rc2;
        }
        const int first_depth = maxL > 0 ? maxL - 1 : 0;
        for (int depth = first_depth; depth >= 0; depth--) {
            const SvxTypes& ty = depth == 0 ? tfinal : t11;
            int lim2 = 0, tamax2 = 0;
            const bool v2 = !tiles && svxl_band2_limits(ty, W, depth, dtype, d, &lim2, &tamax2);  // which band-cost kernel takes this level
            {
                StageScope sc(ctx, S_PATH);
                // (pairs whose source level has more alignment rows than the LDS holds take the kernel's serial path)
                const int src_rows = (maxL > 0 && !any_L0) ? max_A[depth] / 2 + 8 : max_A[depth];
                if ((rc2 = svxl_search_path_batch(ctx, dp, np, depth, max_A[depth], src_rows, v2 ? lim2 : SVX_BC_ROWS - SVX_BC_TB,
                                                  tiles ? 0 : (v2 ? tamax2 : SVX_BC_TAMAX)))) return rc2;
            }
            if (tiles) {
                StageScope sc(ctx, S_TILES);
                // (sub-batches on their own streams use disjoint slices of the ticket region)
                int* tk = reinterpret_cast<int*>(base + o_tickets) + (size_t)max_tnd * (dp - dpairs) + 2 * (dp - dpairs);
                if ((rc2 = svxl_band_tiles_batch(ctx, dp, np, ty, W, dtype, max_tnd, tk, tk + (size_t)max_tnd * np + 1))) return rc2;
            } else {
                {
                    StageScope sc(ctx, depth == 0 ? S_BAND_COSTS0 : S_BAND_COSTSN);
                    if (v2) rc2 = svxl_band_costs2_batch(ctx, dp, np, depth, max_A[depth], ty, W, dtype, d);
                    else rc2 = svxl_band_costs_batch(ctx, dp, np, depth, max_A[depth], ty, W, dtype, d);
                    if (rc2) return rc2;
                }
                {
                    StageScope sc(ctx, depth == 0 ? S_BAND_DP0 : S_BAND_DPN);
                    if ((rc2 = svxl_sparse_dp_batch(ctx, dp, np, depth, ty, B))) return rc2;
                }
            }
            {
                StageScope sc(ctx, S_TRACEBACK);
                if ((rc2 = svxl_sparse_traceback_batch(ctx, dp, np, depth, B, max_A[depth], packable ? 1 : 0))) return rc2;
            }
        }
        return SVX_OK;
    };
    {
        StageScope total(ctx, S_TOTAL);
        {
            StageScope sc(ctx, S_SETUP);
            SVX_HIP(ctx, hipMemcpyAsync(dpairs, host.data(), (size_t)n_pairs * sizeof(SvxPairDev), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_init_batch, dim3(n_pairs), dim3(64), 0, st, dpairs);
            SVX_LAUNCH_CHECK(ctx, "k_init_batch");
            if (prm->num_samps_for_norm == 0) {
                // compute_norms returns ones without samples (dp_utils.py:356-357): rbar = 0 -> 1 - 0
                for (int p = 0; p < n_pairs; p++)
                    for (int l = 0; l <= host[p].L; l++)
                        for (int s = 0; s < 2; s++)
                            SVX_HIP(ctx, hipMemsetAsync(host[p].lev[l].rbar[s], 0, (size_t)d * sizeof(float), st));
            }
        }
        const int S = cx->n_streams < n_pairs ? cx->n_streams : n_pairs;
        if (S <= 1) {
            if ((rc = run_stages(dpairs, n_pairs, nullptr))) return rc;
        } else {
            if (!cx->aux_ready) {
                for (int i = 0; i < 3; i++) {
                    SVX_HIP(ctx, hipStreamCreateWithFlags(&cx->aux[i], hipStreamNonBlocking));
                    SVX_HIP(ctx, hipEventCreateWithFlags(&cx->join_ev[i], hipEventDisableTiming));
                    SVX_HIP(ctx, hipEventCreateWithFlags(&cx->stag_ev[i], hipEventDisableTiming));
                }
                SVX_HIP(ctx, hipEventCreateWithFlags(&cx->fork_ev, hipEventDisableTiming));
                cx->aux_ready = true;
            }
            SVX_HIP(ctx, hipEventRecord(cx->fork_ev, st));
            for (int s2 = 0; s2 < S; s2++) {
                const int lo = (int)((long long)n_pairs * s2 / S), hi = (int)((long long)n_pairs * (s2 + 1) / S);
                hipStream_t ss = s2 == 0 ? st : cx->aux[s2 - 1];
                // sub-batch s2 waits until sub-batch s2-1 has finished its streaming front (pyramid + sampled scores)
                if (s2 > 0) SVX_HIP(ctx, hipStreamWaitEvent(ss, cx->fork_ev, 0));
                if (s2 > 0) SVX_HIP(ctx, hipStreamWaitEvent(ss, cx->stag_ev[s2 - 1], 0));
                ctx->stream = ss;
                rc = run_stages(dpairs + lo, hi - lo, s2 + 1 < S ? cx->stag_ev[s2] : nullptr);
                ctx->stream = st;
                if (rc) return rc;
                if (s2 > 0) {
                    SVX_HIP(ctx, hipEventRecord(cx->join_ev[s2 - 1], ss));
                    SVX_HIP(ctx, hipStreamWaitEvent(st, cx->join_ev[s2 - 1], 0));
                }
            }
        }
    }
    const auto t_done = std::chrono::steady_clock::now();
    cx->ms[S_HOST_PLAN] = std::chrono::duration<double, std::milli>(t_plan - t_enter).count();
    cx->ms[S_HOST_LAUNCH] = std::chrono::duration<double, std::milli>(t_done - t_plan).count();
    if (ctx->profiling == 2) {
        // accumulate: nothing is read and nothing waits here, the next call can be launched behind this one
        for (auto& r : cx->recs) cx->pending.push_back(r);
        cx->recs.clear();
        for (int i = 0; i < S_COUNT; i++) cx->acc_launches[i] += cx->launches[i];
        cx->acc_ms[S_HOST_PLAN] += cx->ms[S_HOST_PLAN];
        cx->acc_ms[S_HOST_LAUNCH] += cx->ms[S_HOST_LAUNCH];
    } else if (ctx->profiling) {
        SVX_HIP(ctx, hipStreamSynchronize(st));
        for (auto& r : cx->recs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) cx->ms[r.stage] += ms;  // (stages of sub-batches overlap)
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
        cx->recs.clear();
    }
    return SVX_OK;
}

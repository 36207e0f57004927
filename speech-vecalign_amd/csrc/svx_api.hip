// svx_api.hip -- the C ABI (include/svx.h): context, scratch arena, per-op entry points and the
// fused batch pipeline that restates dp_utils.vecalign() (svecalign/vecalign/dp_utils.py:381-537)
// as a fixed sequence of batched kernel launches with no host round trip.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <functional>
#include <memory>
#include <type_traits>
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
                                    "traceback",  "setup",    "total",        "host_plan",    "host_launch", "knob_sort", "knob_scores0", "pyr1", "tiles",
                                    "traceback0", "path0"};
enum {
    S_PYR0 = 0, S_PYRN, S_PYR_AUX, S_KNOB_SCORES, S_KNOB, S_DENSE_COSTS, S_DENSE_DP, S_PATH, S_BAND_COSTS0, S_BAND_COSTSN,
    S_BAND_DP0, S_BAND_DPN, S_TRACEBACK, S_SETUP, S_TOTAL, S_HOST_PLAN, S_HOST_LAUNCH, S_KNOB_SORT, S_KNOB_SCORES0, S_PYR1, S_TILES,
    S_TRACEBACK0, S_PATH0, S_COUNT
};

struct StageRec {
    int stage;
    hipEvent_t a, b;
};

// What the batch driver needs to know about one call beyond the pairs themselves.
struct BatchParams {
    SvxTypes tfinal, t11;
    int W, B, dtype, d, nsamp, n_types;
    double frac;
    bool straight, tiles, packable;
};

// One half of a batch in flight (svx_align_batch, software pipeline): its own scratch arena, descriptors and launch
// extents.  Without the pipeline slot 0 carries the whole batch.
struct HalfState {
    char* arena = nullptr;
    size_t arena_bytes = 0;
    SvxPairDev* pinned = nullptr;      // pinned host copy of the descriptors: the upload never waits for the host
    size_t pinned_cap = 0;
    std::vector<SvxPairDev> host;      // host mirror of the descriptors (svx_debug_level)
    SvxPairDev* dpairs = nullptr;
    int n_pairs = 0;
    int maxL = 0, max_ksum = 0, max_kn = 0, max_ds0 = 0, max_ds1 = 0, max_n0 = 0, max_tnd = 0;
    int max_nblk[SVX_MAX_LEVELS] = {0}, max_A[SVX_MAX_LEVELS] = {0};
    bool any_L0 = false;
    size_t o_tickets = 0;
    // inputs of the launch-order cost model (microseconds are estimated from these; accuracy is not needed)
    double b_pyr[SVX_MAX_LEVELS] = {0}, b_bc[SVX_MAX_LEVELS] = {0}, b_knobN = 0, b_knob0 = 0, dense_cells = 0;
    BatchParams bp;
    bool chain_pending = false;        // streaming front launched, refinement chain not yet
    bool chain_done_valid = false;
    hipEvent_t front_done = nullptr, chain_done = nullptr, upload_done = nullptr;
    bool upload_valid = false;
};

struct svx_ctx_ext : svx_ctx {
    double ms[S_COUNT];
    int launches[S_COUNT];
    std::vector<StageRec> recs;
    // profiling mode 2: the event pairs of every call wait here, unread, until a stage time is asked for
    std::vector<StageRec> pending;
    double acc_ms[S_COUNT];
    long long acc_launches[S_COUNT];
    // software pipeline (svx_set_pipeline): a call's batch is cut into two halves; the latency-bound refinement chain
    // of one half (deletion penalty, coarse DP, per level: search path, band DP, traceback) runs on `chain` beside the
    // streaming passes of the other half on the context's stream, and the second half's chain is left for the next
    // call (or svx_flush) to overlap
    int pipeline;
    HalfState half[2];
    int last_split;            // pairs in half[0] at the last call (svx_debug_level)
    hipStream_t chain;
    bool chain_ready;
    std::vector<hipEvent_t> ev_pool;   // events of the launch-order hand-offs, reused round robin
    size_t ev_next;
    // side stream: small latency-bound kernels that do not depend on the streaming passes (the sample sort)
    hipStream_t side;
    hipStream_t helper;        // the pyramid's small per-level helpers (high priority: a pass on the context's stream waits for them)
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
    c->pipeline = 0;
    c->last_split = 0;
    c->chain = nullptr;
    c->chain_ready = false;
    c->ev_next = 0;
    c->last_ntypes = 0;
    c->last_band = 0;
    c->side_ready = false;
    *out = c;
    return SVX_OK;
}

int svx_destroy(svx_ctx* ctx) {
    if (!ctx) return SVX_OK;
    svx_ctx_ext* c = X(ctx);
    (void)hipSetDevice(ctx->device);
    (void)svx_flush(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (c->chain_ready) (void)hipStreamSynchronize(c->chain);
    if (ctx->arena) (void)hipFree(ctx->arena);
    drop_pending(c);
    for (auto& r : c->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    if (c->side_ready) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
        (void)hipStreamSynchronize(c->helper);
        (void)hipStreamDestroy(c->helper);
        (void)hipEventDestroy(c->side_fork);
        (void)hipEventDestroy(c->side_join);
    }
    if (c->chain_ready) (void)hipStreamDestroy(c->chain);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (int h = 0; h < 2; h++) {
        HalfState& H = c->half[h];
        if (H.arena) (void)hipFree(H.arena);
        if (H.pinned) (void)hipHostFree(H.pinned);
        if (H.front_done) (void)hipEventDestroy(H.front_done);
        if (H.chain_done) (void)hipEventDestroy(H.chain_done);
        if (H.upload_done) (void)hipEventDestroy(H.upload_done);
    }
    delete c;
    return SVX_OK;
}

int svx_set_stream(svx_ctx* ctx, void* hip_stream) {
    if (!ctx) return SVX_ERR_ARG;
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return SVX_OK;
}

int svx_synchronize(svx_ctx* ctx) {
    if (!ctx) return SVX_ERR_ARG;
    int rc = svx_flush(ctx);  // (the context's stream then waits for everything the pipeline has in flight)
    if (rc) return rc;
    SVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVX_OK;
}

const char* svx_last_error(const svx_ctx* ctx) { return ctx ? ctx->err : g_err; }

int64_t svx_scratch_bytes(const svx_ctx* ctx) {
    if (!ctx) return 0;
    const svx_ctx_ext* c = static_cast<const svx_ctx_ext*>(ctx);
    return (int64_t)(ctx->arena_bytes + c->half[0].arena_bytes + c->half[1].arena_bytes);
}

int svx_set_pipeline(svx_ctx* ctx, int on) {
    if (!ctx || on < 0 || on > 1) return SVX_ERR_ARG;
    if (X(ctx)->pipeline != on) {
        int rc = svx_flush(ctx);
        if (rc) return rc;
    }
    X(ctx)->pipeline = on;
    return SVX_OK;
}

// mode 2: fold the unread event pairs of the calls so far into the running totals (one stream synchronisation)
static void fold_pending(svx_ctx_ext* c) {
    if (c->pending.empty()) return;
    (void)svx_flush(c);
    (void)hipStreamSynchronize(c->stream);
    if (c->side_ready) { (void)hipStreamSynchronize(c->side); (void)hipStreamSynchronize(c->helper); }
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
    const int total = cx->half[0].n_pairs + (cx->last_split > 0 ? cx->half[1].n_pairs : 0);
    NEED(ctx, pair >= 0 && pair < total, "svx_debug_level: pair %d of the last batch (%d pairs)", pair, total);
    const bool second = cx->last_split > 0 && pair >= cx->last_split;
    const SvxPairDev& P = second ? cx->half[1].host[pair - cx->last_split] : cx->half[0].host[pair];
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
    int rc = svx_synchronize(ctx);  // (also launches what the pipeline still holds back)
    if (rc) return rc;
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

// ------------------------------------------------------------------------------ plan of one half
// Level sizes, scratch layout (inside the half's own arena) and launch extents of pairs[0 .. n).
static int plan_half(svx_ctx* ctx, HalfState& H, const BatchParams& bp, const svx_align_params* prm, const svx_pair* pairs, int n_pairs) {
    const int d = bp.d;
    const size_t esz = bp.dtype == SVX_F32 ? 4 : 2;
    const SvxTypes& tfinal = bp.tfinal;
    const int B = bp.B;
    const bool straight = bp.straight, tiles = bp.tiles, packable = bp.packable;
    int mx = 0, my = 0;
    for (int t = 0; t < tfinal.n; t++) { if (tfinal.x[t] > mx) mx = tfinal.x[t]; if (tfinal.y[t] > my) my = tfinal.y[t]; }
    std::vector<SvxPairDev>& host = H.host;
    host.assign(n_pairs, SvxPairDev());
    H.n_pairs = n_pairs;
    H.bp = bp;
    H.maxL = H.max_ksum = H.max_kn = H.max_ds0 = H.max_ds1 = H.max_n0 = H.max_tnd = 0;
    H.any_L0 = false;
    H.o_tickets = 0;
    H.b_knobN = H.b_knob0 = H.dense_cells = 0;
    for (int l = 0; l < SVX_MAX_LEVELS; l++) { H.max_nblk[l] = H.max_A[l] = 0; H.b_pyr[l] = H.b_bc[l] = 0; }
    Bump bump;
    const size_t o_desc = bump.take((size_t)n_pairs * sizeof(SvxPairDev));
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
        if (L > H.maxL) H.maxL = L;
        if (in.n > H.max_n0) H.max_n0 = in.n;
        if (L == 0) H.any_L0 = true;
        if (in.k0 + in.k1 > H.max_ksum) H.max_ksum = in.k0 + in.k1;
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
                if (Lv.nblk[s] > H.max_nblk[l]) H.max_nblk[l] = Lv.nblk[s];
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
            if (Lv.kn > H.max_kn) H.max_kn = Lv.kn;
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
                if (cap > H.max_A[l]) H.max_A[l] = cap;
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
            // launch-order cost model: bytes the streaming passes of this level move
            const double krows = (double)K[0] * Lv.n[0] + (double)K[1] * Lv.n[1], rows = (double)Lv.n[0] + Lv.n[1];
            if (l == 0) {
                H.b_pyr[0] += krows * d * esz;
                H.b_bc[0] += krows * d * esz;
                H.b_knob0 += (double)kn * d * esz;
            } else {
                if (l == 1) H.b_pyr[1] += ((double)K[0] * in.n + (double)K[1] * in.m) * d * esz + rows * d * 4.0;
                else H.b_pyr[l] += (krows + rows) * d * 4.0;
                H.b_pyr[l - 1] += l >= 2 ? krows * d * 4.0 : 0.0;  // (the pair sums the pass below writes)
                if (l < L) { H.b_bc[l] += rows * d * 4.0; H.b_knobN += (double)kn * d * 4.0; }
            }
        }
        const SvxLevel& top = P.lev[L];
        if (!straight) {
            if (top.n[0] > H.max_ds0) H.max_ds0 = top.n[0];
            if (top.n[1] > H.max_ds1) H.max_ds1 = top.n[1];
            H.dense_cells += (double)top.n[0] * top.n[1];
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
    for (int p = 0; p < n_pairs; p++)
        if (host[p].t_nd > H.max_tnd) H.max_tnd = host[p].t_nd;
    if (tiles) H.o_tickets = bump.take(((size_t)(H.max_tnd + 2) * n_pairs + 2) * sizeof(int));
    // ---- the half's arena (grow-only).  Growing it frees memory that launched work may still use: wait for the device
    // first (rare: arenas settle after the first calls)
    if (bump.off > H.arena_bytes) {
        SVX_HIP(ctx, hipDeviceSynchronize());
        if (H.arena) SVX_HIP(ctx, hipFree(H.arena));
        H.arena = nullptr;
        H.arena_bytes = 0;
        const size_t want = bump.off + bump.off / 16 + (1 << 20);
        hipError_t e = hipMalloc(&H.arena, want);
        if (e != hipSuccess) return svx_fail(ctx, SVX_ERR_NOMEM, "scratch arena: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        H.arena_bytes = want;
    }
    const size_t desc_bytes = (size_t)n_pairs * sizeof(SvxPairDev);
    if (desc_bytes > H.pinned_cap) {
        SVX_HIP(ctx, hipDeviceSynchronize());  // (an upload out of the old buffer may still be queued)
        if (H.pinned) SVX_HIP(ctx, hipHostFree(H.pinned));
        H.pinned = nullptr;
        H.pinned_cap = 0;
        SVX_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&H.pinned), desc_bytes + desc_bytes / 8, hipHostMallocDefault));
        H.pinned_cap = desc_bytes + desc_bytes / 8;
    }
    // patch offsets (+1) into device pointers
    char* base = H.arena;
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
    H.dpairs = reinterpret_cast<SvxPairDev*>(base + o_desc);
    return SVX_OK;
}

// ------------------------------------------------------------------------------ launch order
// The pipeline of one half is a streaming FRONT (pyramid, sampled scores) followed by a refinement CHAIN in which
// short latency-bound groups LB (deletion penalty, coarse DP; per level: band DP, traceback, next search path) alternate
// with one streaming kernel S (the level's band costs).  run_interleaved() launches the front of one half and the
// chain of another in ONE order on two streams: every streaming kernel on the context's stream, where it has the
// chip's bandwidth to itself, every LB group on the chain stream beside them; an S kernel is placed behind enough
// front kernels to cover (by the cost model's estimate) the LB group it waits for.
struct Op {
    double us;                 // estimated duration
    int stage;
    std::function<int()> fn;   // launches on ctx->stream
};

static int run_op(svx_ctx* ctx, const Op& op, hipStream_t st) {
    hipStream_t keep = ctx->stream;
    ctx->stream = st;
    int rc;
    if (op.stage < 0) {
        rc = op.fn();  // (the op times itself)
    } else {
        StageScope sc(ctx, op.stage);
        rc = op.fn();
    }
    ctx->stream = keep;
    return rc;
}

static int next_event(svx_ctx* ctx, hipEvent_t* out) {
    svx_ctx_ext* cx = X(ctx);
    if (cx->ev_pool.size() < 256) {
        hipEvent_t e;
        SVX_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        cx->ev_pool.push_back(e);
        *out = e;
        return SVX_OK;
    }
    *out = cx->ev_pool[cx->ev_next++ % cx->ev_pool.size()];  // (a wait refers to the record it was queued behind: re-recording an event later does not disturb it)
    return SVX_OK;
}

// split: every pyramid pass is launched in `split` slices of the half's pairs.  The chain running beside this front
// then gets its streaming kernels placed at a finer grain, and the small latency-bound helpers of a level (column means
// from the block partials, mean of the sampled rows) run on the side stream for slice q while slice q+1 is still in the
// pass of the level above -- off the critical path of the context's stream.
static void front_ops(svx_ctx* ctx, HalfState& H, std::vector<Op>& ops, int split) {
    svx_ctx_ext* cx = X(ctx);
    HalfState* h = &H;
    const BatchParams& bp = H.bp;
    const int np = H.n_pairs, d = bp.d, dtype = bp.dtype;
    const double waves = (double)((np + 1023) / 1024);
    const int parts = (split > 1 && np >= 16 * split) ? split : 1;
    struct Evs { hipEvent_t pass[SVX_MAX_LEVELS][8]; };
    auto evs = std::make_shared<Evs>();
    // marks the start of this front on the context's stream: the side stream's work must not run ahead of it
    ops.push_back({0.0, -1, [=]() -> int {
        SVX_HIP(ctx, hipEventRecord(cx->side_fork, ctx->stream));
        SVX_HIP(ctx, hipStreamWaitEvent(cx->side, cx->side_fork, 0));
        SVX_HIP(ctx, hipStreamWaitEvent(cx->helper, cx->side_fork, 0));
        return SVX_OK;
    }});
    auto sort_op = [=]() -> int {
        // the counting sort of the sampled (x, y) pairs only needs the descriptors: side stream, beside the pyramid
        hipStream_t main_stream = ctx->stream;
        static const bool inline_sort = getenv("SVX_SORT_INLINE") != nullptr;   // measurements: the sort in front of the pyramid instead of beside it
        // sliced front: the sort starts once the level-0 pass (which its LDS atomics slow most) has left the chip
        if (parts > 1 && !inline_sort) SVX_HIP(ctx, hipStreamWaitEvent(cx->side, evs->pass[0][parts - 1], 0));
        ctx->stream = inline_sort ? main_stream : cx->side;
        int rc;
        {
            StageScope sc(ctx, S_KNOB_SORT);  // (timed on the stream it runs on)
            rc = svxl_knob_scores(ctx, h->dpairs, np, h->maxL, h->max_kn, h->max_n0, dtype, d, 0);
        }
        ctx->stream = main_stream;
        if (rc) return rc;
        SVX_HIP(ctx, hipEventRecord(cx->side_join, inline_sort ? main_stream : cx->side));
        return SVX_OK;
    };
    if (parts == 1) ops.push_back({0.0, -1, sort_op});
    for (int l = 0; l <= H.maxL; l++) {
        for (int q = 0; q < parts; q++) {
            const int lo = (int)((long long)np * q / parts), hi = (int)((long long)np * (q + 1) / parts);
            const double aux_us = (l == 0 ? 700.0 * waves : 250.0) / parts;
            if (parts == 1) {
                ops.push_back({aux_us, S_PYR_AUX, [=]() { return svxl_pyramid_level(ctx, h->dpairs, np, l, dtype, d, h->max_nblk[l], h->max_ksum, 0); }});
                ops.push_back({H.b_pyr[l] / (l == 0 ? 4.9e6 : 5.3e6), l == 0 ? S_PYR0 : (l == 1 ? S_PYR1 : S_PYRN),
                               [=]() { return svxl_pyramid_level(ctx, h->dpairs, np, l, dtype, d, h->max_nblk[l], h->max_ksum, 1); }});
                continue;
            }
            ops.push_back({H.b_pyr[l] / (l == 0 ? 4.9e6 : 5.3e6) / parts, l == 0 ? S_PYR0 : (l == 1 ? S_PYR1 : S_PYRN), [=]() -> int {
                hipStream_t main_stream = ctx->stream;
                int rc;
                if (l == 0 && q == 0) {
                    // the front's very first helper has nothing to hide behind: on the context's stream itself (two
                    // cross-stream hand-offs, ~0.2 ms, off the start of every half-batch)
                    StageScope sc(ctx, S_PYR_AUX);
                    if ((rc = svxl_pyramid_level(ctx, h->dpairs + lo, hi - lo, l, dtype, d, h->max_nblk[l], h->max_ksum, 0))) return rc;
                } else {
                    // the slice's helpers: helper stream, behind the slice's pass of the level above
                    if (l > 0) SVX_HIP(ctx, hipStreamWaitEvent(cx->helper, evs->pass[l - 1][q], 0));
                    ctx->stream = cx->helper;
                    {
                        StageScope sc(ctx, S_PYR_AUX);
                        rc = svxl_pyramid_level(ctx, h->dpairs + lo, hi - lo, l, dtype, d, h->max_nblk[l], h->max_ksum, 0);
                    }
                    ctx->stream = main_stream;
                    if (rc) return rc;
                    hipEvent_t aux_done;
                    if ((rc = next_event(ctx, &aux_done))) return rc;
                    SVX_HIP(ctx, hipEventRecord(aux_done, cx->helper));
                    SVX_HIP(ctx, hipStreamWaitEvent(main_stream, aux_done, 0));
                }
                if ((rc = svxl_pyramid_level(ctx, h->dpairs + lo, hi - lo, l, dtype, d, h->max_nblk[l], h->max_ksum, 1))) return rc;
                if ((rc = next_event(ctx, &evs->pass[l][q]))) return rc;
                SVX_HIP(ctx, hipEventRecord(evs->pass[l][q], main_stream));
                return SVX_OK;
            }});
        }
        if (parts > 1 && l == 0) ops.push_back({0.0, -1, sort_op});   // (beside the level-1 pass, which has vector-issue slots to spare)
    }
    if (!bp.straight) {
        // coarsest level: the dense 1-1 cost matrix, and with it the dot products its sampled scores need
        ops.push_back({H.dense_cells * 2.7e-5, S_DENSE_COSTS, [=]() -> int {
            int rc = svxl_dense_costs_batch(ctx, h->dpairs, np, h->max_ds0, h->max_ds1, dtype, d);
            if (rc) return rc;
            return svxl_knob_from_dots(ctx, h->dpairs, np, h->max_kn);
        }});
    }
    if (H.maxL >= 1) {
        ops.push_back({H.b_knobN / 16.5e6, S_KNOB_SCORES, [=]() -> int {
            SVX_HIP(ctx, hipStreamWaitEvent(ctx->stream, cx->side_join, 0));
            return svxl_knob_scores(ctx, h->dpairs, np, h->maxL, h->max_kn, h->max_n0, dtype, d, 1);
        }});
    }
    ops.push_back({H.b_knob0 / 8.5e6, S_KNOB_SCORES0, [=]() -> int {
        SVX_HIP(ctx, hipStreamWaitEvent(ctx->stream, cx->side_join, 0));
        return svxl_knob_scores(ctx, h->dpairs, np, h->maxL, h->max_kn, h->max_n0, dtype, d, 2);
    }});
}

struct Seg {
    std::vector<Op> lb;   // latency-bound group (chain stream)
    bool has_s = false;
    Op s;                 // the streaming kernel that follows it (context's stream)
};

static void chain_segs(svx_ctx* ctx, HalfState& H, std::vector<Seg>& segs) {
    HalfState* h = &H;
    const BatchParams bp = H.bp;
    const int np = H.n_pairs, d = bp.d, dtype = bp.dtype, W = bp.W, B = bp.B;
    const double waves = (double)((np + 1023) / 1024);
    Seg cur;
    cur.lb.push_back({(np * (H.maxL + 1) * 0.21 < 300.0 ? 300.0 : np * (H.maxL + 1) * 0.21), S_KNOB,
                      [=]() { return svxl_del_penalty_batch(ctx, h->dpairs, np, h->maxL + 1, bp.frac); }});
    if (!bp.straight)
        cur.lb.push_back({(H.max_ds0 + H.max_ds1) * 1.7 * waves, S_DENSE_DP, [=]() { return svxl_dense_stage_batch(ctx, h->dpairs, np, h->max_ds0); }});
    const int first_depth = H.maxL > 0 ? H.maxL - 1 : 0;
    for (int depth = first_depth; depth >= 0; depth--) {
        const SvxTypes ty = depth == 0 ? bp.tfinal : bp.t11;
        int lim2 = 0, tamax2 = 0;
        const bool v2 = !bp.tiles && svxl_band2_limits(ty, W, depth, dtype, d, &lim2, &tamax2);  // which band-cost kernel takes this level
        const int maxA = H.max_A[depth];
        // (pairs whose source level has more alignment rows than the LDS holds take the kernel's serial path)
        const int src_rows = (H.maxL > 0 && !H.any_L0) ? maxA / 2 + 8 : maxA;
        cur.lb.push_back({maxA * 0.08 * waves + 60.0, depth == 0 ? S_PATH0 : S_PATH, [=]() {
            return svxl_search_path_batch(ctx, h->dpairs, np, depth, maxA, src_rows, v2 ? lim2 : SVX_BC_ROWS - SVX_BC_TB,
                                          bp.tiles ? 0 : (v2 ? tamax2 : SVX_BC_TAMAX));
        }});
        cur.has_s = true;
        if (bp.tiles) {
            cur.s = {1.0e4, S_TILES, [=]() {
                int* tk = reinterpret_cast<int*>(h->arena + h->o_tickets);
                return svxl_band_tiles_batch(ctx, h->dpairs, np, ty, W, dtype, h->max_tnd, tk, tk + (size_t)h->max_tnd * np + 1);
            }};
        } else {
            cur.s = {H.b_bc[depth] / 4.0e6, depth == 0 ? S_BAND_COSTS0 : S_BAND_COSTSN, [=]() {
                return v2 ? svxl_band_costs2_batch(ctx, h->dpairs, np, depth, maxA, ty, W, dtype, d)
                          : svxl_band_costs_batch(ctx, h->dpairs, np, depth, maxA, ty, W, dtype, d);
            }};
        }
        segs.push_back(cur);
        cur = Seg();
        if (!bp.tiles)
            cur.lb.push_back({maxA * (0.21 + 0.054 * (ty.n > 1 ? ty.n : 0)) * waves, depth == 0 ? S_BAND_DP0 : S_BAND_DPN,
                              [=]() { return svxl_sparse_dp_batch(ctx, h->dpairs, np, depth, ty, B); }});
        cur.lb.push_back({maxA * 0.27 * waves, depth == 0 ? S_TRACEBACK0 : S_TRACEBACK, [=]() { return svxl_sparse_traceback_batch(ctx, h->dpairs, np, depth, B, maxA, bp.packable ? 1 : 0); }});
    }
    segs.push_back(cur);
}

static int ensure_streams(svx_ctx* ctx) {
    svx_ctx_ext* cx = X(ctx);
    if (!cx->side_ready) {
        int prio_lo = 0, prio_hi = 0;
        SVX_HIP(ctx, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));  // (lowest, highest): the side work only fills gaps
        SVX_HIP(ctx, hipStreamCreateWithPriority(&cx->side, hipStreamNonBlocking, prio_lo));
        SVX_HIP(ctx, hipStreamCreateWithPriority(&cx->helper, hipStreamNonBlocking, prio_hi));
        SVX_HIP(ctx, hipEventCreateWithFlags(&cx->side_fork, hipEventDisableTiming));
        SVX_HIP(ctx, hipEventCreateWithFlags(&cx->side_join, hipEventDisableTiming));
        cx->side_ready = true;
    }
    if (!cx->chain_ready) {
        int prio_lo = 0, prio_hi = 0;
        SVX_HIP(ctx, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        // the chain's few small workgroups should be placed as soon as a streaming workgroup retires
        SVX_HIP(ctx, hipStreamCreateWithPriority(&cx->chain, hipStreamNonBlocking, prio_hi));
        cx->chain_ready = true;
    }
    for (int h = 0; h < 2; h++) {
        if (!cx->half[h].front_done) SVX_HIP(ctx, hipEventCreateWithFlags(&cx->half[h].front_done, hipEventDisableTiming));
        if (!cx->half[h].chain_done) SVX_HIP(ctx, hipEventCreateWithFlags(&cx->half[h].chain_done, hipEventDisableTiming));
        if (!cx->half[h].upload_done) SVX_HIP(ctx, hipEventCreateWithFlags(&cx->half[h].upload_done, hipEventDisableTiming));
    }
    return SVX_OK;
}

// front: half whose descriptors are uploaded and whose streaming front is launched now (or null);
// chain: half whose front is on the context's stream already and whose refinement chain is launched now (or null).
// beside: run the chain's LB groups on the chain stream (else everything goes to the context's stream, in order).
static int run_interleaved(svx_ctx* ctx, HalfState* front, HalfState* chain, bool beside) {
    svx_ctx_ext* cx = X(ctx);
    hipStream_t A = ctx->stream, Bst = beside ? cx->chain : ctx->stream;
    std::vector<Op> fops;
    std::vector<Seg> segs;
    if (front) {
        // the half's arena and outputs are free once ITS previous chain has finished
        if (front->chain_done_valid) SVX_HIP(ctx, hipStreamWaitEvent(A, front->chain_done, 0));
        {
            StageScope sc(ctx, S_SETUP);
            // (the host may be several calls ahead of the device: the previous upload out of this buffer must have left it)
            if (front->upload_valid) SVX_HIP(ctx, hipEventSynchronize(front->upload_done));
            memcpy(front->pinned, front->host.data(), (size_t)front->n_pairs * sizeof(SvxPairDev));
            SVX_HIP(ctx, hipMemcpyAsync(front->dpairs, front->pinned, (size_t)front->n_pairs * sizeof(SvxPairDev), hipMemcpyHostToDevice, A));
            SVX_HIP(ctx, hipEventRecord(front->upload_done, A));
            front->upload_valid = true;
            hipLaunchKernelGGL(k_init_batch, dim3(front->n_pairs), dim3(64), 0, A, front->dpairs);
            SVX_LAUNCH_CHECK(ctx, "k_init_batch");
            if (front->bp.nsamp == 0) {
                // compute_norms returns ones without samples (dp_utils.py:356-357): rbar = 0 -> 1 - 0
                for (int p = 0; p < front->n_pairs; p++)
                    for (int l = 0; l <= front->host[p].L; l++)
                        for (int s = 0; s < 2; s++)
                            SVX_HIP(ctx, hipMemsetAsync(front->host[p].lev[l].rbar[s], 0, (size_t)front->bp.d * sizeof(float), A));
            }
        }
        static const int split_env = [] { const char* e = getenv("SVX_PIPE_SPLIT"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 8 ? v : 2; }();
        front_ops(ctx, *front, fops, beside ? split_env : 1);
    }
    if (chain) chain_segs(ctx, *chain, segs);
    int rc;
    size_t fi = 0;
    if (chain && beside) SVX_HIP(ctx, hipStreamWaitEvent(Bst, chain->front_done, 0));
    for (size_t g = 0; g < segs.size(); g++) {
        double need = 0.0;
        for (const Op& op : segs[g].lb) {
            if ((rc = run_op(ctx, op, Bst))) return rc;
            need += op.us;
        }
        hipEvent_t lb_done = nullptr;
        if (beside) {
            if ((rc = next_event(ctx, &lb_done))) return rc;
            SVX_HIP(ctx, hipEventRecord(lb_done, Bst));
        }
        if (!segs[g].has_s) break;
        // cover the group with front kernels before the streaming kernel that waits for it
        double acc = 0.0;
        need = need * 1.25 + 50.0;   // (1.6, 2.0, 2.6: the same step time)
        while (beside && fi < fops.size() && acc < need) {
            if ((rc = run_op(ctx, fops[fi], A))) return rc;
            acc += fops[fi].us;
            fi++;
        }
        if (beside) SVX_HIP(ctx, hipStreamWaitEvent(A, lb_done, 0));
        if ((rc = run_op(ctx, segs[g].s, A))) return rc;
        if (beside) {
            hipEvent_t s_done;
            if ((rc = next_event(ctx, &s_done))) return rc;
            SVX_HIP(ctx, hipEventRecord(s_done, A));
            SVX_HIP(ctx, hipStreamWaitEvent(Bst, s_done, 0));
        }
    }
    for (; fi < fops.size(); fi++)
        if ((rc = run_op(ctx, fops[fi], A))) return rc;
    if (front) {
        SVX_HIP(ctx, hipEventRecord(front->front_done, A));
        front->chain_pending = true;
    }
    if (chain) {
        SVX_HIP(ctx, hipEventRecord(chain->chain_done, Bst));
        chain->chain_done_valid = true;
        chain->chain_pending = false;
    }
    return SVX_OK;
}

extern "C" int svx_flush(svx_ctx* ctx) {
    if (!ctx) return SVX_ERR_ARG;
    svx_ctx_ext* cx = X(ctx);
    bool any = false;
    for (int h = 0; h < 2; h++) any = any || cx->half[h].chain_pending || cx->half[h].chain_done_valid;
    if (!any) return SVX_OK;
    SVX_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    for (int h = 0; h < 2; h++)
        if (cx->half[h].chain_pending && (rc = run_interleaved(ctx, nullptr, &cx->half[h], true))) return rc;
    // results are complete, in the order of the context's stream, once it has waited for both chains
    for (int h = 0; h < 2; h++)
        if (cx->half[h].chain_done_valid) SVX_HIP(ctx, hipStreamWaitEvent(ctx->stream, cx->half[h].chain_done, 0));
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
    BatchParams bp;
    bp.d = prm->d;
    bp.dtype = prm->dtype;
    NEED(ctx, bp.dtype == SVX_F32 || bp.dtype == SVX_F16 || bp.dtype == SVX_BF16, "svx_align_batch: unknown dtype %d", bp.dtype);
    NEED(ctx, prm->max_size_full_dp >= 1, "svx_align_batch: max_size_full_dp must be >= 1");
    NEED(ctx, prm->costs_sample_size >= 1, "svx_align_batch: costs_sample_size must be >= 1");
    NEED(ctx, prm->num_samps_for_norm >= 0, "svx_align_batch: num_samps_for_norm must be >= 0");
    if ((rc = make_types(ctx, prm->types, prm->n_types, &bp.tfinal))) return rc;
    const int32_t one_one[2] = {1, 1};
    make_types(ctx, one_one, 1, &bp.t11);
    bp.W = prm->width_over2 < 3 ? 3 : prm->width_over2;  // dp_utils.py:391-393
    bp.B = 2 * bp.W;
    bp.nsamp = prm->num_samps_for_norm;
    bp.frac = prm->del_percentile_frac;
    bp.n_types = bp.tfinal.n;
    int mx = 0, my = 0;
    for (int t = 0; t < bp.tfinal.n; t++) { if (bp.tfinal.x[t] > mx) mx = bp.tfinal.x[t]; if (bp.tfinal.y[t] > my) my = bp.tfinal.y[t]; }
    bp.packable = mx <= 15 && my <= 15;  // back-pointers fit 4 bits each
    NEED(ctx, prm->search_mode == SVX_SEARCH_COARSE_TO_FINE || prm->search_mode == SVX_SEARCH_STRAIGHT, "svx_align_batch: unknown search_mode %d", prm->search_mode);
    bp.straight = prm->search_mode == SVX_SEARCH_STRAIGHT;
    // straight search with a band wider than the one-workgroup DP kernel takes: wavefront of tiles (svx_tiles.hip)
    bp.tiles = bp.straight && bp.B > 64;
    if (bp.tiles) {
        NEED(ctx, bp.packable && bp.tfinal.n >= 1 && svxl_band_tiles_ok(bp.tfinal),
             "wide straight band: the tile kernel takes 1..16 alignment types on <= 12 overlap layers with steps <= 8");
    }
    if ((rc = ensure_streams(ctx))) return rc;
    for (int i = 0; i < S_COUNT; i++) { cx->ms[i] = 0.0; cx->launches[i] = 0; }
    for (auto& r : cx->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    cx->recs.clear();
    // the tile sweep is a persistent kernel over all CUs: nothing runs beside it
    const bool piped = cx->pipeline != 0 && !bp.tiles && n_pairs >= 2;
    double plan_ms = 0.0;
    {
        StageScope total(ctx, S_TOTAL);
        if (!piped) {
            if ((rc = svx_flush(ctx))) return rc;   // (what an earlier pipelined call left behind)
            HalfState& H = cx->half[0];
            if ((rc = plan_half(ctx, H, bp, prm, pairs, n_pairs))) return rc;
            cx->last_split = 0;
            plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
            if ((rc = run_interleaved(ctx, &H, nullptr, false))) return rc;
            if ((rc = run_interleaved(ctx, nullptr, &H, false))) return rc;
            H.chain_done_valid = false;  // (everything ran on the context's stream: nothing to wait for)
        } else {
            const int n0 = (n_pairs + 1) / 2;
            HalfState &H0 = cx->half[0], &H1 = cx->half[1];
            if (H0.chain_pending && (rc = run_interleaved(ctx, nullptr, &H0, true))) return rc;  // (cannot happen; kept for safety)
            auto t0 = std::chrono::steady_clock::now();
            if ((rc = plan_half(ctx, H0, bp, prm, pairs, n0))) return rc;
            plan_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            // first half's front beside the chain the previous call left behind
            if ((rc = run_interleaved(ctx, &H0, H1.chain_pending ? &H1 : nullptr, true))) return rc;
            t0 = std::chrono::steady_clock::now();
            if ((rc = plan_half(ctx, H1, bp, prm, pairs + n0, n_pairs - n0))) return rc;
            plan_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            cx->last_split = n0;
            // second half's front beside the first half's chain; its own chain waits for the next call or svx_flush
            if ((rc = run_interleaved(ctx, &H1, &H0, true))) return rc;
        }
    }
    cx->last_ntypes = bp.tfinal.n;
    cx->last_band = bp.B;
    const auto t_done = std::chrono::steady_clock::now();
    cx->ms[S_HOST_PLAN] = plan_ms;
    cx->ms[S_HOST_LAUNCH] = std::chrono::duration<double, std::milli>(t_done - t_enter).count() - plan_ms;
    if (ctx->profiling == 2) {
        // accumulate: nothing is read and nothing waits here, the next call can be launched behind this one
        for (auto& r : cx->recs) cx->pending.push_back(r);
        cx->recs.clear();
        for (int i = 0; i < S_COUNT; i++) cx->acc_launches[i] += cx->launches[i];
        cx->acc_ms[S_HOST_PLAN] += cx->ms[S_HOST_PLAN];
        cx->acc_ms[S_HOST_LAUNCH] += cx->ms[S_HOST_LAUNCH];
    } else if (ctx->profiling) {
        if ((rc = svx_synchronize(ctx))) return rc;
        if (cx->side_ready) { SVX_HIP(ctx, hipStreamSynchronize(cx->side)); SVX_HIP(ctx, hipStreamSynchronize(cx->helper)); }
        for (auto& r : cx->recs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) cx->ms[r.stage] += ms;
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
        cx->recs.clear();
    }
    return SVX_OK;
}

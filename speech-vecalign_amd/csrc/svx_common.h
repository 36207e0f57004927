// svx_common.h -- internal declarations shared by the libsvx translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/svx.h"

#define SVX_WAVE 64

// Row-pair slots handled by one k_pyramid workgroup (4 waves x 8 slots).
#ifndef SVX_PYR_SLOTS
#define SVX_PYR_SLOTS 32
#endif
// Band-cost tiling: path points per chunk, band cells per chunk, rows staged per side.
#define SVX_BC_TA 32
// The fused pipeline cuts the path into chunks of up to SVX_BC_TAMAX points whose extent on either side still
// fits the SVX_BC_ROWS staged rows (k_chunk_path); the per-op entry point keeps fixed chunks of SVX_BC_TA.
#define SVX_BC_TAMAX 64
#define SVX_BC_TB 16
#define SVX_BC_ROWS 48

// ---- element types ---------------------------------------------------------------------
struct ElemF32 {
    using storage = float;
    static constexpr int VEC = 4;  // elements per 16-byte piece
    static constexpr int DT = SVX_F32;
};
struct ElemF16 {
    using storage = uint16_t;
    static constexpr int VEC = 8;
    static constexpr int DT = SVX_F16;
};
struct ElemBF16 {
    using storage = uint16_t;
    static constexpr int VEC = 8;
    static constexpr int DT = SVX_BF16;
};

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

// ---- accesses through global-address-space pointers.  The batch kernels take their pointers out of SvxPairDev
// records in memory, which the compiler can only type as generic: a dereference becomes flat_load / flat_store,
// which count against vmcnt AND lgkmcnt, so every wait for an LDS read also drains the row loads in flight (the
// ISA showed nothing but `s_waitcnt vmcnt(0) lgkmcnt(0)`).  Device buffers are global memory: say so at the access.
#define SVX_GLOBAL(T) __attribute__((address_space(1))) T
typedef uint32_t svx_u32x4 __attribute__((ext_vector_type(4)));
typedef float svx_f32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ T gld(const T* p) { return *(const SVX_GLOBAL(T)*)p; }
template <typename T>
__device__ __forceinline__ void gst(T* p, T v) { *(SVX_GLOBAL(T)*)p = v; }
__device__ __forceinline__ uint4 gld16(const void* p) {
    const svx_u32x4 v = *(const SVX_GLOBAL(svx_u32x4)*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 gld16_nt(const void* p) {  // streaming read: the line is not kept for a later pass
    const svx_u32x4 v = __builtin_nontemporal_load((const SVX_GLOBAL(svx_u32x4)*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gst16(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const svx_u32x4 v = {a, b, c, d};
    *(SVX_GLOBAL(svx_u32x4)*)p = v;
}
__device__ __forceinline__ float4 gldf4(const float* p) {
    const svx_f32x4 v = *(const SVX_GLOBAL(svx_f32x4)*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gstf4(float* p, float a, float b, float c, float d) {
    const svx_f32x4 v = {a, b, c, d};
    *(SVX_GLOBAL(svx_f32x4)*)p = v;
}
__device__ __forceinline__ void gstf4_nt(float* p, float a, float b, float c, float d) {  // streaming store
    const svx_f32x4 v = {a, b, c, d};
    __builtin_nontemporal_store(v, (SVX_GLOBAL(svx_f32x4)*)p);
}

// Widen one 16-byte piece held in registers to VEC floats.
template <typename E>
__device__ __forceinline__ void decode_piece(const uint4& v, float* out);
template <>
__device__ __forceinline__ void decode_piece<ElemF32>(const uint4& v, float* out) {
    out[0] = __uint_as_float(v.x); out[1] = __uint_as_float(v.y); out[2] = __uint_as_float(v.z); out[3] = __uint_as_float(v.w);
}
template <>
__device__ __forceinline__ void decode_piece<ElemF16>(const uint4& v, float* out) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        out[2 * i] = f16_to_f32((uint16_t)(w[i] & 0xffffu));
        out[2 * i + 1] = f16_to_f32((uint16_t)(w[i] >> 16));
    }
}
template <>
__device__ __forceinline__ void decode_piece<ElemBF16>(const uint4& v, float* out) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        out[2 * i] = __uint_as_float(w[i] << 16);
        out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
// One 16-byte piece out of global memory, widened.
template <typename E>
__device__ __forceinline__ void gload_piece(const typename E::storage* p, float* out) {
    decode_piece<E>(gld16(p), out);
}

// Load one 16-byte piece and widen to VEC floats.
template <typename E>
__device__ __forceinline__ void load_piece(const typename E::storage* p, float* out);
template <>
__device__ __forceinline__ void load_piece<ElemF32>(const float* p, float* out) {
    float4 v = *reinterpret_cast<const float4*>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <>
__device__ __forceinline__ void load_piece<ElemF16>(const uint16_t* p, float* out) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        out[2 * i] = f16_to_f32((uint16_t)(w[i] & 0xffffu));
        out[2 * i + 1] = f16_to_f32((uint16_t)(w[i] >> 16));
    }
}
template <>
__device__ __forceinline__ void load_piece<ElemBF16>(const uint16_t* p, float* out) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        out[2 * i] = __uint_as_float(w[i] << 16);
        out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}

// Exchange with lane ^ 16 / lane ^ 32 through the gfx950 permlane swaps (VALU, no LDS round trip).
__device__ __forceinline__ unsigned xchg16_u32(unsigned v, int lane) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return ((lane >> 4) & 1) ? r[0] : r[1];
}
__device__ __forceinline__ unsigned xchg32_u32(unsigned v, int lane) {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane & 32) ? r[0] : r[1];
}
#define SVX_DPP_F32(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false))

// Wave-wide reductions on the VALU only: two quad permutes, the half-row and row mirrors (after which all 16
// lanes of a row agree), then the two permlane swaps across rows.  Every lane ends with the result.
__device__ __forceinline__ float wave_sum(float v) {
    const int lane = threadIdx.x & 63;
    v += SVX_DPP_F32(v, 0xB1);   // quad_perm [1,0,3,2]
    v += SVX_DPP_F32(v, 0x4E);   // quad_perm [2,3,0,1]
    v += SVX_DPP_F32(v, 0x141);  // row_half_mirror
    v += SVX_DPP_F32(v, 0x140);  // row_mirror
    v += __uint_as_float(xchg16_u32(__float_as_uint(v), lane));
    v += __uint_as_float(xchg32_u32(__float_as_uint(v), lane));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    const int lane = threadIdx.x & 63;
    v = fmaxf(v, SVX_DPP_F32(v, 0xB1));
    v = fmaxf(v, SVX_DPP_F32(v, 0x4E));
    v = fmaxf(v, SVX_DPP_F32(v, 0x141));
    v = fmaxf(v, SVX_DPP_F32(v, 0x140));
    v = fmaxf(v, __uint_as_float(xchg16_u32(__float_as_uint(v), lane)));
    v = fmaxf(v, __uint_as_float(xchg32_u32(__float_as_uint(v), lane)));
    return v;
}

// ---- per-pair device descriptors of the fused pipeline (svx_align_batch) ---------------
struct SvxLevel {
    int n[2];          // rows per side at this level
    int nblk[2];       // k_pyramid workgroups along the rows of this level
    int npart[2];      // column-sum partials feeding THIS level's mean (= nblk of level-1)
    int S[2];          // sampled rows per layer taken FROM side s
    float* P[2];       // level >= 1: [K][n][d] pair sums of level-1; layer 0 becomes the normalised rows
    float* part[2];    // level >= 1: [K][npart][d]
    float* mean[2];    // level >= 1: [K][d]
    float* rbar[2];    // [d] mean of the sampled normalised rows of side s
    float* inv[2];     // level 0: [K][n] 1/(||row|| + 1e-5)
    float* nrm[2];     // [K][n] the reference's n0 / n1
    const int* sidx[2];  // [K_s][S_s] sampled row indices into side s
    const int* kx;     // knob sample indices (side 0 / side 1 rows)
    const int* ky;
    int kn;
    int path_cap;
    float* kscore;     // [kn]
    int* korder;       // [kn] sample ids grouped by source row
    int* kys;          // [kn] target row of korder[pos] (clamped), so that the scoring pass needs no dependent lookups
    int* kstart;       // [n0 + 1] first position of each source row in korder
    double* pen;       // deletion penalty of this level
    int* path;         // [path_cap][2]
    int* path_len;
    int* cstart;       // [n_chunks + 1] first path point of every band-cost chunk
    int* nchunks;
    float* costs;      // [A][T][B] (fused pipeline layout: one diagonal's costs are contiguous)
    int* boff;         // [A]
    double* csum;      // [A+2][B]
    int* xp;           // int32 back-pointers (only when the types do not pack into 4 bits)
    int* yp;
    unsigned char* bpk;  // packed back-pointers [A+2][B]: xp << 4 | yp, 0xFF = unreachable
    int* boff_out;     // [A+2]
    int* align;        // [n0+n1+2][4]
    int* n_align;
    double* scores;
};

struct SvxPairDev {
    const void* v[2];
    int K[2];
    int L;             // max_depth
    int d;
    int norm_override[2];
    float* dcost;      // dense stage at level L: [s0][s1]
    float* ddot;       // [s0][s1] raw dot products of the same stage (L >= 1): the level's sampled scores read them
    int* dbp;          // [s0+s1+1][s0+1]: back-pointers of the dense stage by (anti-diagonal, row)
    int* status;       // info[1]
    // straight-band search with the tile sweep (svx_tiles.hip): per tile anti-diagonal s the run of band tiles
    int straight;
    int t_nd, t_cap;   // tile anti-diagonals; capacity of t_flag
    int* t_lo;         // [t_nd] first tile row I of diagonal s
    int* t_cnt;        // [t_nd]
    int* t_pref;       // [t_nd + 1] tiles before diagonal s
    int* t_flag;       // [t_cap] tile finished
    SvxLevel lev[SVX_MAX_LEVELS];
};

struct SvxTypes {
    int n;
    int maxstep;               // max(x+y) over types and the two deletions
    int8_t x[SVX_MAX_TYPES + 2];  // types..., then (0,1), (1,0)
    int8_t y[SVX_MAX_TYPES + 2];
};

// ---- host-side context -----------------------------------------------------------------
struct svx_ctx {
    int device;
    hipStream_t stream;
    char* arena;
    size_t arena_bytes;
    size_t arena_used;
    char err[512];
    int profiling;
};

int svx_fail(svx_ctx* ctx, int code, const char* fmt, ...);
#define SVX_HIP(ctx, call)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return svx_fail(ctx, SVX_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define SVX_LAUNCH_CHECK(ctx, name)                                                            \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess) return svx_fail(ctx, SVX_ERR_HIP, "launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---- kernel launchers (defined in the .hip files) ---------------------------------------
// rows / pyramid (svx_rows.hip)
int svxl_make_norm1(svx_ctx*, float* vecs, int64_t rows, int d);
int svxl_pairsum(svx_ctx*, const float* vecs, int k, int n, int d, float* half, float* part, int nblk);
int svxl_colmean_plain(svx_ctx*, const float* part, int k, int nblk, int d, int count, float* mean);
int svxl_sub_mean(svx_ctx*, float* half, int k, int h, int d, const float* mean);
int svxl_sample_mean_plain(svx_ctx*, const float* vecs, int k, int n, int d, const int* idx, int S, float* rbar);
int svxl_norms_from_rbar(svx_ctx*, const float* vecs, int64_t rows, int d, const float* rbar, float* norms);
int svxl_gather_rows(svx_ctx*, const void* table, long long n_rows, int row_bytes, int dtype, const int* idx, long long n_out, void* out);
int svxl_pyramid_level(svx_ctx*, const SvxPairDev* pairs, int n_pairs, int level, int dtype, int d, int max_nblk,
                       int max_ksum, int part);
// costs (svx_costs.hip)
int svxl_score_path(svx_ctx*, const int* xx, const int* yy, int64_t n, const float* n1, const float* n2,
                    const float* v1, int rows1, const float* v2, int rows2, int d, float* out);
int svxl_knob_scores(svx_ctx*, const SvxPairDev* pairs, int n_pairs, int max_levels, int max_kn, int max_n0, int dtype, int d,
                     int part);
int svxl_dense_costs(svx_ctx*, const float* v0, int s0, const float* v1, int s1, int d, const float* n0,
                     const float* n1, int mul0, int mul1, float* costs);
int svxl_dense_costs_batch(svx_ctx*, const SvxPairDev* pairs, int n_pairs, int max_s0, int max_s1, int dtype, int d);
int svxl_knob_from_dots(svx_ctx*, const SvxPairDev* pairs, int n_pairs, int max_kn);
int svxl_band_costs(svx_ctx*, const void* v0, int k0, int n, const void* v1, int k1, int m, int d, int dtype,
                    const float* inv0, const float* inv1, const float* nrm0, const float* nrm1, const int* path,
                    int A, const SvxTypes& types, int W, float* costs, int* boff, int* status);
int svxl_band_costs_batch(svx_ctx*, const SvxPairDev* pairs, int n_pairs, int depth, int max_A, const SvxTypes& types,
                          int W, int dtype, int d);
// costs, second-generation band kernel of the fused pipeline (svx_band.hip)
bool svxl_band2_limits(const SvxTypes& types, int W, int depth, int dtype, int d, int* lim, int* tamax);
int svxl_band_costs2_batch(svx_ctx*, const SvxPairDev* pairs, int n_pairs, int depth, int max_A, const SvxTypes& types, int W,
                           int dtype, int d);
// wide bands as a wavefront of tiles (svx_tiles.hip)
bool svxl_band_tiles_ok(const SvxTypes& types);
int svxl_band_tiles_batch(svx_ctx*, const SvxPairDev* pairs, int n_pairs, const SvxTypes& types, int W, int dtype, int max_nd,
                          int* gpref, int* ticket);
// dp (svx_dp.hip)
int svxl_dense_dp(svx_ctx*, const float* cost, int s0, int s1, float pen, double* csum, int* bp);
int svxl_dense_stage_batch(svx_ctx*, const SvxPairDev* pairs, int n_pairs, int max_s0);
int svxl_dense_traceback(svx_ctx*, const int* bp, int s0, int s1, int* align, int* count);
int svxl_sparse_dp(svx_ctx*, const float* costs, const int* boff_in, int A, int B, const SvxTypes& types, double pen,
                   int xs, int ys, double* csum, int* xp, int* yp, int* boff_out);
int svxl_sparse_dp_batch(svx_ctx*, const SvxPairDev* pairs, int n_pairs, int depth, const SvxTypes& types, int B);
int svxl_sparse_traceback(svx_ctx*, const double* csum, const int* xp, const int* yp, const int* boff, int a_out,
                          int B, int xs, int ys, int* align, double* scores, int* count);
int svxl_sparse_traceback_batch(svx_ctx*, const SvxPairDev* pairs, int n_pairs, int depth, int B, int max_A, int packed);
int svxl_search_path(svx_ctx*, const int* align, const int* n_align, int upsample, int size0, int size1, int* path,
                     int cap, int* path_len);
int svxl_search_path_batch(svx_ctx*, const SvxPairDev* pairs, int n_pairs, int depth, int max_rows, int max_src_rows, int chunk_lim,
                           int chunk_tamax);  // chunk_tamax <= 0: no band-cost chunks wanted
int svxl_del_penalty(svx_ctx*, const float* scores, int64_t n, double frac, double* out);
int svxl_del_penalty_batch(svx_ctx*, const SvxPairDev* pairs, int n_pairs, int max_levels, double frac);

// svx_margin.hip -- global margin scoring of alignments (gfx950 only).
//
// Restates svecalign/postprocess/score_align.py:118-161 (compute_sim_with_nonflat_idx) for an exact
// ("Flat") database that lives in HBM as unit-norm fp16 / bf16 rows -- the storage the reference gets
// from faiss with gpu_type "fp16-shard" (score_align.py:48-50, prep_index.py:153-185):
//   mean_xy[i] = mean over the k nearest rows of DB_y of <x_i/|x_i|, row>     (cosine = (2 - L2^2) / 2)
//   score[i]   = <x_i, y_i> / ((mean_xy[i] + mean_yx[i]) / 2)                 ("ratio"; "distance" subtracts)
// k_knn_mean is a GEMM fused with a per-row top-k: the 16 x d query block of a wave stays in registers
// as MFMA A-fragments for the whole sweep, database tiles stream through LDS once per workgroup and the
// similarity matrix never exists in memory.
#include <math.h>

#include "svx_common.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

#define KNN_DT 32       // database rows per LDS tile
#define KNN_KSTEPS 32   // 32-element k-steps held in registers (d <= 1024)
#define KNN_KMAX 64
#define KNN_SPAD 33
// LDS row stride of a database tile: always the full 1024 elements + 16 B (conflict-free b128 reads); for
// d < 1024 the columns past d are cleared once and stay zero, so the MFMA loop needs no bounds test.
#define KNN_RS (KNN_KSTEPS * 64 + 16)

__device__ __forceinline__ uint32_t pack_pair(float a, float b, bool bf) {
    if (bf) {
        uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
        ua = (ua + 0x7fffu + ((ua >> 16) & 1u)) >> 16;
        ub = (ub + 0x7fffu + ((ub >> 16) & 1u)) >> 16;
        return ua | (ub << 16);
    }
    const uint16_t ha = __builtin_bit_cast(uint16_t, (_Float16)a), hb = __builtin_bit_cast(uint16_t, (_Float16)b);
    return (uint32_t)ha | ((uint32_t)hb << 16);
}

// 8 consecutive elements of a query row, widened to fp32.
template <typename QE>
__device__ __forceinline__ void load8(const typename QE::storage* p, float* f) {
    if (QE::VEC == 4) {
        load_piece<QE>(p, f);
        load_piece<QE>(p + 4, f + 4);
    } else {
        load_piece<QE>(p, f);
    }
}

template <bool BF>
__device__ __forceinline__ void mma16(f32x4_t& acc, const uint4& a, const uint4& b) {
    if (BF)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    else
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Tile t of the database -> LDS, with no register stop-over (global_load_lds_dwordx4: every lane names its
// own source address, the wave's 64 x 16 B land back to back at a wave-uniform LDS address = half a row).
// Wave w brings rows 8w .. 8w+7.  Rows past the end of the database are read from its last row and masked
// out at the top-k update; lanes past the row's d elements stay idle, so those LDS columns keep their zeros.
__device__ __forceinline__ void knn_fetch(const uint16_t* __restrict__ db, long t, long N, int d, char* buf, int w, int lane) {
#pragma unroll
    for (int i = 0; i < KNN_DT / 4; i++) {
        const int r = w * (KNN_DT / 4) + i;
        long gr = t * KNN_DT + r;
        gr = gr < N ? gr : N - 1;
        const char* src = reinterpret_cast<const char*>(db + gr * (long)d) + lane * 16;
        char* dst = buf + r * KNN_RS;
#pragma unroll
        for (int h = 0; h < 2; h++)
            if (h * 1024 + lane * 16 < 2 * d)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + h * 1024), (lptr_t)(dst + h * 1024), 16, 0, 0);
    }
}

// One database tile: 16 RPW x 32 similarities per wave on the matrix cores, then the per-row top-k update.
// `cur` holds tile t; Sw / heap / thr are this wave's scratch rows, the kept lists and their minima.
template <bool BF, int RPW>
__device__ __forceinline__ void knn_tile(const char* cur, long t, long N, int k, int k4, int hs, const uint4 (&qf)[RPW][KNN_KSTEPS],
                                         float* Sw, float* heap, float* thr, int w, int lane) {
    constexpr int rs = KNN_RS;
    const int lr = lane & 15, lg = lane >> 4;
    const char* buf = cur;
    f32x4_t acc[RPW][2];
#pragma unroll
    for (int b = 0; b < RPW; b++) acc[b][0] = acc[b][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const char* bp = buf + lr * rs + 16 * lg;
    // B-fragments are read two k-steps ahead of the MFMAs that use them (one wave per SIMD: nothing
    // else hides the LDS latency)
    uint4 bq[3][2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        bq[s][0] = *reinterpret_cast<const uint4*>(bp + 64 * s);
        bq[s][1] = *reinterpret_cast<const uint4*>(bp + 16 * rs + 64 * s);
    }
#pragma unroll
    for (int s = 0; s < KNN_KSTEPS; s++) {
        if (s + 2 < KNN_KSTEPS) {
            bq[(s + 2) % 3][0] = *reinterpret_cast<const uint4*>(bp + 64 * (s + 2));
            bq[(s + 2) % 3][1] = *reinterpret_cast<const uint4*>(bp + 16 * rs + 64 * (s + 2));
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of this k-step's MFMAs
#pragma unroll
        for (int b = 0; b < RPW; b++) {
            mma16<BF>(acc[b][0], qf[b][s], bq[s % 3][0]);
            mma16<BF>(acc[b][1], qf[b][s], bq[s % 3][1]);
        }
    }
    // ---- top-k update.  acc[b][j][r] = <query 16(w RPW + b) + 4 lg + r, database row 32 t + 16 j + lr>
    const bool c0 = t * KNN_DT + lr < N, c1 = t * KNN_DT + 16 + lr < N;
    // bal[b][j][r]: lanes whose value beats the current k-th best of its row (rare after the first tiles)
    unsigned long long bal[RPW][2][4];
    unsigned long long any = 0;
#pragma unroll
    for (int b = 0; b < RPW; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float th = thr[(w * RPW + b) * 16 + 4 * lg + r];
            bal[b][0][r] = __ballot(c0 && acc[b][0][r] > th);
            bal[b][1][r] = __ballot(c1 && acc[b][1][r] > th);
            any |= bal[b][0][r] | bal[b][1][r];
        }
    if (any != 0) {  // wave-uniform
#pragma unroll
        for (int b = 0; b < RPW; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float* row = Sw + (b * 16 + 4 * lg + r) * KNN_SPAD;
                row[lr] = acc[b][0][r];
                row[16 + lr] = acc[b][1][r];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 16 * RPW) {
            // the owner lane of a row walks only that row's flagged columns
            const int ob = lane >> 4, orr = lane & 3, sh = 16 * ((lane & 15) >> 2);
            unsigned long long m0 = 0, m1 = 0;
#pragma unroll
            for (int b = 0; b < RPW; b++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (ob == b && orr == r) { m0 = bal[b][0][r]; m1 = bal[b][1][r]; }
            unsigned cols = (unsigned)((m0 >> sh) & 0xffffu) | ((unsigned)((m1 >> sh) & 0xffffu) << 16);
            if (cols) {
                const int qi = w * RPW * 16 + lane;
                float* h = heap + qi * hs;
                float tr = thr[qi];
                const float* row = Sw + lane * KNN_SPAD;
                while (cols) {
                    const int c = __builtin_ctz(cols);
                    cols &= cols - 1;
                    const float v = row[c];
                    if (v > tr) {
                        // replace the smallest kept value; the new threshold is the smaller of v and the runner-up
                        int at = 0;
                        float lo = INFINITY, lo2 = INFINITY;
#pragma unroll 4
                        for (int j = 0; j < k4; j += 4) {
                            const f32x4_t e = *reinterpret_cast<const f32x4_t*>(h + j);
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                if (e[u] < lo) { lo2 = lo; lo = e[u]; at = j + u; }
                                else if (e[u] < lo2) lo2 = e[u];
                            }
                        }
                        h[at] = v;
                        tr = fminf(v, lo2);
                    }
                }
                thr[qi] = tr;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// One workgroup = 4 waves x RPW blocks of 16 query rows.  LDS: two database tiles [KNN_DT][2d + 16 B],
// per-wave similarity scratch, per-row top-k lists (unsorted, with their minimum cached in thr[]).
template <bool BF, typename QE, int RPW>
__global__ __launch_bounds__(256, 1) void k_knn_mean(const typename QE::storage* __restrict__ q, long n,
                                                     const uint16_t* __restrict__ db, long N, int d, int k,
                                                     float* __restrict__ out) {
    // Two tile buffers as two LDS objects: the compiler then knows that the ds_reads of one never touch the
    // tile an LDS-DMA is still filling, and does not drain the DMA (s_waitcnt vmcnt(0)) in front of them.
    __shared__ __attribute__((aligned(16))) char tile0[KNN_DT * KNN_RS];
    __shared__ __attribute__((aligned(16))) char tile1[KNN_DT * KNN_RS];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QT = 64 * RPW;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    float* S = reinterpret_cast<float*>(smem);                   // [4][16 * RPW][KNN_SPAD]
    const int k4 = (k + 3) & ~3, hs = k4 + 4;                    // list stride: 16-byte groups + one group of padding
    float* heap = S + 4 * 16 * RPW * KNN_SPAD;                   // [QT][hs]: k kept values, +INF in the slots past k
    float* thr = heap + QT * hs;                                 // [QT]

    // ---- query rows -> unit norm (faiss.normalize_L2, score_align.py:133-134) -> MFMA A-fragments
    uint4 qf[RPW][KNN_KSTEPS];
#pragma unroll
    for (int b = 0; b < RPW; b++) {
        const long qrow = (long)blockIdx.x * QT + (w * RPW + b) * 16 + lr;
        const bool ok = qrow < n;
        const typename QE::storage* rowp = q + (ok ? qrow : 0) * (long)d;
        float ss = 0.f;
#pragma unroll
        for (int s = 0; s < KNN_KSTEPS; s++) {
            const int kel = 32 * s + 8 * lg;
            if (ok && kel < d) {
                float f[8];
                load8<QE>(rowp + kel, f);
#pragma unroll
                for (int j = 0; j < 8; j++) ss += f[j] * f[j];
            }
        }
        ss += __shfl_xor(ss, 16, SVX_WAVE);
        ss += __shfl_xor(ss, 32, SVX_WAVE);
        const float inv = ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
#pragma unroll
        for (int s = 0; s < KNN_KSTEPS; s++) {
            const int kel = 32 * s + 8 * lg;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok && kel < d) {
                float f[8];
                load8<QE>(rowp + kel, f);
                v.x = pack_pair(f[0] * inv, f[1] * inv, BF);
                v.y = pack_pair(f[2] * inv, f[3] * inv, BF);
                v.z = pack_pair(f[4] * inv, f[5] * inv, BF);
                v.w = pack_pair(f[6] * inv, f[7] * inv, BF);
            }
            qf[b][s] = v;
        }
    }
    for (int i = tid; i < KNN_DT * KNN_RS / 16; i += 256) {
        reinterpret_cast<uint4*>(tile0)[i] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4*>(tile1)[i] = make_uint4(0, 0, 0, 0);
    }
    for (int i = tid; i < QT * hs; i += 256) heap[i] = (i % hs) < k ? -INFINITY : INFINITY;
    for (int i = tid; i < QT; i += 256) thr[i] = -INFINITY;

    const long ntiles = (N + KNN_DT - 1) / KNN_DT;
    __syncthreads();  // (tiles cleared)
    if (ntiles > 0) knn_fetch(db, 0, N, d, tile0, w, lane);
    __syncthreads();

    float* Sw = S + w * 16 * RPW * KNN_SPAD;
    for (long t = 0; t < ntiles; t += 2) {
        if (t + 1 < ntiles) knn_fetch(db, t + 1, N, d, tile1, w, lane);
        knn_tile<BF, RPW>(tile0, t, N, k, k4, hs, qf, Sw, heap, thr, w, lane);
        __syncthreads();
        if (t + 1 >= ntiles) break;
        if (t + 2 < ntiles) knn_fetch(db, t + 2, N, d, tile0, w, lane);
        knn_tile<BF, RPW>(tile1, t + 1, N, k, k4, hs, qf, Sw, heap, thr, w, lane);
        __syncthreads();
    }
    if (lane < 16 * RPW) {
        const int qi = w * RPW * 16 + lane;
        const long qrow = (long)blockIdx.x * QT + qi;
        if (qrow < n) {
            const float* h = heap + qi * hs;
            float sum = 0.f;
            for (int j = 0; j < k; j++) sum += h[j];
            out[qrow] = sum / (float)k;
        }
    }
}

// score[i] = <x_i/|x_i|, y_i/|y_i|> (/ or -) (mean_xy[i] + mean_yx[i]) / 2     (score_align.py:151-160)
template <typename QE>
__global__ __launch_bounds__(256) void k_margin_scores(const typename QE::storage* __restrict__ x,
                                                       const typename QE::storage* __restrict__ y, long n, int d,
                                                       const float* __restrict__ mxy, const float* __restrict__ myx,
                                                       int margin, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const typename QE::storage* xr = x + i * (long)d;
    const typename QE::storage* yr = y + i * (long)d;
    float sxx = 0.f, syy = 0.f, sxy = 0.f;
    for (int c = lane * 8; c < d; c += 64 * 8) {
        float a[8], b[8];
        load8<QE>(xr + c, a);
        load8<QE>(yr + c, b);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            sxx += a[j] * a[j];
            syy += b[j] * b[j];
            sxy += a[j] * b[j];
        }
    }
    sxx = wave_sum(sxx);
    syy = wave_sum(syy);
    sxy = wave_sum(sxy);
    if (lane == 0) {
        const float ix = sxx > 0.f ? 1.0f / sqrtf(sxx) : 1.f, iy = syy > 0.f ? 1.0f / sqrtf(syy) : 1.f;
        const float a = sxy * ix * iy;
        const float b = (mxy[i] + myx[i]) * 0.5f;
        out[i] = margin == 0 ? a / b : a - b;
    }
}

// rows -> unit norm -> fp16 / bf16 (what populate_index keeps, prep_index.py:153-185)
template <typename QE>
__global__ __launch_bounds__(256) void k_unit_rows(const typename QE::storage* __restrict__ in, long n, int d, int bf,
                                                   uint16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const typename QE::storage* r = in + i * (long)d;
    float ss = 0.f;
    for (int c = lane * 8; c < d; c += 64 * 8) {
        float a[8];
        load8<QE>(r + c, a);
#pragma unroll
        for (int j = 0; j < 8; j++) ss += a[j] * a[j];
    }
    ss = wave_sum(ss);
    const float inv = ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
    for (int c = lane * 8; c < d; c += 64 * 8) {
        float a[8];
        load8<QE>(r + c, a);
        uint4 v;
        v.x = pack_pair(a[0] * inv, a[1] * inv, bf != 0);
        v.y = pack_pair(a[2] * inv, a[3] * inv, bf != 0);
        v.z = pack_pair(a[4] * inv, a[5] * inv, bf != 0);
        v.w = pack_pair(a[6] * inv, a[7] * inv, bf != 0);
        *reinterpret_cast<uint4*>(out + i * (long)d + c) = v;
    }
}

// ------------------------------------------------------------------------------------ launchers
static size_t knn_smem(int d, int k, int rpw) {
    return (size_t)4 * 16 * rpw * KNN_SPAD * 4 + (size_t)64 * rpw * (((k + 3) & ~3) + 5) * 4;
}

template <bool BF, typename QE, int RPW>
static int launch_knn(svx_ctx* ctx, const void* q, long n, const void* db, long N, int d, int k, float* out) {
    const size_t smem = knn_smem(d, k, RPW);
    static size_t attr_set = 0;
    if (smem > attr_set) {
        SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_knn_mean<BF, QE, RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = smem;
    }
    const long nblk = (n + 64 * RPW - 1) / (64 * RPW);
    k_knn_mean<BF, QE, RPW><<<dim3((unsigned)nblk), dim3(256), smem, ctx->stream>>>(
        reinterpret_cast<const typename QE::storage*>(q), n, reinterpret_cast<const uint16_t*>(db), N, d, k, out);
    SVX_LAUNCH_CHECK(ctx, "k_knn_mean");
    return SVX_OK;
}

template <bool BF, typename QE>
static int launch_knn_rpw(svx_ctx* ctx, const void* q, long n, const void* db, long N, int d, int k, float* out) {
    // two 16-row blocks per wave halve the LDS reads per MFMA; worth it once the grid still fills the chip
    if (n >= 128 * 256 && knn_smem(d, k, 2) + 2 * KNN_DT * KNN_RS <= 160 * 1024) return launch_knn<BF, QE, 2>(ctx, q, n, db, N, d, k, out);
    return launch_knn<BF, QE, 1>(ctx, q, n, db, N, d, k, out);
}

#define NEED(ctx, cond, ...) \
    do { if (!(cond)) return svx_fail(ctx, SVX_ERR_ARG, __VA_ARGS__); } while (0)

static int check_margin_dim(svx_ctx* ctx, int d) {
    if (d <= 0 || d % 32 != 0 || d > 32 * KNN_KSTEPS)
        return svx_fail(ctx, SVX_ERR_ARG, "embedding dimension %d: must be a positive multiple of 32, at most %d", d, 32 * KNN_KSTEPS);
    return SVX_OK;
}

extern "C" {

int svx_unit_rows(svx_ctx* ctx, const void* rows, int dtype, int64_t n, int d, void* out, int out_dtype) {
    NEED(ctx, ctx && (n == 0 || (rows && out)), "svx_unit_rows: null argument");
    NEED(ctx, out_dtype == SVX_F16 || out_dtype == SVX_BF16, "svx_unit_rows: the database is kept in fp16 or bf16 (got dtype %d)", out_dtype);
    NEED(ctx, n >= 0, "svx_unit_rows: negative row count");
    int rc = check_margin_dim(ctx, d);
    if (rc) return rc;
    if (n == 0) return SVX_OK;
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    uint16_t* o = reinterpret_cast<uint16_t*>(out);
    const int bf = out_dtype == SVX_BF16;
    switch (dtype) {
    case SVX_F32: k_unit_rows<ElemF32><<<grid, block, 0, ctx->stream>>>(reinterpret_cast<const float*>(rows), n, d, bf, o); break;
    case SVX_F16: k_unit_rows<ElemF16><<<grid, block, 0, ctx->stream>>>(reinterpret_cast<const uint16_t*>(rows), n, d, bf, o); break;
    case SVX_BF16: k_unit_rows<ElemBF16><<<grid, block, 0, ctx->stream>>>(reinterpret_cast<const uint16_t*>(rows), n, d, bf, o); break;
    default: return svx_fail(ctx, SVX_ERR_ARG, "svx_unit_rows: unknown dtype %d", dtype);
    }
    SVX_LAUNCH_CHECK(ctx, "k_unit_rows");
    return SVX_OK;
}

int svx_knn_mean_sim(svx_ctx* ctx, const void* queries, int q_dtype, int64_t n, const void* db, int db_dtype, int64_t n_db,
                     int d, int k, float* mean_sim) {
    NEED(ctx, ctx && (n == 0 || (queries && db && mean_sim)), "svx_knn_mean_sim: null argument");
    NEED(ctx, db_dtype == SVX_F16 || db_dtype == SVX_BF16, "svx_knn_mean_sim: the database is kept in fp16 or bf16 (got dtype %d)", db_dtype);
    NEED(ctx, n >= 0 && n_db >= 0, "svx_knn_mean_sim: negative row count");
    NEED(ctx, k >= 1 && k <= KNN_KMAX, "svx_knn_mean_sim: k = %d, supported 1..%d", k, KNN_KMAX);
    NEED(ctx, n_db >= k, "svx_knn_mean_sim: the database has %lld rows, fewer than k = %d", (long long)n_db, k);
    int rc = check_margin_dim(ctx, d);
    if (rc) return rc;
    if (n == 0) return SVX_OK;
    const bool bf = db_dtype == SVX_BF16;
    switch (q_dtype) {
    case SVX_F32:
        return bf ? launch_knn_rpw<true, ElemF32>(ctx, queries, n, db, n_db, d, k, mean_sim)
                  : launch_knn_rpw<false, ElemF32>(ctx, queries, n, db, n_db, d, k, mean_sim);
    case SVX_F16:
        return bf ? launch_knn_rpw<true, ElemF16>(ctx, queries, n, db, n_db, d, k, mean_sim)
                  : launch_knn_rpw<false, ElemF16>(ctx, queries, n, db, n_db, d, k, mean_sim);
    case SVX_BF16:
        return bf ? launch_knn_rpw<true, ElemBF16>(ctx, queries, n, db, n_db, d, k, mean_sim)
                  : launch_knn_rpw<false, ElemBF16>(ctx, queries, n, db, n_db, d, k, mean_sim);
    default: return svx_fail(ctx, SVX_ERR_ARG, "svx_knn_mean_sim: unknown query dtype %d", q_dtype);
    }
}

int svx_margin_scores(svx_ctx* ctx, const void* x, const void* y, int dtype, int64_t n, int d, const float* mean_xy,
                      const float* mean_yx, int margin, float* scores) {
    NEED(ctx, ctx && (n == 0 || (x && y && mean_xy && mean_yx && scores)), "svx_margin_scores: null argument");
    NEED(ctx, margin == SVX_MARGIN_RATIO || margin == SVX_MARGIN_DISTANCE, "Wrong margin type: %d", margin);
    NEED(ctx, n >= 0, "svx_margin_scores: negative row count");
    int rc = check_margin_dim(ctx, d);
    if (rc) return rc;
    if (n == 0) return SVX_OK;
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    switch (dtype) {
    case SVX_F32:
        k_margin_scores<ElemF32><<<grid, block, 0, ctx->stream>>>(reinterpret_cast<const float*>(x), reinterpret_cast<const float*>(y), n, d, mean_xy, mean_yx, margin, scores);
        break;
    case SVX_F16:
        k_margin_scores<ElemF16><<<grid, block, 0, ctx->stream>>>(reinterpret_cast<const uint16_t*>(x), reinterpret_cast<const uint16_t*>(y), n, d, mean_xy, mean_yx, margin, scores);
        break;
    case SVX_BF16:
        k_margin_scores<ElemBF16><<<grid, block, 0, ctx->stream>>>(reinterpret_cast<const uint16_t*>(x), reinterpret_cast<const uint16_t*>(y), n, d, mean_xy, mean_yx, margin, scores);
        break;
    default: return svx_fail(ctx, SVX_ERR_ARG, "svx_margin_scores: unknown dtype %d", dtype);
    }
    SVX_LAUNCH_CHECK(ctx, "k_margin_scores");
    return SVX_OK;
}

}  // extern "C"

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
#include <stdlib.h>

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

__device__ const uint4 knn_zero16 = {0u, 0u, 0u, 0u};

// One 1-KiB piece of a database tile, global -> LDS with no register stop-over (global_load_lds_dwordx4:
// every lane names its own 16 source bytes, the wave's 64 pieces land back to back at a wave-uniform LDS
// address).  Piece i of wave w is half (i & 1) of tile row w * KNN_DT / NW + (i >> 1).  Rows past the end of
// the database are read from its last row and masked out at the top-k update; lanes past the row's d elements
// copy zeros, so the LDS columns past d are always zero and the MFMA loop needs no bounds test.
template <int NW>
__device__ __forceinline__ void knn_fetch_piece(const uint16_t* __restrict__ db, long t, long N, int d, char* buf, int w, int lane, int i) {
    const int r = w * (KNN_DT / NW) + (i >> 1), h = i & 1;
    long gr = t * KNN_DT + r;
    gr = gr < N ? gr : N - 1;
    const char* src = reinterpret_cast<const char*>(db + gr * (long)d) + h * 1024 + lane * 16;
    if (h * 1024 + lane * 16 >= 2 * d) src = reinterpret_cast<const char*>(&knn_zero16);
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(buf + r * KNN_RS + h * 1024), 16, 0, 0);
}

// min over the 16 lanes that share lane >> 4, result in all of them (DPP row rotations)
__device__ __forceinline__ float row16_min(float v) {
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)));
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false)));
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false)));
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false)));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// Register top-k (k <= 16).  The kept list of query row (b, 4 lg + r) lives in hp[b][r] of the 16 lanes of
// lane group lg -- the lanes that also receive that row's similarities from the MFMA -- and th[b][r] is its
// minimum.  `pend` = lanes holding a similarity above their row's minimum.  Each round every lane group takes
// its lowest pending lane, replaces the minimum of that row's list and refreshes the minimum; the four groups
// work on four different rows at once.  Slots past k hold +INF and are never replaced.
__device__ __forceinline__ void knn_insert(float v, unsigned long long pend, float& hp, float& th, int lane) {
    const int lg = lane >> 4, lr = lane & 15;
    while (pend) {  // wave-uniform
        float cand = 0.f;
        bool active = false;
        unsigned long long chosen = 0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const unsigned m = (unsigned)(pend >> (16 * g)) & 0xffffu;
            if (m) {  // uniform
                const int src = 16 * g + __builtin_ctz(m);
                const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
                if (lg == g) { cand = c; active = true; }
                chosen |= 1ull << src;
            }
        }
        pend &= ~chosen;
        const bool doit = active && cand > th;
        const unsigned long long eq = __ballot(doit && hp == th);
        const unsigned e = (unsigned)(eq >> (16 * lg)) & 0xffffu;
        if (doit && lr == __builtin_ctz(e | 0x10000u)) hp = cand;
        th = row16_min(hp);
    }
}

// One database tile: 16 RPW x 32 similarities per wave on the matrix cores, then the per-row top-k update.
// `cur` holds tile t; the pieces of tile t + 1 are issued between the k-steps (an LDS-DMA issued among MFMAs
// costs a fraction of one issued in a burst).  KREG: lists in registers (hp / th), else in LDS (Sw / heap / thr).
template <bool BF, int RPW, int NW, bool KREG>
__device__ __forceinline__ void knn_tile(const char* cur, char* nxt, const uint16_t* __restrict__ db, long t, long N, int d, int k,
                                         int k4, int hs, const uint4 (&qf)[RPW][KNN_KSTEPS], float (&hp)[RPW][4],
                                         float (&th)[RPW][4], float* Sw, float* heap, float* thr, int w, int lane) {
    constexpr int rs = KNN_RS;
    constexpr int PIECES = 2 * KNN_DT / NW, SPP = KNN_KSTEPS / PIECES;  // pieces per wave, k-steps per piece
    const int lr = lane & 15, lg = lane >> 4;
    f32x4_t acc[RPW][2];
#pragma unroll
    for (int b = 0; b < RPW; b++) acc[b][0] = acc[b][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const char* bp = cur + lr * rs + 16 * lg;
    // B-fragments are read two k-steps ahead of the MFMAs that use them
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
        if (s % SPP == 0) knn_fetch_piece<NW>(db, t + 1, N, d, nxt, w, lane, s / SPP);
        __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of this k-step's MFMAs
#pragma unroll
        for (int b = 0; b < RPW; b++) {
            mma16<BF>(acc[b][0], qf[b][s], bq[s % 3][0]);
            mma16<BF>(acc[b][1], qf[b][s], bq[s % 3][1]);
        }
    }
    // ---- top-k update.  acc[b][j][r] = <query 16(w RPW + b) + 4 lg + r, database row 32 t + 16 j + lr>
    const bool c0 = t * KNN_DT + lr < N, c1 = t * KNN_DT + 16 + lr < N;
    if (KREG) {
        bool hit = false;
#pragma unroll
        for (int b = 0; b < RPW; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) hit |= (c0 && acc[b][0][r] > th[b][r]) || (c1 && acc[b][1][r] > th[b][r]);
        if (__any(hit)) {  // rare after the first tiles
#pragma unroll
            for (int b = 0; b < RPW; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    knn_insert(acc[b][0][r], __ballot(c0 && acc[b][0][r] > th[b][r]), hp[b][r], th[b][r], lane);
                    knn_insert(acc[b][1][r], __ballot(c1 && acc[b][1][r] > th[b][r]), hp[b][r], th[b][r], lane);
                }
        }
        return;
    }
    // bal[b][j][r]: lanes whose value beats the current k-th best of its row (rare after the first tiles)
    unsigned long long bal[RPW][2][4];
    unsigned long long any = 0;
#pragma unroll
    for (int b = 0; b < RPW; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float tv = thr[(w * RPW + b) * 16 + 4 * lg + r];
            bal[b][0][r] = __ballot(c0 && acc[b][0][r] > tv);
            bal[b][1][r] = __ballot(c1 && acc[b][1][r] > tv);
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

// One workgroup = NW waves x RPW blocks of 16 query rows, whose d-long rows stay in registers as MFMA
// A-fragments for the whole sweep over the database.  LDS: two database tiles [KNN_DT][2064 B] and, when k > 16,
// per-wave similarity scratch and per-row kept lists (unsorted, their minimum cached in thr[]).
template <bool BF, typename QE, int RPW, int NW, bool KREG>
__global__ __launch_bounds__(64 * NW, 1) void k_knn_mean(const typename QE::storage* __restrict__ q, long n,
                                                         const uint16_t* __restrict__ db, long N, int d, int k,
                                                         float* __restrict__ out, const float* st_in, float* st_out) {
    // (st_in and st_out are the SAME buffer when svx_knn_topk_merge continues its lists: no __restrict__ on them)
    // Two tile buffers as two LDS objects: the compiler then knows that the ds_reads of one never touch the
    // tile an LDS-DMA is still filling, and does not drain the DMA (s_waitcnt vmcnt(0)) in front of them.
    __shared__ __attribute__((aligned(16))) char tile0[KNN_DT * KNN_RS];
    __shared__ __attribute__((aligned(16))) char tile1[KNN_DT * KNN_RS];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QT = 16 * RPW * NW, NT = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    float* S = reinterpret_cast<float*>(smem);                   // [NW][16 * RPW][KNN_SPAD]
    const int k4 = (k + 3) & ~3, hs = k4 + 4;                    // list stride: 16-byte groups + one group of padding
    float* heap = S + NW * 16 * RPW * KNN_SPAD;                  // [QT][hs]: k kept values, +INF in the slots past k
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
    // The kept lists start empty (-INF) or, when the database comes shard by shard (svx_knn_topk_merge), from the
    // lists the sweep over the earlier shards left in st_in [n][k].
    float hp[RPW][4], th[RPW][4];
#pragma unroll
    for (int b = 0; b < RPW; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long qrow = (long)blockIdx.x * QT + (w * RPW + b) * 16 + 4 * lg + r;
            float v = lr < k ? -INFINITY : INFINITY;
            if (KREG && st_in && lr < k && qrow < n) v = st_in[qrow * k + lr];
            hp[b][r] = v;
            th[b][r] = (KREG && st_in) ? row16_min(v) : -INFINITY;
        }
    if (!KREG) {
        for (int i = tid; i < QT * hs; i += NT) {
            const int qi = i / hs, j = i % hs;
            const long qrow = (long)blockIdx.x * QT + qi;
            heap[i] = j < k ? ((st_in && qrow < n) ? st_in[qrow * k + j] : -INFINITY) : INFINITY;
        }
        __syncthreads();
        for (int qi = tid; qi < QT; qi += NT) {
            float m = -INFINITY;
            if (st_in) {
                m = INFINITY;
                for (int j = 0; j < k; j++) m = fminf(m, heap[qi * hs + j]);
            }
            thr[qi] = m;
        }
    }

    const long ntiles = (N + KNN_DT - 1) / KNN_DT;
    constexpr int PIECES = 2 * KNN_DT / NW;
    if (ntiles > 0) {
#pragma unroll
        for (int i = 0; i < PIECES; i++) knn_fetch_piece<NW>(db, 0, N, d, tile0, w, lane, i);
    }
    __syncthreads();

    float* Sw = S + w * 16 * RPW * KNN_SPAD;
    // (the last tile's step fetches "tile ntiles": clamped to the last row, never computed)
    for (long t = 0; t < ntiles; t += 2) {
        knn_tile<BF, RPW, NW, KREG>(tile0, tile1, db, t, N, d, k, k4, hs, qf, hp, th, Sw, heap, thr, w, lane);
        __syncthreads();
        if (t + 1 >= ntiles) break;
        knn_tile<BF, RPW, NW, KREG>(tile1, tile0, db, t + 1, N, d, k, k4, hs, qf, hp, th, Sw, heap, thr, w, lane);
        __syncthreads();
    }
    if (KREG) {
#pragma unroll
        for (int b = 0; b < RPW; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float sum = row16_sum(lr < k ? hp[b][r] : 0.f);
                const long qrow = (long)blockIdx.x * QT + (w * RPW + b) * 16 + 4 * lg + r;
                if (out && lr == 0 && qrow < n) out[qrow] = sum / (float)k;
                if (st_out && lr < k && qrow < n) st_out[qrow * k + lr] = hp[b][r];
            }
    } else if (lane < 16 * RPW) {
        const int qi = w * RPW * 16 + lane;
        const long qrow = (long)blockIdx.x * QT + qi;
        if (qrow < n) {
            const float* h = heap + qi * hs;
            float sum = 0.f;
            for (int j = 0; j < k; j++) sum += h[j];
            if (out) out[qrow] = sum / (float)k;
            if (st_out)
                for (int j = 0; j < k; j++) st_out[qrow * k + j] = h[j];
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
static size_t knn_smem(int k, int rpw, int nw) {
    return (size_t)nw * 16 * rpw * KNN_SPAD * 4 + (size_t)16 * nw * rpw * (((k + 3) & ~3) + 5) * 4;
}

template <bool BF, typename QE, int RPW, int NW, bool KREG>
static int launch_knn_k(svx_ctx* ctx, const void* q, long n, const void* db, long N, int d, int k, float* out, const float* st_in,
                        float* st_out) {
    const size_t smem = KREG ? 0 : knn_smem(k, RPW, NW);
    static size_t attr_set = 0;
    if (smem > attr_set) {
        SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_knn_mean<BF, QE, RPW, NW, KREG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = smem;
    }
    const long qt = 16 * RPW * NW;
    k_knn_mean<BF, QE, RPW, NW, KREG><<<dim3((unsigned)((n + qt - 1) / qt)), dim3(64 * NW), smem, ctx->stream>>>(
        reinterpret_cast<const typename QE::storage*>(q), n, reinterpret_cast<const uint16_t*>(db), N, d, k, out, st_in, st_out);
    SVX_LAUNCH_CHECK(ctx, "k_knn_mean");
    return SVX_OK;
}

template <bool BF, typename QE, int RPW, int NW>
static int launch_knn(svx_ctx* ctx, const void* q, long n, const void* db, long N, int d, int k, float* out, const float* st_in,
                      float* st_out) {
    if (k <= 16) return launch_knn_k<BF, QE, RPW, NW, true>(ctx, q, n, db, N, d, k, out, st_in, st_out);
    return launch_knn_k<BF, QE, RPW, NW, false>(ctx, q, n, db, N, d, k, out, st_in, st_out);
}

// Shapes: (RPW, NW) = (1, 4): 64 queries per workgroup, for small query sets; (2, 4): 128 queries, half the LDS
// fragment reads per MFMA, one wave per SIMD (the default for large query sets: 870 TFLOP/s at 131072^2 x 1024,
// k = 16); (1, 8): 128 queries as two waves per SIMD (854 TFLOP/s).
template <bool BF, typename QE>
static int launch_knn_rpw(svx_ctx* ctx, const void* q, long n, const void* db, long N, int d, int k, float* out,
                          const float* st_in = nullptr, float* st_out = nullptr) {
    const char* force = getenv("SVX_KNN_SHAPE");  // tuning override: "14", "18", "24"
    const int shape = force ? atoi(force) : (n >= 128 * 128 ? 24 : 14);
    const size_t tiles = (size_t)2 * KNN_DT * KNN_RS;
    if (shape == 24 && knn_smem(k, 2, 4) + tiles <= 160 * 1024) return launch_knn<BF, QE, 2, 4>(ctx, q, n, db, N, d, k, out, st_in, st_out);
    if (shape == 18 && knn_smem(k, 1, 8) + tiles <= 160 * 1024) return launch_knn<BF, QE, 1, 8>(ctx, q, n, db, N, d, k, out, st_in, st_out);
    return launch_knn<BF, QE, 1, 4>(ctx, q, n, db, N, d, k, out, st_in, st_out);
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

int svx_knn_topk_merge(svx_ctx* ctx, const void* queries, int q_dtype, int64_t n, const void* db, int db_dtype, int64_t n_db,
                       int d, int k, float* topk, int first, float* mean_sim) {
    NEED(ctx, ctx && (n == 0 || (queries && topk)) && (n_db == 0 || db), "svx_knn_topk_merge: null argument");
    NEED(ctx, db_dtype == SVX_F16 || db_dtype == SVX_BF16, "svx_knn_topk_merge: the database is kept in fp16 or bf16 (got dtype %d)", db_dtype);
    NEED(ctx, n >= 0 && n_db >= 0, "svx_knn_topk_merge: negative row count");
    NEED(ctx, k >= 1 && k <= KNN_KMAX, "svx_knn_topk_merge: k = %d, supported 1..%d", k, KNN_KMAX);
    int rc = check_margin_dim(ctx, d);
    if (rc) return rc;
    if (n == 0) return SVX_OK;
    const float* st_in = first ? nullptr : topk;
    const bool bf = db_dtype == SVX_BF16;
    switch (q_dtype) {
    case SVX_F32:
        return bf ? launch_knn_rpw<true, ElemF32>(ctx, queries, n, db, n_db, d, k, mean_sim, st_in, topk)
                  : launch_knn_rpw<false, ElemF32>(ctx, queries, n, db, n_db, d, k, mean_sim, st_in, topk);
    case SVX_F16:
        return bf ? launch_knn_rpw<true, ElemF16>(ctx, queries, n, db, n_db, d, k, mean_sim, st_in, topk)
                  : launch_knn_rpw<false, ElemF16>(ctx, queries, n, db, n_db, d, k, mean_sim, st_in, topk);
    case SVX_BF16:
        return bf ? launch_knn_rpw<true, ElemBF16>(ctx, queries, n, db, n_db, d, k, mean_sim, st_in, topk)
                  : launch_knn_rpw<false, ElemBF16>(ctx, queries, n, db, n_db, d, k, mean_sim, st_in, topk);
    default: return svx_fail(ctx, SVX_ERR_ARG, "svx_knn_topk_merge: unknown query dtype %d", q_dtype);
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

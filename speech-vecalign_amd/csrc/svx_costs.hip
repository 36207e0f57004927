// svx_costs.hip -- similarity / cost kernels (MFMA): coarse dense costs, band ("sparse") costs
// along the search path, and the sampled path scores that feed the deletion-penalty estimate.
//
// Reference semantics (paths relative to the reference repository):
//   make_dense_costs   svecalign/vecalign/dp_core.pyx:36-77
//   score_path         svecalign/vecalign/dp_core.pyx:143-161
//   make_sparse_costs  svecalign/vecalign/dp_core.pyx:165-267
//
// Design.  A cost is 2pq(1 - <u,v>)/(1e-6 + n0 + n1) with u,v unit rows.  The rows stay in their
// storage type (bf16 / fp16 / fp32, exactly representable products, fp32 accumulate in the matrix
// cores) and the unit-normalisation is applied as the two scalars 1/(||u||+1e-5), 1/(||v||+1e-5)
// in the epilogue, so normalised copies of level 0 are never written to HBM.
//
// Band kernel: one workgroup owns SVX_BC_TA consecutive path points x SVX_BC_TB band cells.  The
// cells of that chunk only touch <= 47 consecutive rows per side, so the workgroup stages those
// rows (all overlap layers) through LDS in 128-byte k-slabs and evaluates every needed
// (type, x-row, y-row) product as 16x16 MFMA tiles; the epilogue picks the band cells out of the
// accumulators, applies the cost formula in double like the reference, stages the [type][a][b]
// block in LDS and writes it out coalesced.
#include <stdlib.h>

#include "svx_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int RS = 144;        // LDS row stride in bytes: 128-byte k-slab + 16 pad (conflict-free b128 reads)

template <typename E>
struct Mma;
// Mma<E>: KS = elements per 128-byte slab, NK = k-steps per slab, frag = one lane's operand of one
// k-step.  `lane_off(lane)` is the lane's byte offset inside a tile (row lane&15, k-group lane>>4).
template <>
struct Mma<ElemBF16> {
    static constexpr int KS = 64, NK = 2, KSTEP_BYTES = 64;
    using frag = uint4;
    __device__ static __forceinline__ int lane_off(int lane) { return (lane & 15) * RS + 16 * (lane >> 4); }
    __device__ static __forceinline__ frag load(const char* p) { return *reinterpret_cast<const uint4*>(p); }
    __device__ static __forceinline__ void mma(f32x4_t& acc, const frag& a, const frag& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <>
struct Mma<ElemF16> {
    static constexpr int KS = 64, NK = 2, KSTEP_BYTES = 64;
    using frag = uint4;
    __device__ static __forceinline__ int lane_off(int lane) { return (lane & 15) * RS + 16 * (lane >> 4); }
    __device__ static __forceinline__ frag load(const char* p) { return *reinterpret_cast<const uint4*>(p); }
    __device__ static __forceinline__ void mma(f32x4_t& acc, const frag& a, const frag& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
    }
};
template <>
struct Mma<ElemF32> {
    static constexpr int KS = 32, NK = 8, KSTEP_BYTES = 16;
    using frag = float;
    __device__ static __forceinline__ int lane_off(int lane) { return (lane & 15) * RS + 4 * (lane >> 4); }
    __device__ static __forceinline__ frag load(const char* p) { return *reinterpret_cast<const float*>(p); }
    __device__ static __forceinline__ void mma(f32x4_t& acc, const frag& a, const frag& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
};

// Stage one 128-byte k-slab of `nrows` rows into LDS.  rowptr[r] = byte pointer to the row start
// in global memory or nullptr for an all-zero row.
template <typename E>
__device__ __forceinline__ void stage_slab(char* slab, const char* const* rowptr, int nrows, int k0, int d, int tid,
                                           int nthreads) {
    using S = typename E::storage;
    const int npieces = nrows * 8;
    for (int q = tid; q < npieces; q += nthreads) {
        const int r = q >> 3, p = q & 7;
        const char* ptr = rowptr[r];
        const int kel = k0 + p * E::VEC;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ptr != nullptr && kel < d) v = *reinterpret_cast<const uint4*>(ptr + (size_t)kel * sizeof(S));
        *reinterpret_cast<uint4*>(slab + r * RS + p * 16) = v;
    }
}

__device__ __forceinline__ float cost_formula(float sumx, int p, int q, float n0, float n1) {
#pragma clang fp contract(off)
    // dp_core.pyx:259-260, evaluated in double like the generated C, stored to float
    return (float)((((2.0 * (double)p) * (double)q) * (1.0 - (double)sumx)) / ((1e-6 + (double)n0) + (double)n1));
}

// ------------------------------------------------------------------------------ dense costs
struct DenseArgs {
    const void* v0;  // [s0][d] rows of the chosen layer
    const void* v1;
    int s0, s1, d;
    const float* inv0;  // [s0] or null
    const float* inv1;
    const float* n0;  // [s0]
    const float* n1;
    int mul0, mul1;  // (offset0+1), (offset1+1)
    float* costs;    // [s0][s1]
    float* dots;     // optional [s0][s1]: the normalised dot products themselves
};

// One workgroup = a DT x DT block of the cost matrix (DT = 64: every wave owns 32 x 32 = four MFMA tiles, so that a
// staged row is used against 64 rows of the other side instead of 32).
constexpr int DT = 64;
template <typename E>
__device__ void dense_block(const DenseArgs& g, int bx, int by, char* smem) {
    using S = typename E::storage;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;  // 4 waves: 2 x 2 blocks of 32
    const char** rowptr = reinterpret_cast<const char**>(smem);    // 2 * DT rows: x rows then y rows
    char* slab = smem + 2 * DT * sizeof(char*);
    const int x0 = bx * DT, y0 = by * DT;
    if (tid < 2 * DT) {
        const int side = tid >= DT, loc = tid - side * DT;
        const int gi = (side ? y0 : x0) + loc;
        const int nn = side ? g.s1 : g.s0;
        const char* base = reinterpret_cast<const char*>(side ? g.v1 : g.v0);
        rowptr[tid] = (gi < nn) ? base + (size_t)gi * g.d * sizeof(S) : nullptr;
    }
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int xt = wave >> 1, yt = wave & 1;
    for (int k0 = 0; k0 < g.d; k0 += Mma<E>::KS) {
        __syncthreads();
        stage_slab<E>(slab, rowptr, 2 * DT, k0, g.d, tid, blockDim.x);
        __syncthreads();
        const char* ap = slab + xt * 32 * RS + Mma<E>::lane_off(lane);
        const char* bp = slab + (DT + yt * 32) * RS + Mma<E>::lane_off(lane);
#pragma unroll
        for (int ks = 0; ks < Mma<E>::NK; ks++) {
            const typename Mma<E>::frag a0 = Mma<E>::load(ap + ks * Mma<E>::KSTEP_BYTES);
            const typename Mma<E>::frag a1 = Mma<E>::load(ap + 16 * RS + ks * Mma<E>::KSTEP_BYTES);
            const typename Mma<E>::frag b0 = Mma<E>::load(bp + ks * Mma<E>::KSTEP_BYTES);
            const typename Mma<E>::frag b1 = Mma<E>::load(bp + 16 * RS + ks * Mma<E>::KSTEP_BYTES);
            Mma<E>::mma(acc[0][0], a0, b0);
            Mma<E>::mma(acc[0][1], a0, b1);
            Mma<E>::mma(acc[1][0], a1, b0);
            Mma<E>::mma(acc[1][1], a1, b1);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int y = y0 + yt * 32 + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int x = x0 + xt * 32 + i * 16 + (lane >> 4) * 4 + r;
                if (x < g.s0 && y < g.s1) {
                    float sumx = acc[i][j][r];
                    if (g.inv0) sumx = sumx * g.inv0[x] * g.inv1[y];
                    float c = cost_formula(sumx, 1, 1, g.n0[x], g.n1[y]);
                    c = (c * (float)g.mul0) * (float)g.mul1;  // dp_core.pyx:75 (float * int)
                    g.costs[(size_t)x * g.s1 + y] = c;
                    if (g.dots) g.dots[(size_t)x * g.s1 + y] = sumx;
                }
            }
        }
}

template <typename E>
__global__ __launch_bounds__(256) void k_dense_costs(DenseArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    dense_block<E>(g, blockIdx.x, blockIdx.y, smem);
}

template <typename E, bool LV0>
__global__ __launch_bounds__(256) void k_dense_costs_batch(const SvxPairDev* __restrict__ pairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SvxPairDev& P = pairs[blockIdx.z];
    if ((P.L == 0) != LV0) return;
    const SvxLevel& Lv = P.lev[P.L];
    DenseArgs g;
    g.s0 = Lv.n[0];
    g.s1 = Lv.n[1];
    if ((int)blockIdx.x * DT >= g.s0 || (int)blockIdx.y * DT >= g.s1) return;
    g.d = P.d;
    g.v0 = LV0 ? P.v[0] : (const void*)Lv.P[0];
    g.v1 = LV0 ? P.v[1] : (const void*)Lv.P[1];
    g.inv0 = LV0 ? Lv.inv[0] : nullptr;
    g.inv1 = LV0 ? Lv.inv[1] : nullptr;
    g.n0 = Lv.nrm[0];
    g.n1 = Lv.nrm[1];
    g.mul0 = 1;
    g.mul1 = 1;
    g.costs = P.dcost;
    g.dots = LV0 ? nullptr : P.ddot;
    dense_block<E>(g, blockIdx.x, blockIdx.y, smem);
}

// ------------------------------------------------------------------------------ score_path
template <typename E, int NCH>
__device__ __forceinline__ float row_dot(const typename E::storage* a, const typename E::storage* b, int d, int lane) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int col = (c * SVX_WAVE + lane) * E::VEC;
        if (col < d) {
            float x[E::VEC], y[E::VEC];
            load_piece<E>(a + col, x);
            load_piece<E>(b + col, y);
#pragma unroll
            for (int i = 0; i < E::VEC; i++) s += x[i] * y[i];
        }
    }
    return wave_sum(s);
}

struct ScoreArgs {
    const void* v1;  // [rows][d]
    const void* v2;
    const float* inv1;  // or null
    const float* inv2;
    const float* n1;
    const float* n2;
    const int* xx;
    const int* yy;
    int64_t n;
    int rows1, rows2, d;
    float* out;
};

template <typename E, int NCH>
__device__ void score_block(const ScoreArgs& g, int64_t first, int64_t stride) {
    using S = typename E::storage;
    const int lane = threadIdx.x & 63;
    for (int64_t i = first; i < g.n; i += stride) {
        int xi = g.xx[i], yi = g.yy[i];
        xi = xi < 0 ? 0 : (xi >= g.rows1 ? g.rows1 - 1 : xi);  // validated on the host; never fault
        yi = yi < 0 ? 0 : (yi >= g.rows2 ? g.rows2 - 1 : yi);
        float dot = row_dot<E, NCH>(reinterpret_cast<const S*>(g.v1) + (size_t)xi * g.d,
                                    reinterpret_cast<const S*>(g.v2) + (size_t)yi * g.d, g.d, lane);
        if (lane == 0) {
            if (g.inv1) dot = dot * g.inv1[xi] * g.inv2[yi];
            const float den = g.n1[xi] + g.n2[yi];  // float add, no epsilon (dp_core.pyx:161)
            g.out[i] = (float)((2.0 * (1.0 - (double)dot)) / (double)den);
        }
    }
}

template <typename E, int NCH>
__global__ __launch_bounds__(256) void k_score_path(ScoreArgs g) {
    score_block<E, NCH>(g, (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), (int64_t)gridDim.x * 4);
}

__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned n);  // defined with the band kernels below

// Counting sort of one level's knob samples by source row (one workgroup per (level, pair)):
// korder = sample ids grouped by x, kstart[x] = first position of row x.  max and histogram do not
// care about sample order, so the scoring kernel may visit the samples row by row and read every
// source row once instead of once per sample.
__global__ __launch_bounds__(256) void k_knob_sort(const SvxPairDev* __restrict__ pairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int part[256];
    const SvxPairDev& P = pairs[blockIdx.y];
    const int level = blockIdx.x;
    if (level > P.L) return;
    if (P.L >= 1 && level == P.L) return;  // the coarsest level's samples are scored unsorted (k_knob_from_dots)
    const SvxLevel& Lv = P.lev[level];
    const int n = Lv.n[0], kn = Lv.kn, tid = threadIdx.x;
    int* cnt = reinterpret_cast<int*>(smem);  // [n]
    for (int i = tid; i < n; i += 256) cnt[i] = 0;
    __syncthreads();
    for (int i = tid; i < kn; i += 256) {
        int x = Lv.kx[i];
        x = x < 0 ? 0 : (x >= n ? n - 1 : x);
        atomicAdd(&cnt[x], 1);
    }
    __syncthreads();
    // exclusive scan: each thread owns a contiguous slice
    const int per = (n + 255) / 256;
    const int lo = tid * per, hi = (lo + per) < n ? (lo + per) : n;
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += cnt[i];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    int run = part[tid];
    for (int i = lo; i < hi; i++) {
        const int v = cnt[i];
        cnt[i] = run;  // becomes the scatter cursor
        Lv.kstart[i] = run;
        run += v;
    }
    if (tid == 0) Lv.kstart[n] = kn;
    __syncthreads();
    for (int i = tid; i < kn; i += 256) {
        int x = Lv.kx[i];
        x = x < 0 ? 0 : (x >= n ? n - 1 : x);
        const int pos = atomicAdd(&cnt[x], 1);
        const int m = Lv.n[1];
        int y = Lv.ky[i];
        y = y < 0 ? 0 : (y >= m ? m - 1 : y);
        Lv.korder[pos] = i;
        Lv.kys[pos] = y;
    }
}

// Target rows in flight per wave in k_knob_scores. Measured on the 1024-pair batch (levels >= 1, fp32): 1 -> 15.3 ms,
// 2 -> 14.9 ms, 3 -> 16.6 ms, 4 -> 16.6 ms, 6 -> 18.6 ms: past 2 the registers cost more waves than the depth gains.
constexpr int KNOB_PF = 2;  // target rows in flight per wave in k_knob_scores

// One wave per source row x: the row stays in registers while the wave walks the samples (x, y_s).
// The grid is 1-D over (pair, level) tasks x `bpt` workgroups; after the XCD remap a task's workgroups
// share one XCD, so the randomly gathered target rows of that level are served by one L2 instead of
// being spread over eight.  level = task % nlev + (LV0 ? 0 : 1).
template <typename E, int NCH, bool LV0>
__global__ __launch_bounds__(256) void k_knob_scores(const SvxPairDev* __restrict__ pairs, int nlev, int bpt) {
    using S = typename E::storage;
    constexpr int EPL = NCH * E::VEC;
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const int task = wg / bpt, blk = wg % bpt;
    const SvxPairDev& P = pairs[task / nlev];
    const int level = task % nlev + (LV0 ? 0 : 1);
    if (level > P.L) return;
    if (!LV0 && level == P.L) return;  // the coarsest level's samples read the dense stage's dots (k_knob_from_dots)
    const SvxLevel& Lv = P.lev[level];
    const int lane = threadIdx.x & 63;
    const int n = Lv.n[0], d = P.d;
    const S* v1 = LV0 ? reinterpret_cast<const S*>(P.v[0]) : reinterpret_cast<const S*>(Lv.P[0]);
    const S* v2 = LV0 ? reinterpret_cast<const S*>(P.v[1]) : reinterpret_cast<const S*>(Lv.P[1]);
    const float* inv1 = LV0 ? Lv.inv[0] : nullptr;
    const float* inv2 = LV0 ? Lv.inv[1] : nullptr;
    for (int x = blk * 4 + (threadIdx.x >> 6); x < n; x += bpt * 4) {
        const int s0 = gld(Lv.kstart + x), s1 = gld(Lv.kstart + x + 1);
        if (s0 >= s1) continue;  // wave-uniform
        float xr[EPL];
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int col = (c * SVX_WAVE + lane) * E::VEC;
            if (col < d) {
                gload_piece<E>(v1 + (size_t)x * d + col, xr + c * E::VEC);
            } else {
#pragma unroll
                for (int i = 0; i < E::VEC; i++) xr[c * E::VEC + i] = 0.f;
            }
        }
        const float nx = gld(Lv.nrm[0] + x);
        const float ix = LV0 ? gld(inv1 + x) : 1.0f;
        // The wave walks its samples with PF target rows in flight (a gather is one memory round trip per sample; the
        // register ring keeps PF of them outstanding per wave). Sample ids and target rows come 64 at a time, one per
        // lane, and are handed round with readlane, so no per-sample index load sits in front of a row load.
        constexpr int PF = KNOB_PF;
        const int ns = s1 - s0;
        for (int cb = 0; cb < ns; cb += SVX_WAVE) {
            const int cn = min(SVX_WAVE, ns - cb);
            const int ys_l = lane < cn ? gld(Lv.kys + s0 + cb + lane) : 0;
            const int is_l = lane < cn ? gld(Lv.korder + s0 + cb + lane) : 0;
            const float nrm_l = lane < cn ? gld(Lv.nrm[1] + ys_l) : 0.f;
            const float inv_l = (LV0 && lane < cn) ? gld(inv2 + ys_l) : 1.f;
            float dots_l = 0.f;
            uint4 yraw[PF][NCH];
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int yu = __builtin_amdgcn_readlane(ys_l, u);
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const int col = (c * SVX_WAVE + lane) * E::VEC;
                    yraw[u][c] = (u < cn && col < d) ? gld16(v2 + (size_t)yu * d + col) : make_uint4(0, 0, 0, 0);
                }
            }
            for (int base = 0; base < cn; base += PF) {
#pragma unroll
                for (int u = 0; u < PF; u++) {
                    const int k = base + u;  // sample k of the chunk sits in slot u (PF divides base)
                    if (k >= cn) break;      // wave-uniform
                    float dot = 0.f;
#pragma unroll
                    for (int c = 0; c < NCH; c++) {
                        float yr[E::VEC];
                        decode_piece<E>(yraw[u][c], yr);
#pragma unroll
                        for (int e = 0; e < E::VEC; e++) dot += xr[c * E::VEC + e] * yr[e];
                    }
                    if (k + PF < cn) {  // refill the slot with sample k + PF
                        const int yn = __builtin_amdgcn_readlane(ys_l, k + PF);
#pragma unroll
                        for (int c = 0; c < NCH; c++) {
                            const int col = (c * SVX_WAVE + lane) * E::VEC;
                            yraw[u][c] = col < d ? gld16(v2 + (size_t)yn * d + col) : make_uint4(0, 0, 0, 0);
                        }
                    }
                    dot = wave_sum(dot);
                    dots_l = lane == k ? dot : dots_l;
                }
            }
            // one lane per sample finishes the chunk's scores side by side
            if (lane < cn) {
                float dot = dots_l;
                if (LV0) dot = dot * ix * inv_l;
                const float den = nx + nrm_l;  // float add, no epsilon (dp_core.pyx:161)
                gst(Lv.kscore + is_l, (float)((2.0 * (1.0 - (double)dot)) / (double)den));
            }
        }
    }
}

// Documents too long for the sort's LDS histogram (> ~38 000 segments): the samples are scored in their drawn
// order, one wave per sample (both rows fetched per sample; slower, no size limit).
template <typename E, int NCH, bool LV0>
__global__ __launch_bounds__(256) void k_knob_scores_unsorted(const SvxPairDev* __restrict__ pairs) {
    const SvxPairDev& P = pairs[blockIdx.z];
    const int level = (int)blockIdx.y + (LV0 ? 0 : 1);
    if (level > P.L) return;
    if (!LV0 && level == P.L) return;  // (k_knob_from_dots)
    const SvxLevel& Lv = P.lev[level];
    ScoreArgs g;
    g.v1 = LV0 ? P.v[0] : (const void*)Lv.P[0];
    g.v2 = LV0 ? P.v[1] : (const void*)Lv.P[1];
    g.inv1 = LV0 ? Lv.inv[0] : nullptr;
    g.inv2 = LV0 ? Lv.inv[1] : nullptr;
    g.n1 = Lv.nrm[0];
    g.n2 = Lv.nrm[1];
    g.xx = Lv.kx;
    g.yy = Lv.ky;
    g.n = Lv.kn;
    g.rows1 = Lv.n[0];
    g.rows2 = Lv.n[1];
    g.d = P.d;
    g.out = Lv.kscore;
    score_block<E, NCH>(g, (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), (int64_t)gridDim.x * 4);
}

// Sampled scores of the coarsest level (L >= 1): score_path (dp_core.pyx:143-161) from the dot products that the
// dense cost stage of the same level has just computed on the matrix cores -- no row is read again.
__global__ __launch_bounds__(256) void k_knob_from_dots(const SvxPairDev* __restrict__ pairs) {
    const SvxPairDev& P = pairs[blockIdx.y];
    if (P.L < 1) return;
    const SvxLevel& Lv = P.lev[P.L];
    const int n = Lv.n[0], m = Lv.n[1];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Lv.kn; i += gridDim.x * 256) {
        int x = Lv.kx[i], y = Lv.ky[i];
        x = x < 0 ? 0 : (x >= n ? n - 1 : x);
        y = y < 0 ? 0 : (y >= m ? m - 1 : y);
        const float dot = P.ddot[(size_t)x * m + y];
        const float den = Lv.nrm[0][x] + Lv.nrm[1][y];  // float add, no epsilon (dp_core.pyx:161)
        Lv.kscore[i] = (float)((2.0 * (1.0 - (double)dot)) / (double)den);
    }
}

// ------------------------------------------------------------------------------ band costs
constexpr int TA = SVX_BC_TA, TAMAX = SVX_BC_TAMAX, TB = SVX_BC_TB, ROWS = SVX_BC_ROWS;
constexpr int CPT = (TAMAX * TB + 511) / 512;  // band cells per thread in the epilogue
constexpr int BC_THREADS = 512;
constexpr int BC_WAVES = BC_THREADS / 64;
static_assert(BC_WAVES == 8, "unit -> (wave, slot) is decoded with shifts");
constexpr int UPW_FULL = 6;                   // (type, x-tile) units per wave per pass
constexpr int TPP_FULL = BC_WAVES * UPW_FULL / 3;  // types per pass (16)
constexpr int DUMP_UNIT = 16 * ROWS;          // floats of one unit's 16 x 48 accumulator block

struct BandArgs {
    const void* v0;  // [k0][n][d]
    const void* v1;
    int n, m, d;
    const float* inv0;  // [k][n] or null
    const float* inv1;
    const float* nrm0;  // [k][n]
    const float* nrm1;
    const int* path;  // [A][2]
    int A, W;
    float* costs;  // [T][A][2W], or [A][T][2W] when atb != 0
    int* boff;     // [A]
    int* status;   // set to SVX_ERR_PATH when the path is not a unit-step lattice path
    int atb;
};

// LDS map of a band workgroup: [header: path chunk, types, row pointers, per-row scalars][work area].
// The work area holds the k-slab rows during the k loop and, afterwards, the accumulator dump and
// the output block Fs.  SW = slab width in 128-byte units (1 or 2).
struct BandLds {
    size_t off_rowptr, off_scal, off_work, work_bytes, total;
    int rs;  // slab row stride in bytes
};
__host__ __device__ inline BandLds band_lds(int kx, int ky, int sw, int ntypes_pass) {
    BandLds L;
    const int NR = (kx + ky) * ROWS;
    size_t o = (2 * TAMAX + 2 * (SVX_MAX_TYPES + 2)) * sizeof(int);  // spx, spy, tx, ty
    L.off_rowptr = (o + 15) & ~(size_t)15;
    o = L.off_rowptr + (size_t)NR * sizeof(char*);
    L.off_scal = (o + 15) & ~(size_t)15;
    o = L.off_scal + 2 * (size_t)NR * sizeof(float);  // normaliser and inverse norm per staged row
    L.off_work = (o + 15) & ~(size_t)15;
    L.rs = sw * 128 + 16;
    const size_t slab = (size_t)NR * L.rs;
    const size_t fs = (size_t)ntypes_pass * TAMAX * TB * sizeof(float);
    const size_t epi = fs + (size_t)BC_WAVES * DUMP_UNIT * sizeof(float);  // Fs + at least one dump round
    L.work_bytes = slab > epi ? slab : epi;
    L.total = L.off_work + L.work_bytes;
    return L;
}

// `live` has bit i set when the thread's i-th piece belongs to a row the chunk really reads.  Dead rows
// are neither loaded nor stored: an MFMA output element only depends on its own A row and B row, and
// the epilogue never looks at elements of dead rows, so whatever the LDS holds there is harmless.
template <typename E, int NPT, int SW>
__device__ __forceinline__ void slab_load(uint4* pre, const char* const* rowptr, unsigned live, int k0, int d, int tid) {
    using S = typename E::storage;
    constexpr int PPR = 8 * SW;  // 16-byte pieces per row
#pragma unroll
    for (int i = 0; i < NPT; i++) {
        const int q = tid + i * BC_THREADS;
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((live >> i) & 1u) {
            const int r = q / PPR, p = q % PPR;
            const int kel = k0 + p * E::VEC;
            if (kel < d) v = *reinterpret_cast<const uint4*>(rowptr[r] + (size_t)kel * sizeof(S));
        }
        pre[i] = v;
    }
}
template <int NPT, int SW>
__device__ __forceinline__ void slab_store(const uint4* pre, char* slab, unsigned live, int tid) {
    constexpr int PPR = 8 * SW, RSB = SW * 128 + 16;
#pragma unroll
    for (int i = 0; i < NPT; i++) {
        const int q = tid + i * BC_THREADS;
        if ((live >> i) & 1u) *reinterpret_cast<uint4*>(slab + (q / PPR) * RSB + (q % PPR) * 16) = pre[i];
    }
}

// NPT = 16-byte pieces a thread stages per k-slab: ceil((kx+ky)*48*8*SW / 512)
template <typename E, int NPT, int SW>
__device__ void band_block(const BandArgs& g, const SvxTypes& ty, int kx, int ky, int a0, int TAe, int chunk_b, char* smem) {
    using S = typename E::storage;
    using M = Mma<E>;
    constexpr int RSB = SW * 128 + 16;
    constexpr int KSB = M::KS * SW, NKB = M::NK * SW;
    // 512-byte slabs are only used for single-layer levels (<= 2 types): one unit per wave keeps the
    // accumulators at 12 registers so that several workgroups share a CU and hide the load latency
    constexpr int UPW = (SW == 4) ? 1 : UPW_FULL;
    constexpr int TPP = BC_WAVES * UPW / 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = 2 * g.W;
    const int b0 = chunk_b * TB;
    const int TBe = (B - b0) < TB ? (B - b0) : TB;
    const int NR = (kx + ky) * ROWS;
    const int npieces = NR * 8 * SW;
    const BandLds L = band_lds(kx, ky, SW, ty.n < TPP ? ty.n : TPP);
    int* spx = reinterpret_cast<int*>(smem);
    int* spy = spx + TAMAX;
    int* ltx = spy + TAMAX;                    // alignment type sizes (kernel-argument arrays cannot be indexed
    int* lty = ltx + (SVX_MAX_TYPES + 2);     //  dynamically without a trip through scratch)
    const char** rowptr = reinterpret_cast<const char**>(smem + L.off_rowptr);
    float* snrm = reinterpret_cast<float*>(smem + L.off_scal);  // [NR] normaliser of each staged row
    float* sinv = snrm + NR;                                     // [NR] inverse norm (1 when rows are unit)
    char* slab = smem + L.off_work;

    if (tid < TAe) {
        spx[tid] = g.path[2 * (a0 + tid)];
        spy[tid] = g.path[2 * (a0 + tid) + 1];
    }
    for (int t = tid; t < ty.n; t += BC_THREADS) {
        ltx[t] = ty.x[t];
        lty[t] = ty.y[t];
    }
    __syncthreads();
    if (tid < TAe) {
        bool ok = (spx[tid] + spy[tid] == a0 + tid);
        if (tid > 0) ok = ok && spx[tid] >= spx[tid - 1] && spy[tid] >= spy[tid - 1];
        if (!ok && g.status) *g.status = SVX_ERR_PATH;
        if (chunk_b == 0) g.boff[a0 + tid] = spy[tid] - g.W;
    }
    const int xlo = spx[0], ylo = spy[0];
    const int X0 = xlo + g.W - (b0 + TBe - 1);
    const int Y0 = ylo - g.W + b0;
    // rows the chunk's cells actually touch: (x_hi - x_lo) + TBe on the source side, same on the target
    // side; the remaining staging rows stay zero and cost no global traffic
    const int NXn = spx[TAe - 1] - xlo + TBe, NYn = spy[TAe - 1] - ylo + TBe;
    for (int r = tid; r < NR; r += BC_THREADS) {
        const int side = r >= kx * ROWS;
        const int rr = side ? r - kx * ROWS : r;
        const int layer = rr / ROWS, loc = rr % ROWS;
        const int gi = (side ? Y0 : X0) + loc;
        const int nn = side ? g.m : g.n;
        const bool live = loc < (side ? NYn : NXn) && gi >= 0 && gi < nn;
        const char* base = reinterpret_cast<const char*>(side ? g.v1 : g.v0);
        rowptr[r] = live ? base + ((size_t)layer * nn + gi) * g.d * sizeof(S) : nullptr;
        const float* nr = side ? g.nrm1 : g.nrm0;
        const float* iv = side ? g.inv1 : g.inv0;
        snrm[r] = live ? nr[(size_t)layer * nn + gi] : 0.f;
        sinv[r] = (live && iv) ? iv[(size_t)layer * nn + gi] : 1.f;
    }
    __syncthreads();
    const int loff = (lane & 15) * RSB + (int)(sizeof(S) == 4 ? 4 : 16) * (lane >> 4);
    unsigned live = 0;
#pragma unroll
    for (int i = 0; i < NPT; i++) {
        const int q = tid + i * BC_THREADS;
        if (q < npieces && rowptr[q / (8 * SW)] != nullptr) live |= 1u << i;
    }
    // 16-row tiles that hold live rows: nxt x nyt of the 3 x 3 (4 of 9 for a diagonal path)
    const int nxt = NXn <= 16 ? 1 : (NXn <= 32 ? 2 : 3);
    const int nyt = NYn <= 16 ? 1 : (NYn <= 32 ? 2 : 3);

    // this thread's output cells (CPT per thread: TAe * TBe <= TAMAX * TB)
    const int ncells = TAe * TBe;
    int c_ai[CPT], c_bi[CPT], c_xloc[CPT], c_yloc[CPT];
    bool c_in[CPT];
#pragma unroll
    for (int cc = 0; cc < CPT; cc++) {
        const int cell = tid + cc * BC_THREADS;
        c_ai[cc] = cell / TBe;
        c_bi[cc] = cell - c_ai[cc] * TBe;
        c_xloc[cc] = 0;
        c_yloc[cc] = 0;
        c_in[cc] = false;
        if (cell < ncells) {
            const int yy = spy[c_ai[cc]] - g.W + b0 + c_bi[cc];
            const int xx = (a0 + c_ai[cc]) - yy;
            c_xloc[cc] = xx - X0;  // 0 <= xloc, yloc < ROWS for a unit-step path
            c_yloc[cc] = yy - Y0;
            c_in[cc] = xx >= 0 && xx < g.n && yy >= 0 && yy < g.m;
            if (c_xloc[cc] < 0 || c_xloc[cc] >= ROWS || c_yloc[cc] < 0 || c_yloc[cc] >= ROWS) { c_xloc[cc] = 0; c_yloc[cc] = 0; c_in[cc] = false; }
        }
    }

    for (int pass = 0; pass * TPP < ty.n; pass++) {
        const int ntp = (ty.n - pass * TPP) < TPP ? (ty.n - pass * TPP) : TPP;
        const int nunits = ntp * nxt;  // (type, live x tile)
        // With at most 4 units (a single alignment type: the deeper pyramid levels) two waves share a unit
        // and split its k-steps (even / odd); the epilogue adds the two partial dumps.
        const bool ksplit = nunits <= 4;
        const int khalf = ksplit ? (wave >> 2) : 0;
        f32x4_t acc[UPW][3];
        int aoff[UPW], boffs[UPW];  // LDS byte offsets of the unit's x tile rows / its type's y layer
        int uid[UPW];
#pragma unroll
        for (int s = 0; s < UPW; s++) {
#pragma unroll
            for (int j = 0; j < 3; j++) acc[s][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            const int u = ksplit ? (s == 0 ? (wave & 3) : nunits) : wave + BC_WAVES * s;
            uid[s] = u;
            const int t = pass * TPP + (u < nunits ? u / nxt : 0);
            const int xt = u % nxt;
            aoff[s] = ((ltx[t] - 1) * ROWS + xt * 16) * RSB + loff;
            boffs[s] = ((kx + lty[t] - 1) * ROWS) * RSB + loff;
        }
        auto mma_slab = [&]() {
#pragma unroll
            for (int s = 0; s < UPW; s++) {
                if (uid[s] < nunits) {  // wave-uniform
                    const char* ap = slab + aoff[s];
                    const char* bp = slab + boffs[s];
                    if (ksplit) {
#pragma unroll
                        for (int ks2 = 0; ks2 < NKB / 2; ks2++) {
                            const int ko = (2 * ks2 + khalf) * M::KSTEP_BYTES;
                            const typename M::frag fa = M::load(ap + ko);
#pragma unroll
                            for (int j = 0; j < 3; j++)
                                if (j < nyt) M::mma(acc[s][j], fa, M::load(bp + j * 16 * RSB + ko));
                        }
                    } else {
#pragma unroll
                        for (int ks = 0; ks < NKB; ks++) {
                            const typename M::frag fa = M::load(ap + ks * M::KSTEP_BYTES);
#pragma unroll
                            for (int j = 0; j < 3; j++)
                                if (j < nyt) M::mma(acc[s][j], fa, M::load(bp + j * 16 * RSB + ks * M::KSTEP_BYTES));
                        }
                    }
                }
            }
        };
        if constexpr (SW == 1) {
            // 128-byte slabs: two register sets, so every global load has two slab periods to land
            // (slab k in LDS, slab k+1 in flight in one set, slab k+2 being issued into the other)
            uint4 preA[NPT], preB[NPT];
            slab_load<E, NPT, SW>(preA, rowptr, live, 0, g.d, tid);
            __syncthreads();  // the previous pass is done with the work area
            slab_store<NPT, SW>(preA, slab, live, tid);
            if (KSB < g.d) slab_load<E, NPT, SW>(preA, rowptr, live, KSB, g.d, tid);
            __syncthreads();
            for (int k0 = 0; k0 < g.d; k0 += 2 * KSB) {
                if (k0 + 2 * KSB < g.d) slab_load<E, NPT, SW>(preB, rowptr, live, k0 + 2 * KSB, g.d, tid);
                __builtin_amdgcn_sched_barrier(0);
                mma_slab();
                __syncthreads();  // every wave has read this slab
                if (k0 + KSB < g.d) {
                    slab_store<NPT, SW>(preA, slab, live, tid);
                    __syncthreads();
                    if (k0 + 3 * KSB < g.d) slab_load<E, NPT, SW>(preA, rowptr, live, k0 + 3 * KSB, g.d, tid);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_slab();
                    __syncthreads();
                    if (k0 + 2 * KSB < g.d) slab_store<NPT, SW>(preB, slab, live, tid);
                    __syncthreads();
                }
            }
        } else {
            // wide slabs: the global loads of slab k+1 are issued before the MFMAs of slab k
            uint4 pre[NPT];
            slab_load<E, NPT, SW>(pre, rowptr, live, 0, g.d, tid);
            __syncthreads();  // the previous pass is done with the work area
            slab_store<NPT, SW>(pre, slab, live, tid);
            __syncthreads();
            for (int k0 = 0; k0 < g.d; k0 += KSB) {
                const bool more = k0 + KSB < g.d;
                if (more) slab_load<E, NPT, SW>(pre, rowptr, live, k0 + KSB, g.d, tid);
                __builtin_amdgcn_sched_barrier(0);
                mma_slab();
                __syncthreads();  // every wave has read this slab
                if (more) slab_store<NPT, SW>(pre, slab, live, tid);
                __syncthreads();
            }
        }
        // epilogue.  Work area = Fs [ai][type][bi] followed by the accumulator dump [unit][16][48].
        // Rounds of SR unit slots: dump -> every thread looks its cell up in each dumped unit whose x tile
        // contains the cell's row -> cost formula -> Fs.  Finally Fs is written out coalesced.
        float* Fs = reinterpret_cast<float*>(slab);
        float* dump = Fs + (size_t)ntp * TAMAX * TB;
        const int du = (int)((L.work_bytes - (size_t)ntp * TAMAX * TB * sizeof(float)) / (DUMP_UNIT * sizeof(float)));
        const int SR = du / BC_WAVES >= UPW ? UPW : (du / BC_WAVES < 1 ? 1 : du / BC_WAVES);
        const int nslots = (nunits + BC_WAVES - 1) / BC_WAVES;
        for (int s0 = 0; s0 < nslots; s0 += SR) {
#pragma unroll
            for (int s = 0; s < UPW; s++) {
                if (s >= s0 && s < s0 + SR && uid[s] < nunits) {
                    float* dw = dump + ((s - s0) * BC_WAVES + wave) * DUMP_UNIT;
#pragma unroll
                    for (int yt = 0; yt < 3; yt++)
#pragma unroll
                        for (int r = 0; r < 4; r++) dw[((lane >> 4) * 4 + r) * ROWS + yt * 16 + (lane & 15)] = acc[s][yt][r];
                }
            }
            __syncthreads();
#pragma unroll
            for (int cc = 0; cc < CPT; cc++) {
                if (tid + cc * BC_THREADS >= ncells) continue;
                const int xq = c_xloc[cc] >> 4;
                const int ulo = s0 * BC_WAVES, uhi = (s0 + SR) * BC_WAVES < nunits ? (s0 + SR) * BC_WAVES : nunits;
                // units of this round whose x tile holds my row: u = nxt*tl + xq
                for (int tl = (ulo - xq + nxt - 1) / nxt; nxt * tl + xq < uhi; tl++) {
                    const int u = nxt * tl + xq;
                    const int t = pass * TPP + tl;
                    const int p = ltx[t], q = lty[t];
                    float c = __builtin_inff();
                    if (c_in[cc]) {
                        const int rx = (p - 1) * ROWS + c_xloc[cc], ry = (kx + q - 1) * ROWS + c_yloc[cc];
                        float dsum = dump[(u - ulo) * DUMP_UNIT + (c_xloc[cc] & 15) * ROWS + c_yloc[cc]];
                        if (ksplit) dsum += dump[(u - ulo + 4) * DUMP_UNIT + (c_xloc[cc] & 15) * ROWS + c_yloc[cc]];
                        const float sumx = dsum * sinv[rx] * sinv[ry];
                        c = cost_formula(sumx, p, q, snrm[rx], snrm[ry]);
                    }
                    Fs[((size_t)c_ai[cc] * ntp + tl) * TB + c_bi[cc]] = c;
                }
            }
            __syncthreads();
        }
        // coalesced write-out of the [ai][type][bi] block
        const int nout = TAe * ntp * TBe;
        for (int idx = tid; idx < nout; idx += BC_THREADS) {
            const int ai = idx / (ntp * TBe);
            const int rem = idx - ai * (ntp * TBe);
            const int tl = rem / TBe, bi = rem - tl * TBe;
            const int t = pass * TPP + tl;
            const size_t o = g.atb ? ((size_t)(a0 + ai) * ty.n + t) * B + (b0 + bi) : ((size_t)t * g.A + (a0 + ai)) * B + (b0 + bi);
            g.costs[o] = Fs[((size_t)ai * ntp + tl) * TB + bi];
        }
    }
}

template <typename E, int NPT, int SW>
__global__ __launch_bounds__(BC_THREADS) void k_band_costs(BandArgs g, SvxTypes ty, int kx, int ky, int nchunk_b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int a0 = (blockIdx.x / nchunk_b) * TA;
    band_block<E, NPT, SW>(g, ty, kx, ky, a0, (g.A - a0) < TA ? (g.A - a0) : TA, blockIdx.x % nchunk_b, smem);
}

// depth 0 uses the final types on the raw rows; deeper levels use (1,1) on normalised fp32 layer 0
// depth 0 uses the final types on the raw rows; deeper levels use (1,1) on normalised fp32 layer 0
// Workgroups are dealt round-robin over the 8 XCDs (id % 8).  Remap the linear id so that each XCD
// owns one contiguous range of (pair, chunk) work items: neighbouring chunks share up to 2W rows per
// side, and with this map they share them through the same L2 (speed only, never correctness).
__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned n) {
    const unsigned q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// depth 0 uses the final types on the raw rows; deeper levels use (1,1) on normalised fp32 layer 0
template <typename E, bool LV0, int NPT, int SW>
__global__ __launch_bounds__(BC_THREADS) void k_band_costs_batch(const SvxPairDev* __restrict__ pairs, int depth, SvxTypes ty,
                                                                 int kx, int ky, int W, int nchunk_b, int per_pair) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const SvxPairDev& P = pairs[wg / per_pair];
    const int item = wg % per_pair;
    if (depth > P.L || (depth == P.L && P.L > 0)) return;  // refined levels only (or level 0 when L == 0)
    const SvxLevel& Lv = P.lev[depth];
    BandArgs g;
    g.A = *Lv.path_len;
    const int chunk_a = item / nchunk_b;
    if (g.A <= 0 || chunk_a >= *Lv.nchunks) return;
    const int a0 = Lv.cstart[chunk_a], TAe = Lv.cstart[chunk_a + 1] - a0;
    g.n = Lv.n[0];
    g.m = Lv.n[1];
    g.d = P.d;
    g.v0 = LV0 ? P.v[0] : (const void*)Lv.P[0];
    g.v1 = LV0 ? P.v[1] : (const void*)Lv.P[1];
    g.inv0 = LV0 ? Lv.inv[0] : nullptr;
    g.inv1 = LV0 ? Lv.inv[1] : nullptr;
    g.nrm0 = Lv.nrm[0];
    g.nrm1 = Lv.nrm[1];
    g.path = Lv.path;
    g.W = W;
    g.costs = Lv.costs;
    g.boff = Lv.boff;
    g.status = P.status;
    g.atb = 1;
    band_block<E, NPT, SW>(g, ty, kx, ky, a0, TAe, item % nchunk_b, smem);
}

inline int nch_f32(int d) {
    int c = (d + 255) / 256;
    return c <= 1 ? 1 : c <= 2 ? 2 : c <= 4 ? 4 : 8;
}
inline int nch_16(int d) {
    int c = (d + 511) / 512;
    return c <= 1 ? 1 : c <= 2 ? 2 : 4;
}

constexpr size_t DENSE_SMEM = 2 * DT * sizeof(char*) + 2 * DT * RS;

}  // namespace

int svxl_dense_costs(svx_ctx* ctx, const float* v0, int s0, const float* v1, int s1, int d, const float* n0,
                     const float* n1, int mul0, int mul1, float* costs) {
    if (s0 <= 0 || s1 <= 0) return SVX_OK;
    DenseArgs g{v0, v1, s0, s1, d, nullptr, nullptr, n0, n1, mul0, mul1, costs, nullptr};
    hipLaunchKernelGGL(k_dense_costs<ElemF32>, dim3((s0 + DT - 1) / DT, (s1 + DT - 1) / DT), dim3(256), DENSE_SMEM, ctx->stream, g);
    SVX_LAUNCH_CHECK(ctx, "k_dense_costs");
    return SVX_OK;
}

int svxl_dense_costs_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int max_s0, int max_s1, int dtype, int d) {
    (void)d;
    if (n_pairs <= 0 || max_s0 <= 0 || max_s1 <= 0) return SVX_OK;
    dim3 grid((max_s0 + DT - 1) / DT, (max_s1 + DT - 1) / DT, n_pairs);
    hipLaunchKernelGGL((k_dense_costs_batch<ElemF32, false>), grid, dim3(256), DENSE_SMEM, ctx->stream, pairs);
    if (dtype == SVX_F32)
        hipLaunchKernelGGL((k_dense_costs_batch<ElemF32, true>), grid, dim3(256), DENSE_SMEM, ctx->stream, pairs);
    else if (dtype == SVX_F16)
        hipLaunchKernelGGL((k_dense_costs_batch<ElemF16, true>), grid, dim3(256), DENSE_SMEM, ctx->stream, pairs);
    else
        hipLaunchKernelGGL((k_dense_costs_batch<ElemBF16, true>), grid, dim3(256), DENSE_SMEM, ctx->stream, pairs);
    SVX_LAUNCH_CHECK(ctx, "k_dense_costs_batch");
    return SVX_OK;
}

int svxl_score_path(svx_ctx* ctx, const int* xx, const int* yy, int64_t n, const float* n1, const float* n2,
                    const float* v1, int rows1, const float* v2, int rows2, int d, float* out) {
    if (n <= 0) return SVX_OK;
    ScoreArgs g{v1, v2, nullptr, nullptr, n1, n2, xx, yy, n, rows1, rows2, d, out};
    int64_t nb = (n + 3) / 4;
    if (nb > 16384) nb = 16384;
#define M(N) hipLaunchKernelGGL((k_score_path<ElemF32, N>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, g)
    switch (nch_f32(d)) {
        case 1: M(1); break;
        case 2: M(2); break;
        case 4: M(4); break;
        default: M(8); break;
    }
#undef M
    SVX_LAUNCH_CHECK(ctx, "k_score_path");
    return SVX_OK;
}

int svxl_knob_from_dots(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int max_kn) {
    if (n_pairs <= 0 || max_kn <= 0) return SVX_OK;
    int nb = (max_kn + 255) / 256;
    if (nb > 16) nb = 16;
    hipLaunchKernelGGL(k_knob_from_dots, dim3(nb, n_pairs), dim3(256), 0, ctx->stream, pairs);
    SVX_LAUNCH_CHECK(ctx, "k_knob_from_dots");
    return SVX_OK;
}

// part 0: counting sort by source row; part 1: levels >= 1 (fp32 rows); part 2: level 0 (input dtype)
int svxl_knob_scores(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int max_L, int max_kn, int max_n0, int dtype, int d,
                     int part) {
    if (n_pairs <= 0 || max_kn <= 0) return SVX_OK;
    hipStream_t st = ctx->stream;
    const bool unsorted = (size_t)(max_n0 + 1) * sizeof(int) > 150 * 1024;  // the sort's histogram does not fit LDS
    if (unsorted) {
        if (part == 0) return SVX_OK;
        const int nbu = (max_kn + 3) / 4 < 2048 ? (max_kn + 3) / 4 : 2048;
#define U32(N, LV0, GY) hipLaunchKernelGGL((k_knob_scores_unsorted<ElemF32, N, LV0>), dim3(nbu, GY, n_pairs), dim3(256), 0, st, pairs)
#define U16(E, N) hipLaunchKernelGGL((k_knob_scores_unsorted<E, N, true>), dim3(nbu, 1, n_pairs), dim3(256), 0, st, pairs)
        if (part == 1) {
            if (max_L >= 1) {
                switch (nch_f32(d)) {
                    case 1: U32(1, false, max_L); break;
                    case 2: U32(2, false, max_L); break;
                    case 4: U32(4, false, max_L); break;
                    default: U32(8, false, max_L); break;
                }
            }
        } else if (dtype == SVX_F32) {
            switch (nch_f32(d)) {
                case 1: U32(1, true, 1); break;
                case 2: U32(2, true, 1); break;
                case 4: U32(4, true, 1); break;
                default: U32(8, true, 1); break;
            }
        } else if (dtype == SVX_F16) {
            switch (nch_16(d)) {
                case 1: U16(ElemF16, 1); break;
                case 2: U16(ElemF16, 2); break;
                default: U16(ElemF16, 4); break;
            }
        } else {
            switch (nch_16(d)) {
                case 1: U16(ElemBF16, 1); break;
                case 2: U16(ElemBF16, 2); break;
                default: U16(ElemBF16, 4); break;
            }
        }
#undef U32
#undef U16
        SVX_LAUNCH_CHECK(ctx, "k_knob_scores_unsorted");
        return SVX_OK;
    }
    if (part == 0) {
        const size_t smem = (size_t)(max_n0 + 1) * sizeof(int);
        if (smem > 64 * 1024)
            SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_knob_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(k_knob_sort, dim3(max_L + 1, n_pairs), dim3(256), smem, st, pairs);
        SVX_LAUNCH_CHECK(ctx, "k_knob_sort");
        return SVX_OK;
    }
    int nb = (max_n0 + 3) / 4;  // one wave per source row
    if (nb > 1024) nb = 1024;
#define M32(N, LV0, GY) hipLaunchKernelGGL((k_knob_scores<ElemF32, N, LV0>), dim3((unsigned)nb * (GY) * n_pairs), dim3(256), 0, st, pairs, GY, nb)
#define M16(E, N) hipLaunchKernelGGL((k_knob_scores<E, N, true>), dim3((unsigned)nb * n_pairs), dim3(256), 0, st, pairs, 1, nb)
    if (part == 1 && max_L >= 1) {
        switch (nch_f32(d)) {
            case 1: M32(1, false, max_L); break;
            case 2: M32(2, false, max_L); break;
            case 4: M32(4, false, max_L); break;
            default: M32(8, false, max_L); break;
        }
    }
    if (part != 2) {
    } else if (dtype == SVX_F32) {
        switch (nch_f32(d)) {
            case 1: M32(1, true, 1); break;
            case 2: M32(2, true, 1); break;
            case 4: M32(4, true, 1); break;
            default: M32(8, true, 1); break;
        }
    } else if (dtype == SVX_F16) {
        switch (nch_16(d)) {
            case 1: M16(ElemF16, 1); break;
            case 2: M16(ElemF16, 2); break;
            default: M16(ElemF16, 4); break;
        }
    } else {
        switch (nch_16(d)) {
            case 1: M16(ElemBF16, 1); break;
            case 2: M16(ElemBF16, 2); break;
            default: M16(ElemBF16, 4); break;
        }
    }
#undef M32
#undef M16
    SVX_LAUNCH_CHECK(ctx, "k_knob_scores");
    return SVX_OK;
}

static int types_layers(const SvxTypes& ty, int* kx, int* ky) {
    int mx = 0, my = 0;
    for (int t = 0; t < ty.n; t++) {
        if (ty.x[t] > mx) mx = ty.x[t];
        if (ty.y[t] > my) my = ty.y[t];
    }
    *kx = mx;
    *ky = my;
    return 0;
}

struct BandPlan {
    int sw, npt;
    size_t smem;
};

// Slab width and staging depth for a layer count: the widest k-slab the LDS budget allows (512 bytes for
// single-layer levels, else 256, else 128): wider slabs mean fewer barrier rounds per workgroup.
static int band_plan(svx_ctx* ctx, int kx, int ky, int ntypes, BandPlan* bp) {
    const int NR = (kx + ky) * ROWS;
    const char* sw_env = getenv("SVX_BAND_SW");  // tuning override: widest slab to consider
    const int sw_max = sw_env ? atoi(sw_env) : 4;
    for (int sw = 4; sw >= 1; sw >>= 1) {
        if (sw > sw_max) continue;
        const int tpp = sw == 4 ? BC_WAVES / 3 : TPP_FULL;
        const int ntp = ntypes < tpp ? ntypes : tpp;
        if (sw == 4 && ntypes > tpp) continue;
        const BandLds L = band_lds(kx, ky, sw, ntp);
        const int npt = (NR * 8 * sw + BC_THREADS - 1) / BC_THREADS;
        if (sw == 4 && (NR > 2 * ROWS || L.total > 64 * 1024)) continue;  // 512-byte slabs: single-layer levels only
        if (sw == 2 && (L.total > 128 * 1024 || npt > 12)) continue;
        if (L.total > 160 * 1024 || npt > 14)
            return svx_fail(ctx, SVX_ERR_ARG, "band costs: %d+%d overlap layers need %zu bytes of LDS / %d staging pieces (limits 160 KiB / 14)",
                            kx, ky, L.total, npt);
        bp->sw = sw;
        bp->npt = npt < 1 ? 1 : npt;
        bp->smem = L.total;
        return SVX_OK;
    }
    return svx_fail(ctx, SVX_ERR_ARG, "band costs: no launch configuration");
}

#define BAND_DISPATCH(CALL)                      \
    do {                                         \
        if (bp.sw == 4) {                        \
            CALL(6, 4);                          \
        } else if (bp.sw == 2) {                 \
            if (bp.npt <= 3) CALL(3, 2);         \
            else if (bp.npt <= 6) CALL(6, 2);    \
            else CALL(12, 2);                    \
        } else {                                 \
            if (bp.npt <= 2) CALL(2, 1);         \
            else if (bp.npt <= 6) CALL(6, 1);    \
            else if (bp.npt <= 8) CALL(8, 1);    \
            else CALL(14, 1);                    \
        }                                        \
    } while (0)

int svxl_band_costs(svx_ctx* ctx, const void* v0, int k0, int n, const void* v1, int k1, int m, int d, int dtype,
                    const float* inv0, const float* inv1, const float* nrm0, const float* nrm1, const int* path, int A,
                    const SvxTypes& types, int W, float* costs, int* boff, int* status) {
    if (A <= 0) return SVX_OK;
    int kx, ky;
    types_layers(types, &kx, &ky);
    if (kx > k0 || ky > k1) return svx_fail(ctx, SVX_ERR_OVERLAPS, "overlaps");
    if (types.n == 0) kx = ky = 0;  // only b_offset is produced (costs has shape [0][A][B])
    BandPlan bp;
    int rc = band_plan(ctx, kx, ky, types.n, &bp);
    if (rc) return rc;
    const size_t smem = bp.smem;
    const int B = 2 * W;
    const int nca = (A + TA - 1) / TA, ncb = (B + TB - 1) / TB;
    BandArgs g{v0, v1, n, m, d, inv0, inv1, nrm0, nrm1, path, A, W, costs, boff, status, 0};
    dim3 grid((unsigned)(nca * ncb));
#define CALL_E(E, NPT, SW)                                                                                          \
    do {                                                                                                            \
        if (smem > 64 * 1024)                                                                                       \
            SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_band_costs<E, NPT, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((k_band_costs<E, NPT, SW>), grid, dim3(BC_THREADS), smem, ctx->stream, g, types, kx, ky, ncb); \
    } while (0)
#define CALL_F32(NPT, SW) CALL_E(ElemF32, NPT, SW)
#define CALL_F16(NPT, SW) CALL_E(ElemF16, NPT, SW)
#define CALL_BF16(NPT, SW) CALL_E(ElemBF16, NPT, SW)
    if (dtype == SVX_F32) BAND_DISPATCH(CALL_F32);
    else if (dtype == SVX_F16) BAND_DISPATCH(CALL_F16);
    else BAND_DISPATCH(CALL_BF16);
#undef CALL_F32
#undef CALL_F16
#undef CALL_BF16
#undef CALL_E
    SVX_LAUNCH_CHECK(ctx, "k_band_costs");
    return SVX_OK;
}

int svxl_band_costs_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int depth, int max_A, const SvxTypes& types,
                          int W, int dtype, int d) {
    (void)d;
    if (n_pairs <= 0 || max_A <= 0) return SVX_OK;
    int kx, ky;
    types_layers(types, &kx, &ky);
    if (types.n == 0) kx = ky = 0;
    BandPlan bp;
    int rc = band_plan(ctx, kx, ky, types.n, &bp);
    if (rc) return rc;
    const size_t smem = bp.smem;
    const int B = 2 * W;
    const int nca = (max_A + TA - 1) / TA, ncb = (B + TB - 1) / TB;
    const int per_pair = nca * ncb;
    dim3 grid((unsigned)per_pair * (unsigned)n_pairs);
#define CALL_E(E, LV0, NPT, SW)                                                                                     \
    do {                                                                                                            \
        if (smem > 64 * 1024)                                                                                       \
            SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_band_costs_batch<E, LV0, NPT, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((k_band_costs_batch<E, LV0, NPT, SW>), grid, dim3(BC_THREADS), smem, ctx->stream, pairs, depth, types, kx, ky, W, ncb, per_pair); \
    } while (0)
#define CALL_DEEP(NPT, SW) CALL_E(ElemF32, false, NPT, SW)
#define CALL_F32(NPT, SW) CALL_E(ElemF32, true, NPT, SW)
#define CALL_F16(NPT, SW) CALL_E(ElemF16, true, NPT, SW)
#define CALL_BF16(NPT, SW) CALL_E(ElemBF16, true, NPT, SW)
    if (depth > 0) BAND_DISPATCH(CALL_DEEP);
    else if (dtype == SVX_F32) BAND_DISPATCH(CALL_F32);
    else if (dtype == SVX_F16) BAND_DISPATCH(CALL_F16);
    else BAND_DISPATCH(CALL_BF16);
#undef CALL_DEEP
#undef CALL_F32
#undef CALL_F16
#undef CALL_BF16
#undef CALL_E
    SVX_LAUNCH_CHECK(ctx, "k_band_costs_batch");
    return SVX_OK;
}

// svx_rows.hip -- row-streaming kernels: unit-normalisation, pyramid down-sampling, column
// means, sampled-row means and the per-row normalisers ("norms") of dp_utils.py.
//
// All of these are HBM-bound streams over [layers][rows][d] tensors.  One wave owns one row at a
// time: lane l holds the 16-byte pieces (chunk*64 + l) of the row, so every wave-instruction
// moves one contiguous KiB; reductions over d are wave butterflies, nothing goes through LDS
// except the per-workgroup column-sum hand-off.
//
// Reference semantics (paths relative to the reference repository):
//   make_norm1            svecalign/vecalign/dp_utils.py:32-40
//   downsample_vectors    svecalign/vecalign/dp_utils.py:362-378
//   compute_norms         svecalign/vecalign/dp_utils.py:326-359
// compute_norms is evaluated as 1 - <row, mean_s(sample_s)> (one dot per row instead of one
// GEMM column per sample); the two are equal in exact arithmetic and agree to ~1e-7 in float32.
#include "svx_common.h"

namespace {

template <typename E, int NCH>
struct Row {
    static constexpr int VEC = E::VEC;
    static constexpr int EPL = NCH * VEC;  // elements per lane
    using S = typename E::storage;

    __device__ static __forceinline__ void load(const S* row, int d, int lane, float* x) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            int col = (c * SVX_WAVE + lane) * VEC;
            if (col < d) {
                gload_piece<E>(row + col, x + c * VEC);
            } else {
#pragma unroll
                for (int i = 0; i < VEC; i++) x[c * VEC + i] = 0.f;
            }
        }
    }
    // float32 vector (mean / rbar / output rows) addressed with the same column map
    __device__ static __forceinline__ void loadf(const float* v, int d, int lane, float* x) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            int col = (c * SVX_WAVE + lane) * VEC;
            if (col < d) {
#pragma unroll
                for (int q = 0; q < VEC / 4; q++) {
                    float4 t = gldf4(v + col + 4 * q);
                    x[c * VEC + 4 * q + 0] = t.x; x[c * VEC + 4 * q + 1] = t.y;
                    x[c * VEC + 4 * q + 2] = t.z; x[c * VEC + 4 * q + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < VEC; i++) x[c * VEC + i] = 0.f;
            }
        }
    }
    // FULL: the row fills the lane map exactly (d == NCH * 64 * VEC), no lane is ever out of range -- without the test
    // the compiler needs no exec-masked branch (and no register copies to merge its two sides) around every piece
    template <bool FULL = false>
    __device__ static __forceinline__ void storef(float* v, int d, int lane, const float* x) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            int col = (c * SVX_WAVE + lane) * VEC;
            if (FULL || col < d) {
#pragma unroll
                for (int q = 0; q < VEC / 4; q++) {
                    gstf4(v + col + 4 * q, x[c * VEC + 4 * q + 0], x[c * VEC + 4 * q + 1], x[c * VEC + 4 * q + 2], x[c * VEC + 4 * q + 3]);
                }
            }
        }
    }
    // streaming store: written once, read by a later kernel after tens of GB of other traffic
    template <bool FULL = false>
    __device__ static __forceinline__ void storef_nt(float* v, int d, int lane, const float* x) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            int col = (c * SVX_WAVE + lane) * VEC;
            if (FULL || col < d) {
#pragma unroll
                for (int q = 0; q < VEC / 4; q++) {
                    gstf4_nt(v + col + 4 * q, x[c * VEC + 4 * q + 0], x[c * VEC + 4 * q + 1], x[c * VEC + 4 * q + 2], x[c * VEC + 4 * q + 3]);
                }
            }
        }
    }
    // lane-register index e  ->  column
    __device__ static __forceinline__ int col_of(int e, int lane) {
        return ((e / VEC) * SVX_WAVE + lane) * VEC + (e % VEC);
    }
};

// One workgroup (4 waves) walks SVX_PYR_SLOTS row-pair slots of one layer:
//   x = row - mean (if mean); den = ||x|| + 1e-5; vn = x / den            (make_norm1)
//   nrm = 1 - <vn, rbar>                                                   (compute_norms)
//   next[slot] = vn[2 slot] + vn[2 slot + 1]; part = column sums of next   (downsample_vectors)
// Row r of level 1 straight from the level-0 rows: vn0[2r] + vn0[2r+1] with vn0 = row * 1/(||row|| + 1e-5) -- the
// same two multiplications and one addition that the level-0 pass used to store (no contraction into an fma).
template <typename E, int NCH>
__device__ __forceinline__ void pair_row(const typename E::storage* rows0, const float* inv0, int r, int d, int lane, float* out) {
#pragma clang fp contract(off)
    using R = Row<E, NCH>;
    float a[R::EPL], b[R::EPL];
    R::load(rows0 + (size_t)(2 * r) * d, d, lane, a);
    R::load(rows0 + (size_t)(2 * r + 1) * d, d, lane, b);
    const float ia = gld(inv0 + 2 * r), ib = gld(inv0 + 2 * r + 1);
#pragma unroll
    for (int e = 0; e < R::EPL; e++) {
        const float pa = a[e] * ia, pb = b[e] * ib;
        out[e] = pa + pb;
    }
}

// One float32 row out of lane registers laid out by the 16-BIT column map (a lane owns 8 consecutive elements = 32 bytes
// of float32): stored straight from the registers, an instruction's 64 lanes write 16-byte pieces at a 32-byte stride --
// half lines -- and the pass streams at 4.7 TB/s instead of 6.0 (profiles/micro/row_stream.hip, last variant).  Through
// `wbuf` (EPL * 64 floats of LDS private to the wave: its slot of the column-sum area, unused until the epilogue) the
// row is re-dealt so that every store instruction writes 1 KB contiguous.  No bank conflicts either way.
template <typename R, bool FULL, bool NT>
__device__ __forceinline__ void store_row_via_lds(float* dst, int d, int lane, const float* x, float* wbuf) {
    constexpr int NCH = R::EPL / R::VEC;
    static_assert(R::VEC == 8, "16-bit column map");
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        float4* w4 = reinterpret_cast<float4*>(wbuf + 512 * c + 8 * lane);
        w4[0] = make_float4(x[8 * c + 0], x[8 * c + 1], x[8 * c + 2], x[8 * c + 3]);
        w4[1] = make_float4(x[8 * c + 4], x[8 * c + 5], x[8 * c + 6], x[8 * c + 7]);
    }
#pragma unroll
    for (int g = 0; g < 2 * NCH; g++) {
        const int col = 256 * g + 4 * lane;
        const float4 v = *reinterpret_cast<const float4*>(wbuf + col);
        if (FULL || col < d) {
            if (NT) gstf4_nt(dst + col, v.x, v.y, v.z, v.w);
            else gstf4(dst + col, v.x, v.y, v.z, v.w);
        }
    }
}

// LDS floats a pyramid workgroup needs: mean and rbar (lane-major), the four waves' column sums, the arrival counter.
#define SVX_PYR_LDS_FLOATS(NCH, VEC) (6 * (NCH) * (VEC) * SVX_WAVE + 4)

// Column-sum partial of a workgroup WITHOUT a barrier: every wave drops its sums into LDS and draws a ticket; the
// wave that draws the last one adds the four in wave order (so the partial is deterministic) and stores the row.
// A barrier here kept three finished waves -- and their registers -- parked until the slowest one arrived, with none of
// the workgroup's loads in flight meanwhile: 1.15 of the level-0 pass's 13.1 ms.  `arrive` is zeroed by the caller
// before its first __syncthreads().
template <typename R, bool FULL>
__device__ __forceinline__ void part_epilogue(float* red, int* arrive, const float* cs, float* part_out, int d, int lane, int w) {
    constexpr int EPL = R::EPL;
#pragma unroll
    for (int e = 0; e < EPL; e++) red[(w * EPL + e) * SVX_WAVE + lane] = cs[e];
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);  // behind this wave's writes
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != 3) return;
    float sum[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        float t = red[(0 * EPL + e) * SVX_WAVE + lane];
        t += red[(1 * EPL + e) * SVX_WAVE + lane];
        t += red[(2 * EPL + e) * SVX_WAVE + lane];
        t += red[(3 * EPL + e) * SVX_WAVE + lane];
        sum[e] = t;
    }
    R::template storef<FULL>(part_out, d, lane, sum);
}

// A logical row as it comes out of memory: raw 16-byte pieces (converted when the row is used, so that several
// rows can be in flight in few registers).  PAIR: the row is the sum of two scaled level-0 rows (pair_row).
template <typename E, int NCH, bool PAIR>
struct RawRow {
    uint4 p[PAIR ? 2 * NCH : NCH];
    float ia, ib;
};

// NT: non-temporal loads (the row is not read again before the cache has turned over)
template <typename E, int NCH, bool PAIR, bool NT = false, bool FULL = false>
__device__ __forceinline__ void fetch_raw(const typename E::storage* rows, const float* inv0, int r, int d, int lane,
                                          RawRow<E, NCH, PAIR>& out) {
    constexpr int VEC = E::VEC;
    auto ld = [](const void* p) { return NT ? gld16_nt(p) : gld16(p); };
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int col = (c * SVX_WAVE + lane) * VEC;
        const bool in = FULL || col < d;
        if (PAIR) {
            out.p[c] = in ? ld(rows + (size_t)(2 * r) * d + col) : make_uint4(0, 0, 0, 0);
            out.p[NCH + c] = in ? ld(rows + (size_t)(2 * r + 1) * d + col) : make_uint4(0, 0, 0, 0);
        } else {
            out.p[c] = in ? ld(rows + (size_t)r * d + col) : make_uint4(0, 0, 0, 0);
        }
    }
    if (PAIR) {
        out.ia = gld(inv0 + 2 * r);
        out.ib = gld(inv0 + 2 * r + 1);
    }
}

template <typename E, int NCH, bool PAIR>
__device__ __forceinline__ void decode_raw(const RawRow<E, NCH, PAIR>& in, float* x) {
#pragma clang fp contract(off)
    constexpr int VEC = E::VEC;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        float a[VEC];
        decode_piece<E>(in.p[c], a);
        if (PAIR) {
            float b[VEC];
            decode_piece<E>(in.p[NCH + c], b);
#pragma unroll
            for (int i = 0; i < VEC; i++) {
                const float pa = a[i] * in.ia, pb = b[i] * in.ib;  // the two products and the sum of pair_row
                x[c * VEC + i] = pa + pb;
            }
        } else {
#pragma unroll
            for (int i = 0; i < VEC; i++) x[c * VEC + i] = a[i];
        }
    }
}

// PAIR: `rows` are the level-0 rows and this block works on level 1, whose rows are formed on the fly
// (pair_row arithmetic) instead of being read back from memory.
template <typename E, int NCH, bool PAIR = false, bool NT = false, bool FULL = false>
__device__ void pyr_block(const typename E::storage* rows, int n, int d, const float* mean, const float* rbar,
                          float* inv_out, float* nrm_out, float* vn_out, float* next, float* part_out, int blk,
                          float* lds, const float* inv0 = nullptr) {
    // lds: [2][NCH*VEC*64] floats for mean / rbar (zero padded), then [4][EPL][64] for the partials
    using R = Row<E, NCH>;
    constexpr int EPL = R::EPL;
    constexpr int DP = EPL * SVX_WAVE;  // padded row length
    constexpr int DEPTH = PAIR ? 2 : (E::VEC == 8 ? 4 : 1);  // rows in flight per wave beyond the one being processed
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* mu_l = lds;
    float* rb_l = lds + DP;
    float* red = lds + 2 * DP;
    int* arrive = reinterpret_cast<int*>(lds + 6 * DP);
    if (threadIdx.x == 0) *arrive = 0;
    // mean / rbar are kept lane-major: the 16-byte group g of lane l sits at float4 index g * 64 + l, so that a
    // wave's ds_read_b128 touches consecutive addresses (the row's own column map would stride them by VEC floats)
    for (int c = threadIdx.x; c < DP; c += blockDim.x) {
        const int ch = c / (SVX_WAVE * E::VEC), l = (c / E::VEC) % SVX_WAVE, i = c % E::VEC;
        const int e = ch * E::VEC + i;
        const int at = ((e >> 2) * SVX_WAVE + l) * 4 + (e & 3);
        mu_l[at] = (mean && c < d) ? gld(mean + c) : 0.f;
        rb_l[at] = (rbar && c < d) ? gld(rbar + c) : 0.f;
    }
    __syncthreads();
    auto lane_major = [&](const float* tab, float* out) {
#pragma unroll
        for (int g4 = 0; g4 < EPL / 4; g4++) {
            const float4 t = *reinterpret_cast<const float4*>(tab + (g4 * SVX_WAVE + lane) * 4);
            out[4 * g4 + 0] = t.x; out[4 * g4 + 1] = t.y; out[4 * g4 + 2] = t.z; out[4 * g4 + 3] = t.w;
        }
    };
    float cs[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) cs[e] = 0.f;
    const int slot0 = blk * SVX_PYR_SLOTS + w * (SVX_PYR_SLOTS / 4);
    const int r0 = 2 * slot0;
    int r1 = r0 + 2 * (SVX_PYR_SLOTS / 4);   // this wave's rows: [r0, r1)
    r1 = r1 < n ? r1 : n;
    float xs[EPL];   // running pair sum
    // a row's scalars (1/norm, n0 / n1) wait in lane (row - r0) and leave as ONE store of up to 16 floats per wave at the
    // end: a 4-byte store from one lane per row cost the level-0 pass 0.9 of its 13.1 ms
    float keep_inv = 0.f, keep_nrm = 0.f;
    RawRow<E, NCH, PAIR> ring[DEPTH];
    // The wave's rows are walked by STRAIGHT-LINE code (the trip count is a compile-time constant, the loads are
    // unconditional with a clamped row index): in a loop the values loaded for the rows ahead reach the next iteration
    // through register copies at the back edge, and the compiler waits for the loads in front of those copies -- every
    // iteration drained the queue it had just filled (s_waitcnt vmcnt(0) right behind the prefetch).
    constexpr int RPWV = 2 * (SVX_PYR_SLOTS / 4);
    static_assert(RPWV % DEPTH == 0, "whole ring revolutions");
    const int rl = r1 - 1;   // last row of this wave
    // this lane's elements of mean and rbar stay in registers for the wave's 16 rows (re-read from LDS for every row they
    // cost 8 ds_read_b128 per row and their waits: levels >= 2 19.2 -> 18.5 ms, level 1 21.6 -> 21.3 ms per 1024 pairs; three
    // workgroups per CU instead of four is what the row walk needs to stream, profiles/micro/row_stream.hip)
    // (wider rows -- 32 elements per lane -- keep reading them from LDS: the registers are not there)
    constexpr bool HOIST = EPL <= 16;
    float mu[EPL], rbv[EPL];
    if (HOIST) {
        lane_major(mu_l, mu);
        lane_major(rb_l, rbv);
    }
    if (r0 < r1) {           // wave-uniform
#pragma unroll
        for (int s = 0; s < DEPTH; s++) fetch_raw<E, NCH, PAIR, NT, FULL>(rows, inv0, (r0 + s) < rl ? (r0 + s) : rl, d, lane, ring[s]);
#pragma unroll
        for (int i = 0; i < RPWV; i++) {
            const int r = r0 + i, s = i % DEPTH;
            const int jp = r >> 1, half = i & 1;   // (r0 is even)
            float x[EPL];
            decode_raw<E, NCH, PAIR>(ring[s], x);
            if (i + DEPTH < RPWV) fetch_raw<E, NCH, PAIR, NT, FULL>(rows, inv0, (r + DEPTH) < rl ? (r + DEPTH) : rl, d, lane, ring[s]);
            if (r >= r1) continue;  // wave-uniform (a partial wave at the end of a layer)
            if (mean) {
                if (!HOIST) lane_major(mu_l, mu);
#pragma unroll
                for (int e = 0; e < EPL; e++) x[e] = x[e] - mu[e];  // columns >= d: 0 - 0
            }
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; e++) ss += x[e] * x[e];
            ss = wave_sum(ss);
            const float den = sqrtf(ss) + 1e-5f;
            const float rden = 1.0f / den;  // one division per row; the elements are scaled by the reciprocal
#pragma unroll
            for (int e = 0; e < EPL; e++) x[e] = x[e] * rden;
            if (rbar) {
                if (!HOIST) lane_major(rb_l, rbv);
                float dt = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; e++) dt += x[e] * rbv[e];
                dt = wave_sum(dt);
                keep_nrm = lane == i ? 1.0f - dt : keep_nrm;
            }
            keep_inv = lane == i ? rden : keep_inv;
            if (vn_out) {
                if constexpr (E::VEC == 8) store_row_via_lds<R, FULL, false>(vn_out + (size_t)r * d, d, lane, x, red + w * EPL * SVX_WAVE);
                else R::template storef<FULL>(vn_out + (size_t)r * d, d, lane, x);
            }
            if (next || part_out) {  // (the level-0 pass keeps only the column sums: level 1 re-forms its rows)
                if (half == 0) {
#pragma unroll
                    for (int e = 0; e < EPL; e++) xs[e] = x[e];
                } else {
#pragma unroll
                    for (int e = 0; e < EPL; e++) {
                        xs[e] = xs[e] + x[e];
                        cs[e] += xs[e];
                    }
                    if (next) {
                        if constexpr (E::VEC == 8) store_row_via_lds<R, FULL, true>(next + (size_t)jp * d, d, lane, xs, red + w * EPL * SVX_WAVE);
                        else R::template storef_nt<FULL>(next + (size_t)jp * d, d, lane, xs);
                    }
                }
            }
        }
    }
    if (lane < RPWV && r0 + lane < r1) {
        if (inv_out) gst(inv_out + r0 + lane, keep_inv);
        if (nrm_out && rbar) gst(nrm_out + r0 + lane, keep_nrm);
    }
    if (part_out) part_epilogue<R, FULL>(red, arrive, cs, part_out, d, lane, w);
}

// ---------------------------------------------------------------- level 0, lean
// The level-0 pass keeps nothing of a row but three scalars -- 1/(||u|| + 1e-5), 1 - <unit row, rbar> -- and its
// share of the column sums of the unit rows' pair sums; the generic block above spends ~200 vector instructions per
// 2 KB row on it, which (not the memory system) bounds it.  This block does the same arithmetic for the values that
// are kept (sum of squares as one fma chain + butterfly, the two products and one sum of every pair-sum element, no
// contraction) with about half the instructions:
//   * rows are taken two at a time (no copy of the first row of a pair into a second set of registers);
//   * <unit row, rbar> is evaluated as rden * <raw row, rbar>, so that both reductions of a row start together;
//   * the four reductions of a row pair (two sums of squares, two dot products) share ONE butterfly: two
//     v_permlane32_swap + add fold the wave's halves of (a, b), one v_permlane16_swap + add puts the four values
//     into the four 16-lane rows of a single register, four DPP steps finish all of them at once.
template <typename E, int NCH, bool FULL = false>
__device__ void pyr0_block(const typename E::storage* rows, int n, int d, const float* rbar, float* inv_out, float* nrm_out,
                           float* part_out, int blk, float* lds) {
    using R = Row<E, NCH>;
    constexpr int EPL = R::EPL;
    constexpr int DP = EPL * SVX_WAVE;
    constexpr int DEPTH = E::VEC == 8 ? 4 : 2;   // rows in flight per wave (an even number: rows travel in pairs)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* rb_l = lds;            // rbar, lane-major (see pyr_block)
    float* red = lds + 2 * DP;
    int* arrive = reinterpret_cast<int*>(lds + 6 * DP);
    if (threadIdx.x == 0) *arrive = 0;
    float keep_inv = 0.f, keep_nrm = 0.f;   // the rows' scalars, one per lane, stored once per wave (see pyr_block)
    if (rbar) {
        for (int c = threadIdx.x; c < DP; c += blockDim.x) {
            const int ch = c / (SVX_WAVE * E::VEC), l = (c / E::VEC) % SVX_WAVE, i = c % E::VEC;
            const int e = ch * E::VEC + i;
            rb_l[((e >> 2) * SVX_WAVE + l) * 4 + (e & 3)] = c < d ? gld(rbar + c) : 0.f;
        }
    }
    const int slot0 = blk * SVX_PYR_SLOTS + w * (SVX_PYR_SLOTS / 4);
    const int r0 = 2 * slot0;
    int r1 = r0 + 2 * (SVX_PYR_SLOTS / 4);
    r1 = r1 < n ? r1 : n;
    RawRow<E, NCH, false> ring[DEPTH];
    constexpr int RPWV = 2 * (SVX_PYR_SLOTS / 4);   // straight-line code over the wave's rows: see pyr_block
    static_assert(RPWV % DEPTH == 0, "whole ring revolutions");
    const int rl = r1 - 1;
    if (r0 < r1) {
#pragma unroll
        for (int s = 0; s < DEPTH; s++) fetch_raw<E, NCH, false, false, FULL>(rows, nullptr, (r0 + s) < rl ? (r0 + s) : rl, d, lane, ring[s]);
    }
    __syncthreads();   // rbar is staged (the rows' loads are already in flight)
    float cs[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) cs[e] = 0.f;
    const int npairs_rows = 2 * (n / 2);   // rows that have a partner (an odd last row only gets its scalars)
    if (r0 < r1) {
#pragma unroll
        for (int i = 0; i < RPWV; i += 2) {
            const int ra = r0 + i, s = i % DEPTH;
            const bool has_b = ra + 1 < r1;
            float xa[EPL], xb[EPL];
            decode_raw<E, NCH, false>(ring[s], xa);
            decode_raw<E, NCH, false>(ring[s + 1], xb);
            if (i + DEPTH < RPWV) {
                fetch_raw<E, NCH, false, false, FULL>(rows, nullptr, (ra + DEPTH) < rl ? (ra + DEPTH) : rl, d, lane, ring[s]);
                fetch_raw<E, NCH, false, false, FULL>(rows, nullptr, (ra + 1 + DEPTH) < rl ? (ra + 1 + DEPTH) : rl, d, lane, ring[s + 1]);
            }
            if (ra >= r1) continue;  // wave-uniform
            if (!has_b) {
#pragma unroll
                for (int e = 0; e < EPL; e++) xb[e] = 0.f;
            }
            float ssa = 0.f, ssb = 0.f, da = 0.f, db = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; e++) { ssa += xa[e] * xa[e]; ssb += xb[e] * xb[e]; }
            if (rbar) {
#pragma unroll
                for (int g4 = 0; g4 < EPL / 4; g4++) {
                    const float4 t = *reinterpret_cast<const float4*>(rb_l + (g4 * SVX_WAVE + lane) * 4);
                    da += xa[4 * g4] * t.x; da += xa[4 * g4 + 1] * t.y; da += xa[4 * g4 + 2] * t.z; da += xa[4 * g4 + 3] * t.w;
                    db += xb[4 * g4] * t.x; db += xb[4 * g4 + 1] * t.y; db += xb[4 * g4 + 2] * t.z; db += xb[4 * g4 + 3] * t.w;
                }
            }
            // ---- one butterfly for the four sums
            float q;
            {
                const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ssa), __float_as_uint(ssb), false, false);
                const float S = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);   // lanes 0-31: a, lanes 32-63: b
                const auto d32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(da), __float_as_uint(db), false, false);
                const float D = __uint_as_float(d32[0]) + __uint_as_float(d32[1]);
                const auto q16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(S), __float_as_uint(D), false, false);
                q = __uint_as_float(q16[0]) + __uint_as_float(q16[1]);   // 16-lane rows: ss(a), dot(a), ss(b), dot(b)
                q += SVX_DPP_F32(q, 0xB1);
                q += SVX_DPP_F32(q, 0x4E);
                q += SVX_DPP_F32(q, 0x141);
                q += SVX_DPP_F32(q, 0x140);
            }
            const float rd_v = 1.0f / (sqrtf(q) + 1e-5f);   // meaningful in the ss rows
            const float rden_a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(rd_v), 0));
            const float rden_b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(rd_v), 32));
            keep_inv = lane == i ? rden_a : (lane == i + 1 ? rden_b : keep_inv);
            if (nrm_out) {
                const float dta = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(q), 16));
                const float dtb = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(q), 48));
                keep_nrm = lane == i ? 1.0f - rden_a * dta : (lane == i + 1 ? 1.0f - rden_b * dtb : keep_nrm);
            }
            if (part_out && ra + 1 < npairs_rows) {
#pragma clang fp contract(off)
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    const float pa = xa[e] * rden_a, pb = xb[e] * rden_b;  // the products and the sum of pair_row
                    const float sum = pa + pb;
                    cs[e] += sum;
                }
            }
        }
    }
    if (lane < RPWV && r0 + lane < r1) {
        gst(inv_out + r0 + lane, keep_inv);
        if (nrm_out) gst(nrm_out + r0 + lane, keep_nrm);
    }
    if (part_out) part_epilogue<R, FULL>(red, arrive, cs, part_out, d, lane, w);
}

// Mean of K*S sampled rows (optionally mean-subtracted and unit-normalised first).
template <typename E, int NCH>
__device__ void sample_mean_block(const typename E::storage* base, int K, int n, int d, const int* idx, int S,
                                  const float* mean /*[K][d] or null*/, bool normalize, float* rbar, float* lds,
                                  const float* inv0 = nullptr, int n0 = 0) {
    using R = Row<E, NCH>;
    constexpr int EPL = R::EPL;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) acc[e] = 0.f;
    const int total = K * S;
#pragma unroll 1
    for (int i = w; i < total; i += 4) {
        const int k = i / S;
        int r = idx[i];
        r = r < 0 ? 0 : (r >= n ? n - 1 : r);  // indices are validated on the host; never fault
        float x[EPL];
        if (inv0) pair_row<E, NCH>(base + (size_t)k * n0 * d, inv0 + (size_t)k * n0, r, d, lane, x);
        else R::load(base + ((size_t)k * n + r) * d, d, lane, x);
        if (mean) {
            float mu[EPL];
            R::loadf(mean + (size_t)k * d, d, lane, mu);
#pragma unroll
            for (int e = 0; e < EPL; e++) x[e] = x[e] - mu[e];
        }
        if (normalize) {
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; e++) ss += x[e] * x[e];
            ss = wave_sum(ss);
            const float den = sqrtf(ss) + 1e-5f;
#pragma unroll
            for (int e = 0; e < EPL; e++) x[e] = x[e] / den;
        }
#pragma unroll
        for (int e = 0; e < EPL; e++) acc[e] += x[e];
    }
#pragma unroll
    for (int e = 0; e < EPL; e++) lds[(w * EPL + e) * SVX_WAVE + lane] = acc[e];
    __syncthreads();
    for (int i2 = threadIdx.x; i2 < EPL * SVX_WAVE; i2 += blockDim.x) {
        const int e = i2 / SVX_WAVE, l = i2 % SVX_WAVE;
        const int col = R::col_of(e, l);
        if (col < d) {
            float s = lds[(0 * EPL + e) * SVX_WAVE + l];
            s += lds[(1 * EPL + e) * SVX_WAVE + l];
            s += lds[(2 * EPL + e) * SVX_WAVE + l];
            s += lds[(3 * EPL + e) * SVX_WAVE + l];
            rbar[col] = s / (float)total;
        }
    }
}

// ---------------------------------------------------------------- fused-pipeline kernels
// MODE 0: levels >= 2 (fp32 pair sums of the level above), 1: level 0 (the inputs), 2: level 1, rows formed from
// the level-0 inputs and their 1/norm on the fly.
#ifndef SVX_PYR_MINW
#define SVX_PYR_BOUNDS __launch_bounds__(256)
#else
#define SVX_PYR_BOUNDS __launch_bounds__(256, SVX_PYR_MINW)
#endif
// FULL: d == NCH * 64 * E::VEC, every lane of every piece is inside the row (the launcher checks it): the loads and
// stores need no per-lane range test.
template <typename E, int NCH, int MODE, bool FULL>
__global__ SVX_PYR_BOUNDS void k_pyramid(const SvxPairDev* __restrict__ pairs, int level) {
    constexpr bool LV0 = MODE == 1;
    __shared__ float lds[SVX_PYR_LDS_FLOATS(NCH, E::VEC)];
    const SvxPairDev& P = pairs[blockIdx.z];
    if (level > P.L) return;
    int side, k;
    if ((int)blockIdx.y < P.K[0]) {
        side = 0;
        k = blockIdx.y;
    } else {
        side = 1;
        k = blockIdx.y - P.K[0];
        if (k >= P.K[1]) return;
    }
    const SvxLevel& Lv = P.lev[level];
    if ((int)blockIdx.x >= Lv.nblk[side]) return;
    const int n = Lv.n[side], d = P.d;
    using S = typename E::storage;
    const int n0 = P.lev[0].n[side];
    const S* rows = LV0 ? reinterpret_cast<const S*>(P.v[side]) + (size_t)k * n * d
                        : (MODE == 2 ? reinterpret_cast<const S*>(P.v[side]) + (size_t)k * n0 * d
                                     : reinterpret_cast<const S*>(Lv.P[side] + (size_t)k * n * d));
    const float* inv0 = MODE == 2 ? P.lev[0].inv[side] + (size_t)k * n0 : nullptr;
    const float* mean = LV0 ? nullptr : Lv.mean[side] + (size_t)k * d;
    const bool want_nrm = !(LV0 && P.norm_override[side]);
    const float* rbar = want_nrm ? Lv.rbar[1 - side] : nullptr;
    float* inv_out = LV0 ? Lv.inv[side] + (size_t)k * n : nullptr;
    float* nrm_out = want_nrm ? Lv.nrm[side] + (size_t)k * n : nullptr;
    float* vn_out = (!LV0 && k == 0) ? Lv.P[side] : nullptr;  // layer 0 is normalised in place
    const bool has_next = level < P.L;
    float* next = (has_next && !LV0) ? P.lev[level + 1].P[side] + (size_t)k * (n / 2) * d : nullptr;
    float* part = has_next ? P.lev[level + 1].part[side] + ((size_t)k * Lv.nblk[side] + blockIdx.x) * d : nullptr;
#ifndef SVX_PYR_NT
#define SVX_PYR_NT 0   // bit 0: level 0, bit 1: level 1, bit 2: levels >= 2 read their rows with non-temporal loads
#endif
    constexpr bool NT = ((SVX_PYR_NT >> (MODE == 1 ? 0 : (MODE == 2 ? 1 : 2))) & 1) != 0;
#ifndef SVX_PYR0_GENERIC
    if constexpr (LV0) {
        pyr0_block<E, NCH, FULL>(rows, n, d, rbar, inv_out, nrm_out, part, blockIdx.x, lds);
        return;
    }
#endif
    pyr_block<E, NCH, MODE == 2, NT, FULL>(rows, n, d, mean, rbar, inv_out, nrm_out, vn_out, next, part, blockIdx.x, lds, inv0);
}

__global__ __launch_bounds__(256) void k_colmean(const SvxPairDev* __restrict__ pairs, int level) {
    const SvxPairDev& P = pairs[blockIdx.z];
    if (level > P.L || level < 1) return;
    int side, k;
    if ((int)blockIdx.y < P.K[0]) {
        side = 0;
        k = blockIdx.y;
    } else {
        side = 1;
        k = blockIdx.y - P.K[0];
        if (k >= P.K[1]) return;
    }
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.d) return;
    const SvxLevel& Lv = P.lev[level];
    const int np = Lv.npart[side];
    const float* part = Lv.part[side] + (size_t)k * np * P.d;
    // (same left-to-right order of additions as a plain loop; the loads of eight partials are in flight together)
    float s = 0.f;
    int b = 0;
    for (; b + 8 <= np; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = part[(size_t)(b + u) * P.d + c];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; b < np; b++) s += part[(size_t)b * P.d + c];
    Lv.mean[side][(size_t)k * P.d + c] = s / (float)Lv.n[side];
}

template <typename E, int NCH, int MODE>
__global__ __launch_bounds__(256) void k_sample_mean(const SvxPairDev* __restrict__ pairs, int level) {
    constexpr bool LV0 = MODE == 1;
    __shared__ float lds[4 * NCH * E::VEC * SVX_WAVE];
    const SvxPairDev& P = pairs[blockIdx.y];
    if (level > P.L) return;
    const int side = blockIdx.x;
    const SvxLevel& Lv = P.lev[level];
    if (Lv.S[side] <= 0 || Lv.sidx[side] == nullptr) return;
    using S = typename E::storage;
    const S* base = (LV0 || MODE == 2) ? reinterpret_cast<const S*>(P.v[side]) : reinterpret_cast<const S*>(Lv.P[side]);
    sample_mean_block<E, NCH>(base, P.K[side], Lv.n[side], P.d, Lv.sidx[side], Lv.S[side], LV0 ? nullptr : Lv.mean[side],
                              true, Lv.rbar[side], lds, MODE == 2 ? P.lev[0].inv[side] : nullptr, P.lev[0].n[side]);
}

// ---------------------------------------------------------------- per-op (plain pointer) kernels
template <int NCH>
__global__ __launch_bounds__(256) void k_norm1_plain(float* vecs, int64_t rows, int d, const float* mean) {
    // make_norm1 over a flat [rows][d] tensor, in place; treats consecutive rows as pair slots
    __shared__ float lds[SVX_PYR_LDS_FLOATS(NCH, 4)];
    int64_t nslots_total = (rows + 1) / 2;
    int64_t blk = blockIdx.x;
    if (blk * SVX_PYR_SLOTS >= nslots_total) return;
    // pyr_block indexes rows with int: hand it a window of at most 2*SVX_PYR_SLOTS rows
    int64_t row0 = blk * 2 * SVX_PYR_SLOTS;
    int n = (int)((rows - row0) < (int64_t)(2 * SVX_PYR_SLOTS) ? (rows - row0) : (int64_t)(2 * SVX_PYR_SLOTS));
    float* base = vecs + row0 * d;
    pyr_block<ElemF32, NCH>(base, n, d, mean, nullptr, nullptr, nullptr, base, nullptr, nullptr, 0, lds);
}

template <int NCH>
__global__ __launch_bounds__(256) void k_pairsum_plain(const float* vecs, int n, int d, float* half, float* part,
                                                       int nblk) {
    // grid (nblk, k): half[k][j] = vecs[k][2j] + vecs[k][2j+1]; part[k][blk] = column sums
    using R = Row<ElemF32, NCH>;
    constexpr int EPL = R::EPL;
    __shared__ float lds[4 * EPL * SVX_WAVE];
    const int k = blockIdx.y, blk = blockIdx.x;
    const int h = n / 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* src = vecs + (size_t)k * n * d;
    float* dst = half + (size_t)k * h * d;
    float cs[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) cs[e] = 0.f;
    const int slot0 = blk * SVX_PYR_SLOTS + w * (SVX_PYR_SLOTS / 4);
#pragma unroll 1
    for (int i = 0; i < SVX_PYR_SLOTS / 4; i++) {
        const int j = slot0 + i;
        if (j >= h) break;
        float xa[EPL], xb[EPL];
        R::load(src + (size_t)(2 * j) * d, d, lane, xa);
        R::load(src + (size_t)(2 * j + 1) * d, d, lane, xb);
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            xa[e] = xa[e] + xb[e];
            cs[e] += xa[e];
        }
        R::storef(dst + (size_t)j * d, d, lane, xa);
    }
#pragma unroll
    for (int e = 0; e < EPL; e++) lds[(w * EPL + e) * SVX_WAVE + lane] = cs[e];
    __syncthreads();
    float* po = part + ((size_t)k * nblk + blk) * d;
    for (int idx = threadIdx.x; idx < EPL * SVX_WAVE; idx += blockDim.x) {
        const int e = idx / SVX_WAVE, l = idx % SVX_WAVE;
        const int col = R::col_of(e, l);
        if (col < d) {
            float s = lds[(0 * EPL + e) * SVX_WAVE + l];
            s += lds[(1 * EPL + e) * SVX_WAVE + l];
            s += lds[(2 * EPL + e) * SVX_WAVE + l];
            s += lds[(3 * EPL + e) * SVX_WAVE + l];
            po[col] = s;
        }
    }
}

__global__ void k_colmean_plain(const float* part, int nblk, int d, int count, float* mean) {
    // grid (ceil(d/256), k)
    const int k = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const float* p = part + (size_t)k * nblk * d;
    float s = 0.f;
    for (int b = 0; b < nblk; b++) s += p[(size_t)b * d + c];
    mean[(size_t)k * d + c] = s / (float)count;
}

template <int NCH>
__global__ __launch_bounds__(256) void k_subnorm_plain(float* half, int h, int d, const float* mean) {
    // grid (nblk, k): half[k][j] = (half[k][j] - mean[k]) / (||.|| + 1e-5), in place
    __shared__ float lds[SVX_PYR_LDS_FLOATS(NCH, 4)];
    const int k = blockIdx.y;
    float* base = half + (size_t)k * h * d;
    pyr_block<ElemF32, NCH>(base, h, d, mean + (size_t)k * d, nullptr, nullptr, nullptr, base, nullptr, nullptr,
                            blockIdx.x, lds);
}

template <int NCH>
__global__ __launch_bounds__(256) void k_sample_mean_plain(const float* vecs, int K, int n, int d, const int* idx,
                                                           int S, float* rbar) {
    __shared__ float lds[4 * NCH * 4 * SVX_WAVE];
    sample_mean_block<ElemF32, NCH>(vecs, K, n, d, idx, S, nullptr, false, rbar, lds);
}

template <int NCH>
__global__ __launch_bounds__(256) void k_rownorms_plain(const float* vecs, int64_t rows, int d, const float* rbar,
                                                        float* norms) {
    using R = Row<ElemF32, NCH>;
    constexpr int EPL = R::EPL;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float rb[EPL];
    R::loadf(rbar, d, lane, rb);
    for (int64_t r = (int64_t)blockIdx.x * 4 + w; r < rows; r += (int64_t)gridDim.x * 4) {
        float x[EPL];
        R::load(vecs + r * d, d, lane, x);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; e++) s += x[e] * rb[e];
        s = wave_sum(s);
        if (lane == 0) norms[r] = 1.0f - s;
    }
}

// Candidate-tensor assembly (svecalign/utils/embedding_utils.py:135-203): out[r] = table[idx[r]], or an
// all-zero row where idx[r] < 0 (PAD / ignored / missing candidate) or where the source row holds a NaN
// (embedding_utils.py:183-190: "loaded a vector with nan value ... Will reset to zero").  One wave per output
// row: its 16-byte pieces sit in registers while the wave votes on the NaN test, then the row is stored.
// DT: 0 float32, 1 float16, 2 bfloat16 (which bit patterns are NaNs); PPL = pieces per lane.
template <int DT>
__device__ __forceinline__ bool piece_has_nan(const uint4& v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (DT == SVX_F32) {
            bad |= (w[i] & 0x7fffffffu) > 0x7f800000u;
        } else {
            const uint32_t lim = DT == SVX_F16 ? 0x7c00u : 0x7f80u;
            bad |= (w[i] & 0x7fffu) > lim;
            bad |= ((w[i] >> 16) & 0x7fffu) > lim;
        }
    }
    return bad;
}

template <int DT, int PPL>
__global__ __launch_bounds__(256) void k_gather_rows(const uint4* __restrict__ table, const int* __restrict__ idx, long long n_out,
                                                     int row_pieces, long long n_rows, uint4* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_out; r += (long long)gridDim.x * 4) {
        const int src = idx[r];
        const bool have = src >= 0 && src < n_rows;  // wave-uniform
        uint4 v[PPL];
        bool bad = false;
#pragma unroll
        for (int i = 0; i < PPL; i++) {
            const int p = lane + 64 * i;
            v[i] = make_uint4(0, 0, 0, 0);
            if (have && p < row_pieces) {
                v[i] = table[(long long)src * row_pieces + p];
                bad |= piece_has_nan<DT>(v[i]);
            }
        }
        const bool zero = __ballot(bad) != 0ull;
#pragma unroll
        for (int i = 0; i < PPL; i++) {
            const int p = lane + 64 * i;
            if (p < row_pieces) out[r * row_pieces + p] = zero ? make_uint4(0, 0, 0, 0) : v[i];
        }
    }
}

inline int nch_f32(int d) {
    int c = (d + 255) / 256;
    return c <= 1 ? 1 : c <= 2 ? 2 : c <= 4 ? 4 : 8;
}
inline int nch_16(int d) {
    int c = (d + 511) / 512;
    return c <= 1 ? 1 : c <= 2 ? 2 : 4;
}

}  // namespace

#define SVX_SWITCH_NCH_F32(d, M) \
    switch (nch_f32(d)) {        \
        case 1: M(1); break;     \
        case 2: M(2); break;     \
        case 4: M(4); break;     \
        default: M(8); break;    \
    }
#define SVX_SWITCH_NCH_16(d, M) \
    switch (nch_16(d)) {        \
        case 1: M(1); break;    \
        case 2: M(2); break;    \
        default: M(4); break;   \
    }

int svxl_make_norm1(svx_ctx* ctx, float* vecs, int64_t rows, int d) {
    if (rows <= 0) return SVX_OK;
    int64_t nblk = ((rows + 1) / 2 + SVX_PYR_SLOTS - 1) / SVX_PYR_SLOTS;
#define M(N) hipLaunchKernelGGL(k_norm1_plain<N>, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, vecs, rows, d, (const float*)nullptr)
    SVX_SWITCH_NCH_F32(d, M)
#undef M
    SVX_LAUNCH_CHECK(ctx, "k_norm1_plain");
    return SVX_OK;
}

int svxl_pairsum(svx_ctx* ctx, const float* vecs, int k, int n, int d, float* half, float* part, int nblk) {
    if (n / 2 <= 0 || k <= 0) return SVX_OK;
#define M(N) hipLaunchKernelGGL(k_pairsum_plain<N>, dim3(nblk, k), dim3(256), 0, ctx->stream, vecs, n, d, half, part, nblk)
    SVX_SWITCH_NCH_F32(d, M)
#undef M
    SVX_LAUNCH_CHECK(ctx, "k_pairsum_plain");
    return SVX_OK;
}

int svxl_colmean_plain(svx_ctx* ctx, const float* part, int k, int nblk, int d, int count, float* mean) {
    hipLaunchKernelGGL(k_colmean_plain, dim3((d + 255) / 256, k), dim3(256), 0, ctx->stream, part, nblk, d, count, mean);
    SVX_LAUNCH_CHECK(ctx, "k_colmean_plain");
    return SVX_OK;
}

int svxl_sub_mean(svx_ctx* ctx, float* half, int k, int h, int d, const float* mean) {
    if (h <= 0) return SVX_OK;
    int nblk = ((h + 1) / 2 + SVX_PYR_SLOTS - 1) / SVX_PYR_SLOTS;
#define M(N) hipLaunchKernelGGL(k_subnorm_plain<N>, dim3(nblk, k), dim3(256), 0, ctx->stream, half, h, d, mean)
    SVX_SWITCH_NCH_F32(d, M)
#undef M
    SVX_LAUNCH_CHECK(ctx, "k_subnorm_plain");
    return SVX_OK;
}

int svxl_sample_mean_plain(svx_ctx* ctx, const float* vecs, int k, int n, int d, const int* idx, int S, float* rbar) {
#define M(N) hipLaunchKernelGGL(k_sample_mean_plain<N>, dim3(1), dim3(256), 0, ctx->stream, vecs, k, n, d, idx, S, rbar)
    SVX_SWITCH_NCH_F32(d, M)
#undef M
    SVX_LAUNCH_CHECK(ctx, "k_sample_mean_plain");
    return SVX_OK;
}

int svxl_norms_from_rbar(svx_ctx* ctx, const float* vecs, int64_t rows, int d, const float* rbar, float* norms) {
    if (rows <= 0) return SVX_OK;
    int64_t nb = (rows + 3) / 4;
    if (nb > 8192) nb = 8192;
#define M(N) hipLaunchKernelGGL(k_rownorms_plain<N>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, vecs, rows, d, rbar, norms)
    SVX_SWITCH_NCH_F32(d, M)
#undef M
    SVX_LAUNCH_CHECK(ctx, "k_rownorms_plain");
    return SVX_OK;
}

// One pyramid level of the fused pipeline: (level >= 1: column means) -> sampled-row means ->
// the streaming pass (norms, normalisers, normalised layer 0, pair sums for level+1).
// part 0: the small helpers (column means of this level, sampled-row means); part 1: the streaming pass.
int svxl_pyramid_level(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int level, int dtype, int d, int max_nblk,
                       int max_ksum, int part) {
    if (n_pairs <= 0 || max_nblk <= 0) return SVX_OK;
    hipStream_t st = ctx->stream;
    if (part == 0 && level >= 1) {
        hipLaunchKernelGGL(k_colmean, dim3((d + 255) / 256, max_ksum, n_pairs), dim3(256), 0, st, pairs, level);
        SVX_LAUNCH_CHECK(ctx, "k_colmean");
    }
#define LAUNCH(E, N, MODE)                                                                                             \
    do {                                                                                                               \
        if (part == 0) hipLaunchKernelGGL((k_sample_mean<E, N, MODE>), dim3(2, n_pairs), dim3(256), 0, st, pairs, level); \
        else if (d == N * E::VEC * SVX_WAVE) hipLaunchKernelGGL((k_pyramid<E, N, MODE, true>), dim3(max_nblk, max_ksum, n_pairs), dim3(256), 0, st, pairs, level); \
        else hipLaunchKernelGGL((k_pyramid<E, N, MODE, false>), dim3(max_nblk, max_ksum, n_pairs), dim3(256), 0, st, pairs, level); \
    } while (0)
    const int mode = level == 0 ? 1 : (level == 1 ? 2 : 0);
    if (mode == 0) {
#define M(N) LAUNCH(ElemF32, N, 0)
        SVX_SWITCH_NCH_F32(d, M)
#undef M
    } else if (dtype == SVX_F32) {
#define M(N) do { if (mode == 1) LAUNCH(ElemF32, N, 1); else LAUNCH(ElemF32, N, 2); } while (0)
        SVX_SWITCH_NCH_F32(d, M)
#undef M
    } else if (dtype == SVX_F16) {
#define M(N) do { if (mode == 1) LAUNCH(ElemF16, N, 1); else LAUNCH(ElemF16, N, 2); } while (0)
        SVX_SWITCH_NCH_16(d, M)
#undef M
    } else {
#define M(N) do { if (mode == 1) LAUNCH(ElemBF16, N, 1); else LAUNCH(ElemBF16, N, 2); } while (0)
        SVX_SWITCH_NCH_16(d, M)
#undef M
    }
#undef LAUNCH
    SVX_LAUNCH_CHECK(ctx, "k_pyramid");
    return SVX_OK;
}

int svxl_gather_rows(svx_ctx* ctx, const void* table, long long n_rows, int row_bytes, int dtype, const int* idx, long long n_out,
                     void* out) {
    if (n_out <= 0) return SVX_OK;
    const int pieces = row_bytes / 16;
    long long nb = (n_out + 3) / 4;
    if (nb > 65535) nb = 65535;
    const int ppl = (pieces + 63) / 64;
    const uint4* t4 = reinterpret_cast<const uint4*>(table);
    uint4* o4 = reinterpret_cast<uint4*>(out);
#define G(DT, PPL) hipLaunchKernelGGL((k_gather_rows<DT, PPL>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, t4, idx, n_out, pieces, n_rows, o4)
#define GP(DT)                                 \
    do {                                       \
        if (ppl <= 1) G(DT, 1);                \
        else if (ppl <= 2) G(DT, 2);           \
        else if (ppl <= 4) G(DT, 4);           \
        else G(DT, 8);                         \
    } while (0)
    if (dtype == SVX_F32) GP(SVX_F32);
    else if (dtype == SVX_F16) GP(SVX_F16);
    else GP(SVX_BF16);
#undef GP
#undef G
    SVX_LAUNCH_CHECK(ctx, "k_gather_rows");
    return SVX_OK;
}

// svx_dp.hip -- the dynamic programmes and the integer glue around them: coarse dense DP,
// band ("sparse") DP, tracebacks, path up-sampling, and the deletion-penalty estimate.
//
// Reference semantics (paths relative to the reference repository):
//   dense_dp             svecalign/vecalign/dp_core.pyx:79-141
//   sparse_dp            svecalign/vecalign/dp_core.pyx:269-404
//   dense_traceback      svecalign/vecalign/dp_utils.py:146-174
//   sparse_traceback     svecalign/vecalign/dp_utils.py:105-143 (+ process_scores :89-102)
//   upsample / extend / alignment_to_search_path / append_slant
//                        svecalign/vecalign/dp_utils.py:261-275, 228-258, 199-225, 177-196
//   DeletionKnob         svecalign/vecalign/dp_utils.py:43-79
//
// Both DPs are anti-diagonal wavefronts: every node of diagonal a = x + y depends only on earlier
// diagonals, so the lanes of a workgroup own the cells of one diagonal and the last few diagonals
// of the float64 cumulative cost live in an LDS ring.  One workgroup per document pair; a batch
// fills the chip with pairs.  Sums are float64, ties are broken by the first strictly smaller
// candidate in transition order, exactly like the reference, so given the same float32 costs the
// results are bit-identical.
#include <stdlib.h>

#include "svx_common.h"

namespace {

// ------------------------------------------------------------------------------ dense DP
struct DenseDpArgs {
    const float* cost;  // [s0][s1]
    int s0, s1;
    float pen;
    double* csum;  // [s0+1][s1+1] or null
    int* bp;       // [s0+1][s1+1]; or, with diag != 0, [s0+s1+1][s0+1] indexed by (anti-diagonal, row):
                   // neighbouring threads then store neighbouring words (fused pipeline only)
    int diag;
};

__device__ void dense_dp_block(const DenseDpArgs& g, double* ring) {
    const int rmax = g.s0 + 1, cmax = g.s1 + 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const double pen_d = (double)g.pen;
    // The cost of this thread's first cell of diagonal k+1 is fetched while diagonal k is being finished (costs do
    // not depend on the DP state), so that no global load sits between two barriers.
    auto cost_of = [&](int k, int r) -> float {
        const int c = k - r;
        return (r >= 1 && r <= g.s0 && c >= 1 && c <= g.s1) ? g.cost[(size_t)(r - 1) * g.s1 + (c - 1)] : 0.f;
    };
    float cnext = cost_of(0, tid);
    for (int k = 0; k <= g.s0 + g.s1; k++) {
        const int rlo = k - g.s1 > 0 ? k - g.s1 : 0;
        const int rhi = k < g.s0 ? k : g.s0;
        double* cur = ring + (size_t)(k % 3) * rmax;
        const double* p1 = ring + (size_t)((k + 2) % 3) * rmax;  // diagonal k-1
        const double* p2 = ring + (size_t)((k + 1) % 3) * rmax;  // diagonal k-2
        const float cfirst = cnext;
        {
            const int k1 = k + 1;
            const int rlo1 = k1 - g.s1 > 0 ? k1 - g.s1 : 0;
            cnext = cost_of(k1, rlo1 + tid);
        }
        for (int r = rlo + tid; r <= rhi; r += nt) {
            const int c = k - r;
            double v;
            int b;
            if (r == 0) {
                v = (double)((float)c * g.pen);  // int * float -> float (dp_core.pyx:109)
                b = 1;
                if (c == 0) { v = 0.0; b = 4; }
            } else if (c == 0) {
                v = (double)((float)r * g.pen);
                b = 2;
            } else {
                const float cf = (r == rlo + tid) ? cfirst : g.cost[(size_t)(r - 1) * g.s1 + (c - 1)];
                const double cost0 = p2[r - 1] + (double)cf;
                const double cost1 = p1[r] + pen_d;
                const double cost2 = p1[r - 1] + pen_d;
                v = cost0;
                b = 0;
                if (cost1 < v) { v = cost1; b = 1; }
                if (cost2 < v) { v = cost2; b = 2; }
            }
            cur[r] = v;
            g.bp[g.diag ? (size_t)k * rmax + r : (size_t)r * cmax + c] = b;
            if (g.csum) g.csum[(size_t)r * cmax + c] = v;
        }
        __syncthreads();
    }
}

// Walks bp from (s0,s1) to (0,0); rows (x_start,x_len,y_start,y_len) land in document order.
// Single thread.  Returns the count or -SVX_ERR_BP.
__device__ int dense_traceback_thread(const int* bp, int s0, int s1, int* out, int diag = 0) {
    const int cmax = s1 + 1, rmax = s0 + 1;
    int xx = s0, yy = s1, n = 0;
    const int cap = s0 + s1;
    while (!(xx == 0 && yy == 0)) {
        if (n >= cap) return -SVX_ERR_BP;
        const int b = bp[diag ? (size_t)(xx + yy) * rmax + xx : (size_t)xx * cmax + yy];
        int* o = out + 4 * (size_t)(cap - 1 - n);  // fill from the back: document order at the end
        if (b == 0) { o[0] = xx - 1; o[1] = 1; o[2] = yy - 1; o[3] = 1; xx--; yy--; }
        else if (b == 1) { o[0] = xx; o[1] = 0; o[2] = yy - 1; o[3] = 1; yy--; }
        else if (b == 2) { o[0] = xx - 1; o[1] = 1; o[2] = yy; o[3] = 0; xx--; }
        else return -SVX_ERR_BP;
        n++;
    }
    // move [cap-n, cap) to the front (dest index < source index: ascending copy is safe)
    if (n < cap)
        for (int i = 0; i < n; i++)
            for (int c = 0; c < 4; c++) out[4 * (size_t)i + c] = out[4 * (size_t)(cap - n + i) + c];
    return n;
}

__global__ __launch_bounds__(256) void k_dense_dp(DenseDpArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    dense_dp_block(g, reinterpret_cast<double*>(smem));
}

__global__ void k_dense_traceback(const int* bp, int s0, int s1, int* align, int* count) {
    if (threadIdx.x == 0) *count = dense_traceback_thread(bp, s0, s1, align);
}

__global__ __launch_bounds__(256) void k_dense_stage_batch(const SvxPairDev* __restrict__ pairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SvxPairDev& P = pairs[blockIdx.x];
    if (*P.status != 0) return;
    const SvxLevel& Lv = P.lev[P.L];
    DenseDpArgs g;
    g.cost = P.dcost;
    g.s0 = Lv.n[0];
    g.s1 = Lv.n[1];
    g.pen = (float)(*Lv.pen);  // the reference passes the float64 penalty through a C float parameter
    g.csum = nullptr;
    g.bp = P.dbp;
    g.diag = 1;
    dense_dp_block(g, reinterpret_cast<double*>(smem));
    if (threadIdx.x == 0) {
        const int n = dense_traceback_thread(P.dbp, g.s0, g.s1, Lv.align, 1);
        *Lv.n_align = n;
        if (n < 0) *P.status = -n;
    }
}

// ------------------------------------------------------------------------------ band DP
struct SparseDpArgs {
    const float* costs;  // [T][A][B] (atb == 0, the reference layout) or [A][T][B] (atb != 0, fused pipeline)
    const int* boff_in;  // [A]
    int A, B;
    double pen;
    int xs, ys;          // x_in_size, y_in_size
    double* csum;        // [A+2][B]
    int* xp;             // [A+2][B] or null when bpk is given
    int* yp;
    unsigned char* bpk;  // optional packed back-pointers: xp << 4 | yp, 0xFF = unreachable (-42)
    int* boff_out;       // [A+2]
    int atb;
};

__device__ __forceinline__ size_t cost_index(const SparseDpArgs& g, int T, int t, int a, int b) {
    return g.atb ? ((size_t)a * T + t) * g.B + b : ((size_t)t * g.A + a) * g.B + b;
}
__device__ __forceinline__ void store_node(const SparseDpArgs& g, size_t o, double v, int bx, int by) {
    g.csum[o] = v;
    if (g.xp) { g.xp[o] = bx; g.yp[o] = by; }
    if (g.bpk) g.bpk[o] = bx < 0 ? (unsigned char)0xFF : (unsigned char)((bx << 4) | by);
}

// Generic kernel: any band width, cells strided over the workgroup, costs read straight from global memory.
// RING: the last maxstep+1 diagonals of csum live in LDS, else they are re-read from the csum output array
// (the workgroup's own stores: one CU, one L1).  Everything that is the same for all cells of a diagonal --
// b_offset_out[a], the cell shift and the ring slot of each move -- is tabulated for WIDE_CH diagonals at a
// time, and a cell's moves are relaxed without branches (clamped addresses), so that their loads are in flight
// together instead of one L2 round trip after the other.
constexpr int WIDE_CH = 32;
__host__ __device__ inline size_t wide_tab_bytes(int NTt) { return (size_t)WIDE_CH * (2 * NTt + 1) * sizeof(int) + (size_t)NTt * sizeof(int); }

template <bool RING>
__device__ void sparse_dp_block(const SparseDpArgs& g, const SvxTypes& ty, double* ring, int* tab) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int A = g.A, B = g.B, Aout = g.A + 2;
    const int T = ty.n, NTt = ty.n + 2;
    const int RD = ty.maxstep + 1;
    const int x_out = g.xs + 1, y_out = g.ys + 1;
    int* tpk = tab;                       // [NTt] xo | yo << 8 | step << 16
    int* shs = tab + NTt;                 // [WIDE_CH][NTt] cell shift of move t on diagonal a0 + i
    int* sls = shs + WIDE_CH * NTt;       // [WIDE_CH][NTt] ring slot (RING) of diagonal a - step
    int* bos = sls + WIDE_CH * NTt;       // [WIDE_CH] b_offset_out
    for (int t = tid; t < NTt; t += nt) tpk[t] = (int)ty.x[t] | ((int)ty.y[t] << 8) | (((int)ty.x[t] + (int)ty.y[t]) << 16);
    for (int a = tid; a < Aout; a += nt) g.boff_out[a] = a < 2 ? g.boff_in[0] : g.boff_in[a - 2] + 1;
    __syncthreads();
    const double inf = __builtin_inf();
    for (int a0 = 0; a0 < Aout; a0 += WIDE_CH) {
        for (int e = tid; e < WIDE_CH * NTt; e += nt) {
            const int i = e / NTt, t = e - i * NTt;
            const int a = a0 + i, st = tpk[t] >> 16, yo = (tpk[t] >> 8) & 255;
            const int ap = a - st;
            int v = 1 << 24, sl = 0;  // no predecessor diagonal: the shift pushes the cell out of the band
            if (a < Aout && ap >= 0) {
                v = g.boff_out[a] - yo - g.boff_out[ap];
                sl = ap % RD;
            }
            shs[e] = v;
            sls[e] = sl;
        }
        for (int i = tid; i < WIDE_CH; i += nt) bos[i] = a0 + i < Aout ? g.boff_out[a0 + i] : 0;
        __syncthreads();
        const int a_end = a0 + WIDE_CH < Aout ? a0 + WIDE_CH : Aout;
        for (int a = a0; a < a_end; a++) {
            const int i = a - a0;
            const int bo = bos[i];
            const int* sh = shs + i * NTt;
            const int* sl = sls + i * NTt;
            const int ac = a - 2;
            for (int b = tid; b < B; b += nt) {
                const size_t o = (size_t)a * B + b;
                const int yy = b + bo;
                const int xx = a - yy;
                // general node: the cost cell (xx-1, yy-1) exists; its band index is b (b_offset_out[a] = b_offset_in[a-2] + 1)
                const bool general = 1 <= xx && xx < x_out && 1 <= yy && yy < y_out && ac < A;
                double best = inf;
                int bx = -42, by = -42;
#pragma unroll 6
                for (int t = 0; t < NTt; t++) {
                    const int pk = tpk[t];
                    const int xo = pk & 255, yo = (pk >> 8) & 255, st = pk >> 16;
                    const int bpv = b + sh[t];
                    const bool ok = general && xo <= xx && yo <= yy && 0 <= bpv && bpv < B;
                    const size_t ci = (ok && t < T) ? cost_index(g, T, t, ac, b) : 0;
                    const double ac_cost = t < T ? (double)g.costs[ci] : g.pen;
                    const double prev = RING ? ring[ok ? sl[t] * B + bpv : 0] : g.csum[ok ? (size_t)(a - st) * B + bpv : 0];
                    const double tot = prev + ac_cost;
                    if (ok && tot < best) { best = tot; bx = xo; by = yo; }
                }
                if (xx == 0 && 0 <= yy && yy < y_out) {
                    best = g.pen * (double)yy; bx = 0; by = 1;
                } else if (yy == 0 && 0 <= xx && xx < x_out) {
                    best = g.pen * (double)xx; bx = 1; by = 0;
                }
                if (RING) ring[(size_t)(a % RD) * B + b] = best;
                store_node(g, o, best, bx, by);
            }
            __syncthreads();
        }
    }
}

// Fast kernel for narrow bands (B <= 64).  Waves 1-3 turn the next chunk of CH node diagonals into three LDS
// tables -- the cost of every (diagonal, type, cell), the csum-ring slot of its predecessor node (or a slot that
// holds +inf when the move is not allowed) and one word per node with its border case and the lanes of its two
// deletion predecessors -- so that wave 0, which carries the serial chain, does no index arithmetic at all: per
// type move two table reads, one ring read, one add and one compare.
#ifndef SVX_DPF_THREADS
#define SVX_DPF_THREADS 256
#endif
constexpr int DPF_THREADS = SVX_DPF_THREADS;   // wave 0 sweeps, the others stage the next chunk's tables and flush the last one's results

__host__ __device__ inline size_t dpf_align16(size_t v) { return (v + 15) & ~(size_t)15; }
__host__ __device__ inline size_t dpf_chunk_bytes(int T, int B, int CH) {
    const size_t cells = (size_t)CH * (T > 0 ? T : 1) * B;
    return dpf_align16(cells * sizeof(float)) + dpf_align16(cells * sizeof(unsigned short)) + dpf_align16((size_t)CH * B * sizeof(unsigned));
}
__host__ __device__ inline size_t dpf_smem_bytes(int T, int B, int RD, int CH) {
    return dpf_align16(((size_t)RD * B + 1) * sizeof(double))   // csum ring + the +inf slot
           + dpf_align16((size_t)(SVX_MAX_TYPES + 2) * sizeof(int))  // packed transitions
           + 2 * dpf_chunk_bytes(T, B, CH) + 2 * (dpf_align16((size_t)CH * B * sizeof(double)) + dpf_align16((size_t)CH * B * sizeof(int)))
           + dpf_align16((size_t)4 * (CH + RD) * sizeof(int));   // per-wave band-offset tables of the staging code
}

// node word: kind (2 bits: 0 general, 1 border x == 0, 2 border y == 0, 3 outside) | lane of the (0,1) predecessor
// (7 bits, 127 = none) << 2 | lane of the (1,0) predecessor << 9.  (A border node of diagonal a costs pen * a.)
__device__ __forceinline__ int dpf_boff_out(const SparseDpArgs& g, int a) { return a < 2 ? g.boff_in[0] : g.boff_in[a - 2] + 1; }

// Stage chunk c (node diagonals a0 .. a0+CH-1) into buffer c & 1.  Every staging wave first copies the band
// offsets of diagonals a0-RD .. a0+CH-1 into its own small LDS table (one global round trip, no hand-off between
// waves); the chunk's costs are one contiguous run in the pipeline's [A][T][B] layout and move as 16-byte pieces.
// wboff: [4 waves][CH + RD] ints.
__device__ __forceinline__ void dpf_stage(const SparseDpArgs& g, const int* tpk, int T, int RD, int CH, int c, char* bufs, int* wboff,
                                          int tsub, int nsub) {
    const int A = g.A, B = g.B, Aout = A + 2;
    const int a0 = c * CH;
    const int Tn = T > 0 ? T : 1;
    char* base = bufs + (size_t)(c & 1) * dpf_chunk_bytes(T, B, CH);
    float* cost = reinterpret_cast<float*>(base);
    unsigned short* idx = reinterpret_cast<unsigned short*>(base + dpf_align16((size_t)CH * Tn * B * sizeof(float)));
    unsigned* node = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(idx) + dpf_align16((size_t)CH * Tn * B * sizeof(unsigned short)));
    const int dummy = RD * B;
    const int TB = T * B;
    const int lim = Aout - 1;
    // ---- this wave's table of b_offset_out[a0 - RD + k], k < CH + RD
    int* wb = wboff + ((threadIdx.x >> 6) & 3) * (CH + RD);
    for (int k = threadIdx.x & 63; k < CH + RD; k += 64) {
        int a = a0 - RD + k;
        a = a < 0 ? 0 : (a > lim ? lim : a);
        const int raw = g.boff_in[a < 2 ? 0 : a - 2];
        wb[k] = a < 2 ? raw : raw + 1;
    }
    // ---- costs
    const bool block_copy = g.atb && (TB % 4 == 0) && ((reinterpret_cast<size_t>(g.costs) & 15) == 0);
    if (block_copy) {
        // cost row of node diagonal a is a - 2: rows a0-2 .. a0+CH-3, rows outside [0, A) are zero
        const long long first = (long long)(a0 - 2) * TB, limit = (long long)A * TB;
        for (int q = tsub; q < CH * TB / 4; q += nsub) {
            const long long o = first + 4ll * q;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (o >= 0 && o + 4 <= limit) v = *reinterpret_cast<const uint4*>(g.costs + o);
            reinterpret_cast<uint4*>(cost)[q] = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- index tables: one thread per column -- a (type, cell) pair or, past those, a node cell -- walking down
    // the chunk's diagonals; with fewer columns than staging threads the diagonals are split over thread groups
    const int ncols = TB + B;
    const int ngroups = nsub / ncols > 0 ? nsub / ncols : 1;
    for (int cg = tsub; cg < ncols * ngroups; cg += nsub) {
        const int col = cg % ncols, grp = cg / ncols;
        const bool is_node = col >= TB;
        const int t = is_node ? 0 : col / B;
        const int b = is_node ? col - TB : col - t * B;
        const int pk = (!is_node && T > 0) ? tpk[t] : (1 << 16);
        const int xo = pk & 255, yo = (pk >> 8) & 255, st = is_node ? 1 : pk >> 16;
        int sl = (a0 + grp - st) % RD;  // ring slot of diagonal a - st
        sl = sl < 0 ? sl + RD : sl;
        const int step = ngroups % RD;
#pragma unroll 4
        for (int i = grp; i < CH; i += ngroups) {
            const int a = a0 + i;
            const bool in = a <= lim;
            const int ac = a - 2, ap = a - st;
            const int boa = wb[RD + i];        // b_offset_out[a]
            const int bop = wb[RD + i - st];   // b_offset_out[a - st]
            const int yy = b + boa, xx = a - yy;
            const bool general = in && 1 <= xx && xx <= g.xs && 1 <= yy && yy <= g.ys && ac < A;
            if (is_node) {
                const int b01 = b + boa - 1 - bop, b10 = b + boa - bop;
                unsigned w = 3u | (127u << 2) | (127u << 9);
                if (in && xx == 0 && 0 <= yy && yy <= g.ys) w = 1u | (127u << 2) | (127u << 9);
                else if (in && yy == 0 && 0 <= xx && xx <= g.xs) w = 2u | (127u << 2) | (127u << 9);
                else if (general)
                    w = ((unsigned)((0 <= b01 && b01 < B) ? b01 : 127) << 2) | ((unsigned)((0 <= b10 && b10 < B) ? b10 : 127) << 9);
                node[i * B + b] = w;
            } else {
                if (!block_copy) {
                    const int acc = ac < 0 ? 0 : (ac >= A ? A - 1 : ac);
                    const float cv = g.costs[cost_index(g, T, t, acc, b)];
                    cost[i * TB + col] = (in && ac >= 0 && ac < A) ? cv : 0.f;
                }
                const int bpv = b + boa - yo - bop;
                const bool ok = general && ap >= 0 && xo <= xx && yo <= yy && 0 <= bpv && bpv < B;
                idx[i * TB + col] = (unsigned short)(ok ? sl * B + bpv : dummy);
            }
            sl += step;
            sl = sl >= RD ? sl - RD : sl;
        }
    }
}

__device__ __forceinline__ double xchg16_f64(double v, int lane) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = xchg16_u32((unsigned)u, lane), hi = xchg16_u32((unsigned)(u >> 32), lane);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double xchg32_f64(double v, int lane) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = xchg32_u32((unsigned)u, lane), hi = xchg32_u32((unsigned)(u >> 32), lane);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// Results of a chunk wait in LDS (csum + winning move per node) and are written to the node arrays by the staging
// waves while wave 0 is already sweeping the next chunk.
__host__ __device__ inline size_t dpf_out_bytes(int B, int CH) { return dpf_align16((size_t)CH * B * sizeof(double)) + dpf_align16((size_t)CH * B * sizeof(int)); }

__device__ __forceinline__ void dpf_flush(const SparseDpArgs& g, const char* obuf, int B, int CH, int a0, int a_end, int tsub, int nsub) {
    const double* ob = reinterpret_cast<const double*>(obuf);
    const int* ok = reinterpret_cast<const int*>(obuf + dpf_align16((size_t)CH * B * sizeof(double)));
    const int n = (a_end - a0) * B;
    for (int e = tsub; e < n; e += nsub) {
        const int key = ok[e];
        const int bx = key == 0x7fffffff ? -42 : (key & 255), by = key == 0x7fffffff ? -42 : ((key >> 8) & 255);
        store_node(g, (size_t)a0 * B + e, ob[e], bx, by);
    }
}

// G lane groups of LB = 64/G lanes share one diagonal: lane = grp*LB + b.  Group grp relaxes the alignment types
// t = grp, grp+G, ...; the per-group winners are merged by (total, t) so that the reference's "first strictly
// smaller candidate wins" order is preserved exactly.  Alignment types move at least two diagonals back, so their
// part of diagonal a+1 only needs diagonals <= a-1 and is worked out while diagonal a is being finished; what
// stays on the serial chain is the pair of deletions, the only moves that look at diagonal a-1: that diagonal is
// kept in a register (`cur`, every lane group holds a copy) and its two neighbours come through ds_bpermute.
// The sweep is a single wave issuing one instruction every few cycles, so its cost is its instruction count:
// no index arithmetic (staged tables), no branches, no global stores (dpf_flush).
template <int TPLT, int G>
__device__ void sparse_dp_fast(const SparseDpArgs& g, const SvxTypes& ty, int CH, char* smem) {
    constexpr int DPF_TPL = TPLT > 0 ? TPLT : 1;
    constexpr bool unrolled = TPLT > 0;  // the launcher picks TPLT = ceil(T / G) when it is <= 6
    constexpr int LB = 64 / G;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the sweep's loop counters and offsets stay on the SALU
    // (g.A comes out of a vector load of the pair's path length: make it, and every count derived from it, scalar)
    const int A = __builtin_amdgcn_readfirstlane(g.A), B = g.B, Aout = A + 2, T = ty.n, NTt = ty.n + 2, RD = ty.maxstep + 1;
    const int b = lane & (LB - 1), grp = lane / LB;
    const int Tn = T > 0 ? T : 1;
    double* ring = reinterpret_cast<double*>(smem);
    int* tpk = reinterpret_cast<int*>(smem + dpf_align16(((size_t)RD * B + 1) * sizeof(double)));
    char* bufs = reinterpret_cast<char*>(tpk) + dpf_align16((size_t)(SVX_MAX_TYPES + 2) * sizeof(int));
    const size_t chunk_bytes = dpf_chunk_bytes(T, B, CH);
    char* obufs = bufs + 2 * chunk_bytes;
    const size_t out_bytes = dpf_out_bytes(B, CH);
    int* wboff = reinterpret_cast<int*>(obufs + 2 * out_bytes);
    const size_t idx_off = dpf_align16((size_t)CH * Tn * B * sizeof(float));
    const size_t node_off = idx_off + dpf_align16((size_t)CH * Tn * B * sizeof(unsigned short));
    const double inf = __builtin_inf();
    for (int t = tid; t < NTt; t += DPF_THREADS) tpk[t] = (int)ty.x[t] | ((int)ty.y[t] << 8) | (((int)ty.x[t] + (int)ty.y[t]) << 16);
    for (int a = tid; a < Aout; a += DPF_THREADS) g.boff_out[a] = a < 2 ? g.boff_in[0] : g.boff_in[a - 2] + 1;
    for (int e = tid; e <= RD * B; e += DPF_THREADS) ring[e] = inf;
    __syncthreads();
    dpf_stage(g, tpk, T, RD, CH, 0, bufs, wboff, tid, DPF_THREADS);
    __syncthreads();
    const int nchunks = (Aout + CH - 1) / CH;
    const double pen = g.pen;
    const int key01 = 1 << 8, key10 = 1;             // (xo, yo) of the two deletions; they come last in the order
    const int gbase = lane & ~(LB - 1);
    const int bb = b < B ? b : 0;  // idle lanes of a group shadow cell 0 (they never store)
    // this lane's alignment types: offsets into one diagonal's [T][B] tables and merge keys
    int toff[DPF_TPL], tkey[DPF_TPL];
    bool tval[DPF_TPL];
    int tand[DPF_TPL], tor[DPF_TPL];  // slot = (idx[o] & tand) | tor: the table entry, or the +inf slot for a type this lane does not have
#pragma unroll
    for (int j = 0; j < DPF_TPL; j++) {
        const int t = grp + G * j;
        const int tc = t < T ? t : Tn - 1;
        tval[j] = t < T;
        tand[j] = tval[j] ? 0xffff : 0;
        tor[j] = tval[j] ? 0 : RD * B;
        toff[j] = tc * B + bb;
        tkey[j] = (t << 16) | (tpk[tc < NTt ? tc : 0] & 0xffff);  // ordered by t; carries (xo, yo) through the merge
    }
    int slot = 0;       // a % RD, maintained incrementally (wave 0 only)
    double cur = inf;   // csum of this lane's cell on the previous diagonal
    for (int c = 0; c < nchunks; c++) {
        const int a0 = c * CH;
        if (wave >= 1) {
            if (c + 1 < nchunks) dpf_stage(g, tpk, T, RD, CH, c + 1, bufs, wboff, tid - 64, DPF_THREADS - 64);
            if (c > 0) dpf_flush(g, obufs + (size_t)((c - 1) & 1) * out_bytes, B, CH, a0 - CH, a0, tid - 64, DPF_THREADS - 64);
        } else {
            const char* base = bufs + (size_t)(c & 1) * chunk_bytes;
            const float* cost = reinterpret_cast<const float*>(base);
            const unsigned short* idx = reinterpret_cast<const unsigned short*>(base + idx_off);
            const unsigned* node = reinterpret_cast<const unsigned*>(base + node_off);
            double* obest = reinterpret_cast<double*>(obufs + (size_t)(c & 1) * out_bytes);
            int* okey = reinterpret_cast<int*>(reinterpret_cast<char*>(obest) + dpf_align16((size_t)CH * B * sizeof(double)));
            const int a_end = (a0 + CH) < Aout ? (a0 + CH) : Aout;
            const int TB = T * B;
            struct Part { double tot; int key; };
            // group merge of one lane's best type move (shared by both sweeps below)
            auto merge = [&](Part r) {
                // (bitwise | and &: no short-circuit, so the comparisons stay mask arithmetic instead of exec branches)
                if (G == 4) {
                    const double ob = xchg16_f64(r.tot, lane);
                    const int ok2 = (int)xchg16_u32((unsigned)r.key, lane);
                    const bool take = (ob < r.tot) | ((ob == r.tot) & (ok2 < r.key));
                    r.tot = take ? ob : r.tot;
                    r.key = take ? ok2 : r.key;
                }
                if (G >= 2) {
                    const double ob = xchg32_f64(r.tot, lane);
                    const int ok2 = (int)xchg32_u32((unsigned)r.key, lane);
                    const bool take = (ob < r.tot) | ((ob == r.tot) & (ok2 < r.key));
                    r.tot = take ? ob : r.tot;
                    r.key = take ? ok2 : r.key;
                }
                return r;
            };
            // the type moves of node diagonal a0 + i (generic type count): this lane's candidates, merged across groups
            auto partial = [&](int i) {
                Part r{inf, 0x7fffffff};
                for (int t = grp; t < T; t += G) {
                    const int o = i * TB + t * B + bb;
                    const double tot = ring[idx[o]] + (double)cost[o];
                    if (tot < r.tot) { r.tot = tot; r.key = (t << 16) | (tpk[t] & 0xffff); }
                }
                return merge(r);
            };
            // stage 2 of diagonal a: the deletions, from the register copy of diagonal a-1 (p01 / p10), then the
            // node's border / outside cases; stores the node and returns its csum
            auto finish = [&](int a, int i, unsigned nw, const Part& part, double p01, double p10) {
                const int kind = nw & 3, s01 = (nw >> 2) & 127, s10 = (nw >> 9) & 127;
                double best = part.tot;
                int bk = part.key;
                const double t01 = p01 + pen, t10 = p10 + pen;
                if (s01 < B && t01 < best) { best = t01; bk = key01; }
                if (s10 < B && t10 < best) { best = t10; bk = key10; }
                // borders cost pen * a; nodes outside the lattice and unreachable nodes are +inf / "none"
                const double border = pen * (double)a;
                const bool k0 = kind == 0, k3 = kind == 3, none = bk == 0x7fffffff;
                const double general = none ? inf : best, edge = k3 ? inf : border;
                best = k0 ? general : edge;
                const int ekey = kind == 1 ? key01 : (kind == 2 ? key10 : 0x7fffffff);
                bk = k0 ? bk : ekey;
                if (b < B && grp == 0) {
                    ring[slot * B + b] = best;
                    obest[i * B + b] = best;
                    okey[i * B + b] = bk;
                }
                return best;
            };
            if (unrolled) {
                // Software pipeline over the LDS round trips: at the top of the step for diagonal a the wave issues
                // together (1) the two neighbour fetches of diagonal a-1 for stage 2 of a, (2) the ring reads of the
                // type moves of a+1, whose slots and costs were fetched one step earlier, and (3) the table rows of
                // a+2 -- and then waits once.
                const int nrows = a_end - a0;
                int sl1[DPF_TPL];
                float cs1[DPF_TPL];
                Part part{inf, 0x7fffffff};
                {
                    double pv0[DPF_TPL];
                    float cs0[DPF_TPL];
                    const int i1 = nrows > 1 ? 1 : 0;
#pragma unroll
                    for (int j = 0; j < DPF_TPL; j++) {
                        const int o = toff[j];
                        cs0[j] = cost[o];
                        pv0[j] = ring[((int)idx[o] & tand[j]) | tor[j]];
                        sl1[j] = ((int)idx[i1 * TB + o] & tand[j]) | tor[j];
                        cs1[j] = cost[i1 * TB + o];
                    }
#pragma unroll
                    for (int j = 0; j < DPF_TPL; j++) {
                        const double tot = pv0[j] + (double)cs0[j];
                        if (tot < part.tot) { part.tot = tot; part.key = tkey[j]; }
                    }
                    part = merge(part);
                }
                unsigned nw = node[bb];
                unsigned nw1 = node[(nrows > 1 ? 1 : 0) * B + bb];
                for (int a = a0; a < a_end; a++) {
                    const int i = a - a0;
                    const int i2 = __builtin_amdgcn_readfirstlane(i + 2 < nrows ? i + 2 : nrows - 1);  // (past the chunk: re-read its last row, unused)
                    const int s01 = (nw >> 2) & 127, s10 = (nw >> 9) & 127;
                    const double p01 = __shfl(cur, gbase + (s01 < B ? s01 : bb), SVX_WAVE);
                    const double p10 = __shfl(cur, gbase + (s10 < B ? s10 : bb), SVX_WAVE);
                    double pv[DPF_TPL];
                    int sl2[DPF_TPL];
                    float cs2[DPF_TPL];
#pragma unroll
                    for (int j = 0; j < DPF_TPL; j++) {
                        pv[j] = ring[sl1[j]];
                        const int o = i2 * TB + toff[j];
                        sl2[j] = ((int)idx[o] & tand[j]) | tor[j];
                        cs2[j] = cost[o];
                    }
                    const unsigned nw2 = node[i2 * B + bb];
                    cur = finish(a, i, nw, part, p01, p10);
                    Part np{inf, 0x7fffffff};
#pragma unroll
                    for (int j = 0; j < DPF_TPL; j++) {
                        const double tot = pv[j] + (double)cs1[j];
                        if (tot < np.tot) { np.tot = tot; np.key = tkey[j]; }
                    }
                    part = merge(np);
#pragma unroll
                    for (int j = 0; j < DPF_TPL; j++) { sl1[j] = sl2[j]; cs1[j] = cs2[j]; }
                    nw = nw1;
                    nw1 = nw2;
                    slot = __builtin_amdgcn_readfirstlane((slot + 1 == RD) ? 0 : slot + 1);
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                Part part = T > 0 ? partial(0) : Part{inf, 0x7fffffff};
                unsigned nw = node[bb];
                for (int a = a0; a < a_end; a++) {
                    const int i = a - a0;
                    const int inext = a + 1 < a_end ? i + 1 : i;  // (the last diagonal re-reads itself: no branch)
                    const unsigned nnw = node[inext * B + bb];
                    Part npart{inf, 0x7fffffff};
                    if (T > 0) npart = partial(inext);
                    const int s01 = (nw >> 2) & 127, s10 = (nw >> 9) & 127;
                    const double p01 = __shfl(cur, gbase + (s01 < B ? s01 : bb), SVX_WAVE);
                    const double p10 = __shfl(cur, gbase + (s10 < B ? s10 : bb), SVX_WAVE);
                    cur = finish(a, i, nw, part, p01, p10);
                    part = npart;
                    nw = nnw;
                    slot = (slot + 1 == RD) ? 0 : slot + 1;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        __syncthreads();
    }
    {
        const int a0 = (nchunks - 1) * CH;
        dpf_flush(g, obufs + (size_t)((nchunks - 1) & 1) * out_bytes, B, CH, a0, Aout, tid, DPF_THREADS);
    }
}

template <int TPLT, int G>
__global__ __launch_bounds__(DPF_THREADS) void k_sparse_dp_fast(SparseDpArgs g, SvxTypes ty, int CH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sparse_dp_fast<TPLT, G>(g, ty, CH, smem);
}

template <bool RING>
__global__ __launch_bounds__(1024) void k_sparse_dp(SparseDpArgs g, SvxTypes ty) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const size_t ring_bytes = RING ? (((size_t)(ty.maxstep + 1) * g.B * sizeof(double) + 15) & ~(size_t)15) : 0;
    sparse_dp_block<RING>(g, ty, reinterpret_cast<double*>(smem), reinterpret_cast<int*>(smem + ring_bytes));
}

__device__ __forceinline__ bool batch_dp_args(const SvxPairDev& P, int depth, int B, SparseDpArgs* g) {
    if (depth > P.L || (depth == P.L && P.L > 0)) return false;
    if (*P.status != 0) return false;
    const SvxLevel& Lv = P.lev[depth];
    g->A = *Lv.path_len;
    if (g->A <= 0) return false;
    g->costs = Lv.costs;
    g->boff_in = Lv.boff;
    g->B = B;
    g->pen = *Lv.pen;
    g->xs = Lv.n[0];
    g->ys = Lv.n[1];
    g->csum = Lv.csum;
    g->xp = Lv.xp;
    g->yp = Lv.yp;
    g->bpk = Lv.bpk;
    g->boff_out = Lv.boff_out;
    g->atb = 1;
    return true;
}

template <bool RING>
__global__ __launch_bounds__(1024) void k_sparse_dp_batch(const SvxPairDev* __restrict__ pairs, int depth, SvxTypes ty, int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SparseDpArgs g;
    if (!batch_dp_args(pairs[blockIdx.x], depth, B, &g)) return;
    const size_t ring_bytes = RING ? (((size_t)(ty.maxstep + 1) * g.B * sizeof(double) + 15) & ~(size_t)15) : 0;
    sparse_dp_block<RING>(g, ty, reinterpret_cast<double*>(smem), reinterpret_cast<int*>(smem + ring_bytes));
}

template <int TPLT, int G>
__global__ __launch_bounds__(DPF_THREADS) void k_sparse_dp_fast_batch(const SvxPairDev* __restrict__ pairs, int depth,
                                                                       SvxTypes ty, int B, int CH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SparseDpArgs g;
    if (!batch_dp_args(pairs[blockIdx.x], depth, B, &g)) return;
    sparse_dp_fast<TPLT, G>(g, ty, CH, smem);
}

// ------------------------------------------------------------------------------ band traceback
// Thread 0 walks the back-pointers from (xs,ys) to (0,0) out of an LDS window that slides down the diagonals
// (sparse_traceback_block), then the whole workgroup moves the rows to the front in document order and computes
// the scores from csum (process_scores).  cap = xs + ys + 2 rows / doubles.
struct TbArgs {
    const double* csum;
    const int* xp;             // int32 back-pointers, or null when bpk is given
    const int* yp;
    const unsigned char* bpk;  // packed back-pointers or null
    const int* boff;           // b_offset_out [Aout]
    int Aout, B, xs, ys;
    int* align;
    double* scores;
    int* count;
    int* status;
    int chunk;                 // > 0: diagonals per LDS window (two windows of back-pointers + b_offset_out); 0: walk global memory
    int cw;                    // > 0 (packed back-pointers, wide bands): a window holds only `cw` columns of each diagonal, a
                               // corridor centred on the walk's column when the window before it was entered (the column moves by
                               // at most one per diagonal, so the walk stays inside; a step that does not reads global memory)
};

// One LDS window = `chunk` diagonals of back-pointers (bytes from bpk, or 16-bit words packed from xp/yp) and
// of b_offset_out.
__host__ __device__ inline size_t tb_win_bytes(int chunk, int B, bool wide) {
    const size_t bp = (size_t)chunk * B * (wide ? 2 : 1);
    return ((bp + 15) & ~(size_t)15) + (size_t)chunk * sizeof(int);
}
constexpr int TB_COR_CHUNK = 64, TB_COR_W = 320;  // corridor windows: 64 diagonals x 320 columns (drift <= 2 x 64 each way)
__host__ __device__ inline size_t tb_smem_bytes(int chunk, int B, bool wide) { return 2 * tb_win_bytes(chunk, B, wide); }
// diagonals per window: a multiple of 16 (so that a window of bytes starts 16-byte aligned) within ~20 KB
__host__ __device__ inline int tb_chunk(int B, bool wide, int win_kb = 20) {
    const int c = (int)(((size_t)win_kb * 1024) / ((size_t)B * (wide ? 2 : 1) + sizeof(int))) & ~15;
    if (c >= 64) return c;
    const int c20 = (int)((20 * 1024) / ((size_t)B * (wide ? 2 : 1) + sizeof(int))) & ~15;
    return c20 >= 64 ? c20 : 0;  // very wide bands walk global memory instead
}
static int tb_win_kb() {
    static const int v = [] { const char* e = getenv("SVX_TB_WIN_KB"); const int k = e ? atoi(e) : 20; return k >= 2 && k <= 64 ? k : 20; }();
    return v;
}

// window j = diagonals [j*chunk, min((j+1)*chunk, Aout)) -> LDS
// corridor window j: columns [c0, c0 + cw) of diagonals [j*chunk, (j+1)*chunk), row stride cw bytes.  c0 and cw are
// multiples of 16 and the corridor lies inside [0, B) (B a multiple of 16): rows move as 16-byte pieces.
__device__ __forceinline__ void tb_load_corridor(const TbArgs& g, char* win, int j, int c0, int tsub, int nsub) {
    const int lo = j * g.chunk;
    const int hi = (lo + g.chunk) < g.Aout ? (lo + g.chunk) : g.Aout;
    int* lbo = reinterpret_cast<int*>(win + ((((size_t)g.chunk * g.cw) + 15) & ~(size_t)15));
    for (int i = tsub; i < hi - lo; i += nsub) lbo[i] = g.boff[lo + i];
    const int ppr = g.cw / 16;  // pieces per row
    const int n = (hi - lo) * ppr;
    for (int e = tsub; e < n; e += nsub) {
        const int i = e / ppr, p = e - i * ppr;
        reinterpret_cast<uint4*>(win)[e] = *reinterpret_cast<const uint4*>(g.bpk + (size_t)(lo + i) * g.B + c0 + 16 * p);
    }
}

__device__ __forceinline__ void tb_load_window(const TbArgs& g, char* win, int j, int tsub, int nsub) {
    const int B = g.B, lo = j * g.chunk;
    const int hi = (lo + g.chunk) < g.Aout ? (lo + g.chunk) : g.Aout;
    const bool wide = g.bpk == nullptr;
    int* lbo = reinterpret_cast<int*>(win + ((((size_t)g.chunk * B * (wide ? 2 : 1)) + 15) & ~(size_t)15));
    for (int i = tsub; i < hi - lo; i += nsub) lbo[i] = g.boff[lo + i];
    const size_t first = (size_t)lo * B, nb = (size_t)(hi - lo) * B;
    if (!wide) {
        unsigned char* lbp = reinterpret_cast<unsigned char*>(win);
        const unsigned char* src = g.bpk + first;
        if ((reinterpret_cast<size_t>(src) & 15) == 0) {
            const size_t nv = nb / 16;
            for (size_t i = tsub; i < nv; i += nsub) reinterpret_cast<uint4*>(lbp)[i] = reinterpret_cast<const uint4*>(src)[i];
            for (size_t i = nv * 16 + tsub; i < nb; i += nsub) lbp[i] = src[i];
        } else {
            for (size_t i = tsub; i < nb; i += nsub) lbp[i] = src[i];
        }
    } else {
        unsigned short* lbw = reinterpret_cast<unsigned short*>(win);
        for (size_t i = tsub; i < nb; i += nsub) {
            const int px = g.xp[first + i], py = g.yp[first + i];
            lbw[i] = (px < 0 || py < 0 || px > 255 || py > 255) ? (unsigned short)0xFFFF : (unsigned short)((px << 8) | py);
        }
    }
}

// Thread 0 walks the back-pointers from (xs,ys) to (0,0) and writes the alignment rows from the back of the
// buffer.  The walk only ever moves to smaller diagonals, so the back-pointers pass through LDS as a sliding
// pair of windows: while thread 0 walks window j the other waves fetch window j-1.
// One window's share of the back-pointer walk, on the first wave with wave-uniform state (see sparse_traceback_block).
// MODE 0: no LDS window (back-pointers and band offsets from global memory), 1: packed bytes in the window, 2: 16-bit pairs in
// the window, 3: corridor window (packed bytes, cw columns from column c0w; outside it: global memory).
template <int MODE>
__device__ __forceinline__ void tb_walk(const TbArgs& g, const unsigned char* lbp, const unsigned short* lbw, const int* lbo, int lo, int c0w,
                                        int cap, int& xx, int& yy, int& nw, int& err, bool& done) {
    const int Aout = g.Aout, B = g.B;
    // The rows the walk produces wait in the lanes (row number nw in lane nw & 63, one compare and four selects per step) and leave 64 at a
    // time, one 16-byte store per lane, instead of as one masked store with its address arithmetic on every step.
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    int first_pending = __builtin_amdgcn_readfirstlane(nw);
    auto flush = [&](int first, int last) {   // rows [first, last), all inside one block of 64
        const int idx = (first & ~63) + (int)threadIdx.x;
        if (idx >= first && idx < last) gst16(g.align + 4 * (size_t)(cap - 1 - idx), (uint32_t)r0, (uint32_t)r1, (uint32_t)r2, (uint32_t)r3);
    };
    for (;;) {
        xx = __builtin_amdgcn_readfirstlane(xx);   // (said again every step: the compiler keeps the loop-carried copies in
        yy = __builtin_amdgcn_readfirstlane(yy);   //  vector registers otherwise, and the whole step with them)
        nw = __builtin_amdgcn_readfirstlane(nw);
        if (xx == 0 && yy == 0) { done = true; break; }
        const int aa = xx + yy;
        if (aa < lo) break;  // continues in the next window
        if (aa < 0 || aa >= Aout || nw >= cap - 1) { err = SVX_ERR_TRACEBACK; break; }
        int bo_a;
        if (MODE != 0) bo_a = lbo[aa - lo];
        else bo_a = g.boff[aa];
        const int bb = yy - __builtin_amdgcn_readfirstlane(bo_a);
        if (bb < 0 || bb >= B) { err = SVX_ERR_TRACEBACK; break; }
        int px, py;
        if (MODE == 3) {
            const int cc = bb - c0w;
            const int v = __builtin_amdgcn_readfirstlane((int)((cc >= 0 && cc < g.cw) ? lbp[(size_t)(aa - lo) * g.cw + cc] : gld(g.bpk + (size_t)aa * B + bb)));
            px = v == 0xFF ? -42 : (v >> 4);
            py = v == 0xFF ? -42 : (v & 15);
        } else if (MODE == 1) {
            const int v = __builtin_amdgcn_readfirstlane((int)lbp[(size_t)(aa - lo) * B + bb]);
            px = v == 0xFF ? -42 : (v >> 4);
            py = v == 0xFF ? -42 : (v & 15);
        } else if (MODE == 2) {
            const int v = __builtin_amdgcn_readfirstlane((int)lbw[(size_t)(aa - lo) * B + bb]);
            px = v == 0xFFFF ? -42 : (v >> 8);
            py = v == 0xFFFF ? -42 : (v & 255);
        } else {
            const size_t o = (size_t)aa * B + bb;
            if (g.bpk) {
                const int v = __builtin_amdgcn_readfirstlane((int)g.bpk[o]);
                px = v == 0xFF ? -42 : (v >> 4);
                py = v == 0xFF ? -42 : (v & 15);
            } else {
                px = __builtin_amdgcn_readfirstlane(g.xp[o]);
                py = __builtin_amdgcn_readfirstlane(g.yp[o]);
            }
        }
        if (px < 0 || py < 0 || (px == 0 && py == 0) || px > xx || py > yy) { err = SVX_ERR_TRACEBACK; break; }
        const int ln = nw & 63;
        {
            const bool me = (int)threadIdx.x == ln;
            r0 = me ? xx - px : r0;
            r1 = me ? px : r1;
            r2 = me ? yy - py : r2;
            r3 = me ? py : r3;
        }
        xx -= px;
        yy -= py;
        nw++;
        if ((nw & 63) == 0) {   // (global stores, not flat ones: a flat store would count against lgkmcnt and the next step's LDS
            flush(first_pending, nw);   //  reads would wait for its acknowledgement)
            first_pending = nw;
        }
    }
    nw = __builtin_amdgcn_readfirstlane(nw);
    if (first_pending < nw) flush(first_pending, nw);
}

__device__ void sparse_traceback_block(const TbArgs& g, char* smem) {
    __shared__ int sh_n;
    __shared__ int sh_c0[2];   // corridor windows: first column of the window in buffer 0 / 1
    const int cap = g.xs + g.ys + 2;
    const int Aout = g.Aout, B = g.B;
    const bool wide = g.bpk == nullptr;
    const bool cor = g.cw > 0;
    const size_t win_bytes = g.chunk > 0 ? (cor ? tb_win_bytes(g.chunk, g.cw, false) : tb_win_bytes(g.chunk, B, wide)) : 0;
    const int nwin = g.chunk > 0 ? (Aout + g.chunk - 1) / g.chunk : 1;
    int xx = __builtin_amdgcn_readfirstlane(g.xs), yy = __builtin_amdgcn_readfirstlane(g.ys), nw = 0, err = 0;  // walk state (wave-uniform: scalar registers)
    const bool wave0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0;   // (a branch the compiler knows to be wave-uniform)
    bool done = false;
    if (g.chunk > 0) {
        if (cor) {
            const int aa = g.xs + g.ys;
            const int b_end = (aa >= 0 && aa < Aout) ? g.ys - g.boff[aa] : 0;
            if (threadIdx.x == 0) {
                int c = (b_end - g.cw / 2) & ~15;
                c = c < 0 ? 0 : (c > B - g.cw ? B - g.cw : c);
                sh_c0[(nwin - 1) & 1] = c;
            }
            __syncthreads();
            tb_load_corridor(g, smem + (size_t)((nwin - 1) & 1) * win_bytes, nwin - 1, sh_c0[(nwin - 1) & 1], threadIdx.x, blockDim.x);
        } else {
            tb_load_window(g, smem + (size_t)((nwin - 1) & 1) * win_bytes, nwin - 1, threadIdx.x, blockDim.x);
        }
        __syncthreads();
    }
    for (int j = nwin - 1; j >= 0; j--) {
        if (cor && j > 0) {
            // the next window's corridor is centred on the column the walk has now, on entering window j
            if (threadIdx.x == 0) {
                const int aa = xx + yy;
                const int bnow = (aa >= 0 && aa < Aout) ? yy - g.boff[aa] : 0;
                int c = (bnow - g.cw / 2) & ~15;
                c = c < 0 ? 0 : (c > B - g.cw ? B - g.cw : c);
                sh_c0[(j - 1) & 1] = c;
            }
            __syncthreads();
            if (threadIdx.x >= 64) tb_load_corridor(g, smem + (size_t)((j - 1) & 1) * win_bytes, j - 1, sh_c0[(j - 1) & 1], threadIdx.x - 64, blockDim.x - 64);
        } else if (g.chunk > 0 && j > 0 && threadIdx.x >= 64) tb_load_window(g, smem + (size_t)((j - 1) & 1) * win_bytes, j - 1, threadIdx.x - 64, blockDim.x - 64);
        // The walk runs on the whole first wave with every value it reads made wave-uniform (v_readfirstlane): its state
        // (xx, yy, the band column, the counters) then lives in scalar registers and the step's tests are scalar branches
        // instead of exec-mask juggling on a single lane: 0.26 -> 0.21 us per step.  What bounds a step now is the issue rate
        // of ONE wave (an instruction every four cycles at best, ~60 per step), not its two LDS reads: fetching the band
        // offsets 64 at a time (one per lane, picked by v_readlane) to save one of them changed nothing.
        if (wave0 && !done && !err) {
            const int lo = g.chunk > 0 ? j * g.chunk : 0;
            const char* win = smem + (size_t)(j & 1) * win_bytes;
            const unsigned char* lbp = reinterpret_cast<const unsigned char*>(win);
            const unsigned short* lbw = reinterpret_cast<const unsigned short*>(win);
            const int* lbo = reinterpret_cast<const int*>(win + ((((size_t)g.chunk * (cor ? g.cw : B) * ((wide && !cor) ? 2 : 1)) + 15) & ~(size_t)15));
            const int c0w = cor ? __builtin_amdgcn_readfirstlane(sh_c0[j & 1]) : 0;
            // (one copy of the walk per storage mode: the mode tests would otherwise be re-evaluated on every step)
            if (cor) tb_walk<3>(g, lbp, lbw, lbo, lo, c0w, cap, xx, yy, nw, err, done);
            else if (g.chunk > 0 && !wide) tb_walk<1>(g, lbp, lbw, lbo, lo, c0w, cap, xx, yy, nw, err, done);
            else if (g.chunk > 0) tb_walk<2>(g, lbp, lbw, lbo, lo, c0w, cap, xx, yy, nw, err, done);
            else tb_walk<0>(g, lbp, lbw, lbo, lo, c0w, cap, xx, yy, nw, err, done);
        }
        if (g.chunk > 0) __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (!err && !done) err = SVX_ERR_TRACEBACK;

        if (!err) {  // the end node itself must be inside the band (the reference indexes it first)
            const int aa = g.xs + g.ys;
            const int bb = (aa >= 0 && aa < Aout) ? g.ys - g.boff[aa] : -1;
            if (bb < 0 || bb >= B) err = SVX_ERR_TRACEBACK;
        }
        sh_n = err ? -err : nw;
    }
    __syncthreads();
    const int n = sh_n;
    if (n < 0) {
        if (threadIdx.x == 0) {
            *g.count = n;
            if (g.status) *g.status = -n;
        }
        return;
    }
    // alignment j (document order) sits at row cap-n+j; its cost is csum[end node] - csum[start node]
    const int nt = blockDim.x;
    for (int base = 0; base < n; base += nt) {
        const int j = base + threadIdx.x;
        int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        double s = 0.0;
        if (j < n) {
            const int* r = g.align + 4 * (size_t)(cap - n + j);
            r0 = r[0]; r1 = r[1]; r2 = r[2]; r3 = r[3];
            const int xa = r0, ya = r2, xb = r0 + r1, yb = r2 + r3;
            const double c0 = g.csum[(size_t)(xa + ya) * B + (ya - g.boff[xa + ya])];
            const double c1 = g.csum[(size_t)(xb + yb) * B + (yb - g.boff[xb + yb])];
            const double cost = c1 - c0;
            s = cost < 0.0 ? 0.0 : cost;  // np.clip(a_min=0)
            if (r1 == 0 || r3 == 0) s = 0.0;
            else s = s / (double)r1 / (double)r3;
        }
        __syncthreads();
        if (j < n) {
            int* w = g.align + 4 * (size_t)j;
            w[0] = r0; w[1] = r1; w[2] = r2; w[3] = r3;
            g.scores[j] = s;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *g.count = n;
}

__global__ __launch_bounds__(256) void k_sparse_traceback(TbArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sparse_traceback_block(g, smem);
}

__global__ __launch_bounds__(256) void k_sparse_traceback_batch(const SvxPairDev* __restrict__ pairs, int depth, int B,
                                                                int chunk, int cw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SvxPairDev& P = pairs[blockIdx.x];
    if (depth > P.L || (depth == P.L && P.L > 0)) return;
    if (*P.status != 0) return;
    const SvxLevel& Lv = P.lev[depth];
    const int A = *Lv.path_len;
    if (A <= 0) return;
    TbArgs g;
    g.csum = Lv.csum; g.xp = Lv.xp; g.yp = Lv.yp; g.bpk = Lv.bpk; g.boff = Lv.boff_out;
    g.Aout = A + 2; g.B = B; g.xs = Lv.n[0]; g.ys = Lv.n[1];
    g.align = Lv.align; g.scores = Lv.scores; g.count = Lv.n_align; g.status = P.status;
    g.chunk = chunk;
    g.cw = cw;
    sparse_traceback_block(g, smem);
}

// ------------------------------------------------------------------------------ search path
// append_slant (dp_utils.py:177-196); python round() is round-half-even = rint in the default mode
__device__ int slant(int* path, int n, int cap, int xw, int yw, int& lx, int& ly) {
    // (lx, ly) = path[n-1], carried in registers so that the walk never reads back what it wrote
    const int NN = xw + yw;
    const int xs = lx, ys = ly;
    for (int ii = 1; ii <= NN; ii++) {
        const int x = xs + (int)rint((double)((long long)xw * ii) / (double)NN);
        const int y = ys + (int)rint((double)((long long)yw * ii) / (double)NN);
        const int delta = x + y - lx - ly;
        int nx, ny;
        if (delta == 1) { nx = x; ny = y; }
        else if (delta == 2) { nx = x - 1; ny = y; }
        else if (delta == 0) { nx = x + 1; ny = y; }
        else continue;
        if (n >= cap) return -SVX_ERR_PATH;
        *reinterpret_cast<int2*>(path + 2 * (size_t)n) = make_int2(nx, ny);
        lx = nx;
        ly = ny;
        n++;
    }
    return n;
}

// upsample_alignment + extend_alignments + alignment_to_search_path, fused (single thread).
__device__ int search_path_thread(const int* align, int n_align, int upsample, int size0, int size1, int* path, int cap) {
    if (cap < 1) return -SVX_ERR_PATH;
    int n = 1, xdel = 0, ydel = 0;
    const int f = upsample ? 2 : 1;
    int xmax = 0, ymax = 0;
    int lx = 0, ly = 0;
    path[0] = 0;
    path[1] = 0;
    for (int i = 0; i < n_align; i++) {
        const int4 r = *reinterpret_cast<const int4*>(align + 4 * (size_t)i);
        const int p = r.y * f, q = r.w * f;
        if (r.y > 0) { const int m = (r.x + r.y) * f - 1; if (m > xmax) xmax = m; }
        if (r.w > 0) { const int m = (r.z + r.w) * f - 1; if (m > ymax) ymax = m; }
        if (p > 0 && q > 0) {
            n = slant(path, n, cap, xdel, ydel, lx, ly);
            if (n < 0) return n;
            xdel = 0; ydel = 0;
            n = slant(path, n, cap, p, q, lx, ly);
            if (n < 0) return n;
        } else if (p > 0) xdel += p;
        else if (q > 0) ydel += q;
    }
    if (upsample) {
        if (xmax > size0 || ymax > size1) return -SVX_ERR_EXTEND;
        const int ex = size0 - xmax, ey = size1 - ymax;
        if (ex == 0) ydel += ey;
        else if (ey == 0) xdel += ex;
        else {
            n = slant(path, n, cap, xdel, ydel, lx, ly);
            if (n < 0) return n;
            xdel = 0; ydel = 0;
            n = slant(path, n, cap, ex, ey, lx, ly);
            if (n < 0) return n;
        }
    }
    return slant(path, n, cap, xdel, ydel, lx, ly);
}

// Parallel form.  A slant never looks at more of the path than the SUM x+y of the last point, and
// that sum grows by exactly one per appended point, so point number a of the path is a pure function
// of the slant that contains it: thread 0 only lists the slants (start offset, start x), then all
// threads evaluate the points independently.  LDS: dx, dy [R+1] and slant offsets / x [R+3].
__host__ __device__ inline size_t sp_smem_bytes(int rows) { return (size_t)(2 * (rows + 1) + 2 * (rows + 3)) * sizeof(int); }

__device__ void search_path_block(const int* align, int n_align, int upsample, int size0, int size1, int* path, int cap,
                                  int* path_len, int* status, char* smem) {
    __shared__ int sh_xmax, sh_ymax, sh_nseg, sh_A;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int R = n_align, f = upsample ? 2 : 1;
    int* dx = reinterpret_cast<int*>(smem);
    int* dy = dx + (R + 1);
    int* soff = dy + (R + 1);
    int* sxs = soff + (R + 3);
    if (tid == 0) { sh_xmax = 0; sh_ymax = 0; }
    __syncthreads();
    for (int i = tid; i < R; i += nt) {
        const int4 r = *reinterpret_cast<const int4*>(align + 4 * (size_t)i);
        dx[i] = r.y * f;
        dy[i] = r.w * f;
        if (r.y > 0) atomicMax(&sh_xmax, (r.x + r.y) * f - 1);
        if (r.w > 0) atomicMax(&sh_ymax, (r.z + r.w) * f - 1);
    }
    __syncthreads();
    if (tid == 0) {
        int Rr = R, err = 0;
        if (upsample) {
            const int xmax = sh_xmax, ymax = sh_ymax;
            if (xmax > size0 || ymax > size1) err = SVX_ERR_EXTEND;
            dx[Rr] = size0 - xmax;  // extend_alignments: (extra_x, extra_y) as one more row; an empty side
            dy[Rr] = size1 - ymax;  //  turns it into a deletion row, exactly like the reference's branches
            Rr++;
        }
        int nseg = 0, off = 1, x = 0, xdel = 0, ydel = 0;
        for (int i = 0; i < Rr && !err; i++) {
            const int p = dx[i], q = dy[i];
            if (p > 0 && q > 0) {
                if (xdel + ydel > 0) {
                    soff[nseg] = off; sxs[nseg] = x; nseg++;
                    off += xdel + ydel; x += xdel;
                    xdel = 0; ydel = 0;
                }
                soff[nseg] = off; sxs[nseg] = x; nseg++;
                off += p + q; x += p;
            } else {
                xdel += p;
                ydel += q;
            }
        }
        if (xdel + ydel > 0) {
            soff[nseg] = off; sxs[nseg] = x; nseg++;
            off += xdel + ydel; x += xdel;
        }
        soff[nseg] = off;  // sentinel: total number of points
        sxs[nseg] = x;
        if (!err && off > cap) err = SVX_ERR_PATH;
        sh_nseg = nseg;
        sh_A = err ? -err : off;
    }
    __syncthreads();
    const int A = sh_A, nseg = sh_nseg;
    if (A < 0) {
        if (tid == 0) { *path_len = A; if (status) *status = -A; }
        return;
    }
    if (tid == 0) { path[0] = 0; path[1] = 0; }
    for (int a = 1 + tid; a < A; a += nt) {
        int lo = 0, hi = nseg - 1;  // largest k with soff[k] <= a
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (soff[mid] <= a) lo = mid; else hi = mid - 1;
        }
        const int o0 = soff[lo], xs = sxs[lo], ys = o0 - 1 - xs;
        const int NN = soff[lo + 1] - o0, xw = sxs[lo + 1] - xs, yw = NN - xw;
        const int ii = a - o0 + 1;
        const int rx = (int)rint((double)((long long)xw * ii) / (double)NN);
        const int ry = (int)rint((double)((long long)yw * ii) / (double)NN);
        const int delta = rx + ry - ii + 1;  // x + y - (sum of the previous point)
        const int px = xs + rx + (delta == 2 ? -1 : (delta == 0 ? 1 : 0));
        *reinterpret_cast<int2*>(path + 2 * (size_t)a) = make_int2(px, ys + ry);
    }
    if (tid == 0) *path_len = A;
}

__global__ __launch_bounds__(256) void k_search_path(const int* align, const int* n_align, int upsample, int size0, int size1,
                                                     int* path, int cap, int* path_len, int lds_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int na = *n_align;
    if (na < 0) {
        if (threadIdx.x == 0) *path_len = na;
        return;
    }
    if (na <= lds_rows) {
        search_path_block(align, na, upsample, size0, size1, path, cap, path_len, nullptr, smem);
    } else if (threadIdx.x == 0) {
        *path_len = search_path_thread(align, na, upsample, size0, size1, path, cap);
    }
}

__global__ __launch_bounds__(256) void k_search_path_batch(const SvxPairDev* __restrict__ pairs, int depth, int lds_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SvxPairDev& P = pairs[blockIdx.x];
    if (depth > P.L || (depth == P.L && P.L > 0)) return;
    if (*P.status != 0) return;
    const SvxLevel& dst = P.lev[depth];
    const SvxLevel& src = P.lev[P.L == 0 ? 0 : depth + 1];
    const int na = *src.n_align;
    if (na >= 0 && na <= lds_rows) {
        search_path_block(src.align, na, P.L > 0, dst.n[0], dst.n[1], dst.path, dst.path_cap, dst.path_len, P.status, smem);
    } else if (threadIdx.x == 0) {
        int n = na < 0 ? na : search_path_thread(src.align, na, P.L > 0, dst.n[0], dst.n[1], dst.path, dst.path_cap);
        *dst.path_len = n;
        if (n < 0) *P.status = -n;
    }
}

// Band-cost chunks of the fused pipeline: as many path points as possible (<= SVX_BC_TAMAX) whose source and
// target extents both stay within SVX_BC_ROWS - SVX_BC_TB rows, so that a chunk's cells never leave the rows its
// workgroup stages.  The end of the chunk that starts at s is found for every s at once (the path is monotone:
// binary search), then thread 0 hops from chunk start to chunk start.
__global__ __launch_bounds__(256) void k_chunk_path(const SvxPairDev* __restrict__ pairs, int depth, int lds_ints, int LIM, int TAMAX) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* nxt = reinterpret_cast<int*>(smem);
    const SvxPairDev& P = pairs[blockIdx.x];
    if (depth > P.L || (depth == P.L && P.L > 0)) return;
    const SvxLevel& Lv = P.lev[depth];
    const int A = *Lv.path_len;
    if (A <= 0 || *P.status != 0) {
        if (threadIdx.x == 0) *Lv.nchunks = 0;
        return;
    }
    const int2* path = reinterpret_cast<const int2*>(Lv.path);
    if (A <= lds_ints) {
        for (int s0 = threadIdx.x; s0 < A; s0 += 256) {
            const int2 p0 = path[s0];
            int lo = s0 + 1, hi = (s0 + TAMAX) < A ? (s0 + TAMAX) : A;
            while (lo < hi) {  // smallest i in (s0, hi] that is hi or leaves the window
                const int mid = (lo + hi) >> 1;
                const int2 p = path[mid];
                if (p.x - p0.x > LIM || p.y - p0.y > LIM) hi = mid;
                else lo = mid + 1;
            }
            nxt[s0] = lo;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int nc = 0;
            for (int s0 = 0; s0 < A; s0 = nxt[s0]) Lv.cstart[nc++] = s0;
            Lv.cstart[nc] = A;
            *Lv.nchunks = nc;
        }
    } else if (threadIdx.x == 0) {
        int nc = 0, s0 = 0;
        int2 p0 = path[0];
        Lv.cstart[0] = 0;
        for (int i = 1; i < A; i++) {
            const int2 p = path[i];
            if (i - s0 >= TAMAX || p.x - p0.x > LIM || p.y - p0.y > LIM) {
                Lv.cstart[++nc] = i;
                s0 = i;
                p0 = p;
            }
        }
        Lv.cstart[++nc] = A;
        *Lv.nchunks = nc;
    }
}

// ------------------------------------------------------------------------------ deletion penalty
// DeletionKnob (dp_utils.py:50-79) with numpy's arithmetic: float32 bin edges i*(max/1000),
// density histogram, float64 cdf, 27 interior knots k/28, np.interp at `frac`.
__device__ void del_penalty_block(const float* scores, long long n, double frac, double* out) {
#pragma clang fp contract(off)  // numpy rounds every multiply and add separately
    __shared__ float edges[1001];
    __shared__ int cnt[1000];
    __shared__ float red[256];
    const int tid = threadIdx.x;
    float mx = -__builtin_inff();
    for (long long i = tid; i < n; i += blockDim.x) mx = fmaxf(mx, scores[i]);
    red[tid] = mx;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    float res_max = red[0];
    if (!(0.0f < res_max)) res_max = 1e-4f;  // dp_utils.py:55-57 (res_min = 0)
    const float step = res_max / 1000.0f;
    for (int i = tid; i < 1001; i += blockDim.x) edges[i] = (i == 1000) ? res_max : (float)i * step;
    for (int i = tid; i < 1000; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    for (long long i = tid; i < n; i += blockDim.x) {
        const float x = scores[i];
        if (!(x >= 0.0f && x <= res_max)) continue;  // outside the range (or NaN): not counted
        int idx = (int)((x / res_max) * 1000.0f);
        idx = idx < 0 ? 0 : (idx > 999 ? 999 : idx);
        while (idx > 0 && x < edges[idx]) idx--;
        while (idx < 999 && x >= edges[idx + 1]) idx++;
        atomicAdd(&cnt[idx], 1);
    }
    __syncthreads();
    if (tid == 0) {
        long long total = 0;
        for (int i = 0; i < 1000; i++) total += cnt[i];
        const double dx = (double)(edges[1] - edges[0]);
        double xs[29], ys[29];
        xs[0] = 0.0;
        ys[0] = 0.0;
        const double kstep = 1.0 / 28.0;
        int k = 1;
        double run = 0.0;
        for (int i = 0; i < 1000 && k <= 27; i++) {
            const double db = (double)(edges[i + 1] - edges[i]);
            run += (double)cnt[i] / db / (double)total;
            const double cdf = run * dx;
            while (k <= 27 && cdf >= (double)k * kstep) {  // searchsorted(cdf, knot, 'left') == i
                xs[k] = (double)k * kstep;
                ys[k] = 0.0 + (double)i / 1000.0 * (double)res_max;
                k++;
            }
        }
        for (; k <= 27; k++) {  // knot above the whole cdf: searchsorted returns len(cdf)
            xs[k] = (double)k * kstep;
            ys[k] = 0.0 + 1000.0 / 1000.0 * (double)res_max;
        }
        xs[28] = 1.0;
        ys[28] = (double)res_max;
        // np.interp
        double res;
        if (frac < xs[0]) res = ys[0];
        else if (frac > xs[28]) res = ys[28];
        else {
            int j = 0;
            while (j < 28 && xs[j + 1] <= frac) j++;
            if (j == 28 || xs[j] == frac) res = ys[j];
            else {
                // numpy evaluates slope*(x - xp[j]) + fp[j] as a separate multiply and add: no FMA contraction
                const double slope = (ys[j + 1] - ys[j]) / (xs[j + 1] - xs[j]);
                res = slope * (frac - xs[j]) + ys[j];
                if (res != res) {
                    res = slope * (frac - xs[j + 1]) + ys[j + 1];
                    if (res != res && ys[j] == ys[j + 1]) res = ys[j];
                }
            }
        }
        *out = res;
    }
}

__global__ __launch_bounds__(256) void k_del_penalty(const float* scores, long long n, double frac, double* out) {
    del_penalty_block(scores, n, frac, out);
}

__global__ __launch_bounds__(256) void k_del_penalty_batch(const SvxPairDev* __restrict__ pairs, double frac) {
    const SvxPairDev& P = pairs[blockIdx.y];
    const int level = blockIdx.x;
    if (level > P.L) return;
    const SvxLevel& Lv = P.lev[level];
    del_penalty_block(Lv.kscore, Lv.kn, frac, Lv.pen);
}

inline int dp_threads(int B) {
    int t = ((B + 63) / 64) * 64;
    return t < 64 ? 64 : (t > 1024 ? 1024 : t);
}

}  // namespace

int svxl_dense_dp(svx_ctx* ctx, const float* cost, int s0, int s1, float pen, double* csum, int* bp) {
    const size_t smem = 3 * (size_t)(s0 + 1) * sizeof(double);
    if (smem > 150 * 1024) return svx_fail(ctx, SVX_ERR_ARG, "dense_dp: %d rows exceed the LDS ring (max 6399)", s0);
    DenseDpArgs g{cost, s0, s1, pen, csum, bp, 0};
    if (smem > 64 * 1024)
        SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_dense_dp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k_dense_dp, dim3(1), dim3(256), smem, ctx->stream, g);
    SVX_LAUNCH_CHECK(ctx, "k_dense_dp");
    return SVX_OK;
}

int svxl_dense_stage_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int max_s0) {
    if (n_pairs <= 0) return SVX_OK;
    const size_t smem = 3 * (size_t)(max_s0 + 1) * sizeof(double);
    if (smem > 150 * 1024) return svx_fail(ctx, SVX_ERR_ARG, "dense_dp: %d rows exceed the LDS ring (max 6399)", max_s0);
    if (smem > 64 * 1024)
        SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_dense_stage_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k_dense_stage_batch, dim3(n_pairs), dim3(256), smem, ctx->stream, pairs);
    SVX_LAUNCH_CHECK(ctx, "k_dense_stage_batch");
    return SVX_OK;
}

int svxl_dense_traceback(svx_ctx* ctx, const int* bp, int s0, int s1, int* align, int* count) {
    hipLaunchKernelGGL(k_dense_traceback, dim3(1), dim3(64), 0, ctx->stream, bp, s0, s1, align, count);
    SVX_LAUNCH_CHECK(ctx, "k_dense_traceback");
    return SVX_OK;
}

// transitions per lane of the pipelined sweep: ceil((T+2)/G) rounded up to an instantiated size, 0 = generic loop
// lane groups sharing one diagonal: as many as the band width allows, but never more than there are types
static int dpf_groups(int T, int B) {
    int G = B <= 16 ? 4 : (B <= 32 ? 2 : 1);
    while (G > 1 && T < G) G >>= 1;
    return G;
}
static int dpf_tpl(int T, int B) {
    const int G = dpf_groups(T, B);
    int t = (T + G - 1) / G;  // alignment types per lane (the two deletions are handled by every lane)
    if (t < 1) t = 1;
    return t <= 4 ? t : (t <= 6 ? 6 : 0);
}

static int dpf_choose_ch(int T, int B, int maxstep) {
    if (B > 64 || maxstep > 120) return 0;  // transitions are packed into 8-bit fields
    const int opts[8] = {64, 48, 32, 24, 16, 12, 8, 4};
    // The sweep is latency-bound, so what matters is how many pairs a CU can host at once -- and, when the sweep runs
    // beside another half-batch's streaming kernels (svx_set_pipeline), that a workgroup fits into the LDS those leave
    // free: SVX_DP_LDS_KB (default 36: four workgroups per CU) bounds the chunk tables.
    static const int budget_kb = [] { const char* e = getenv("SVX_DP_LDS_KB"); const int v = e ? atoi(e) : 36; return v >= 6 ? v : 36; }();
    for (int i = 0; i < 8; i++)
        if (dpf_smem_bytes(T, B, maxstep + 1, opts[i]) <= (size_t)budget_kb * 1024) return opts[i];
    for (int i = 0; i < 5; i++)   // (>= 16 diagonals per chunk)
        if (dpf_smem_bytes(T, B, maxstep + 1, opts[i]) <= 52 * 1024) return opts[i];
    for (int i = 0; i < 8; i++)
        if (dpf_smem_bytes(T, B, maxstep + 1, opts[i]) <= 150 * 1024) return opts[i];
    return 0;
}

int svxl_sparse_dp(svx_ctx* ctx, const float* costs, const int* boff_in, int A, int B, const SvxTypes& types, double pen,
                   int xs, int ys, double* csum, int* xp, int* yp, int* boff_out) {
    if (A <= 0 || B <= 0) return SVX_OK;
    SparseDpArgs g{costs, boff_in, A, B, pen, xs, ys, csum, xp, yp, nullptr, boff_out, 0};
    const int CH = dpf_choose_ch(types.n, B, types.maxstep);
    if (CH > 0) {
        const size_t smem = dpf_smem_bytes(types.n, B, types.maxstep + 1, CH);
#define DPF_LAUNCH(TPLT, GG)                                                                                          \
    do {                                                                                                              \
        if (smem > 64 * 1024)                                                                                         \
            SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sparse_dp_fast<TPLT, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((k_sparse_dp_fast<TPLT, GG>), dim3(1), dim3(DPF_THREADS), smem, ctx->stream, g, types, CH); \
    } while (0)
#define DPF_LAUNCH_G(TPLT)                                                                                            \
    do {                                                                                                              \
        if (G == 4) DPF_LAUNCH(TPLT, 4); else if (G == 2) DPF_LAUNCH(TPLT, 2); else DPF_LAUNCH(TPLT, 1);               \
    } while (0)
        const int G = dpf_groups(types.n, B);
        switch (dpf_tpl(types.n, B)) {
            case 1: DPF_LAUNCH_G(1); break;
            case 2: DPF_LAUNCH_G(2); break;
            case 3: DPF_LAUNCH_G(3); break;
            case 4: DPF_LAUNCH_G(4); break;
            case 6: DPF_LAUNCH_G(6); break;
            default: DPF_LAUNCH_G(0); break;
        }
#undef DPF_LAUNCH_G
#undef DPF_LAUNCH
        SVX_LAUNCH_CHECK(ctx, "k_sparse_dp_fast");
        return SVX_OK;
    }
    const size_t ring = (((size_t)(types.maxstep + 1) * B * sizeof(double)) + 15) & ~(size_t)15;
    const size_t tabs = wide_tab_bytes(types.n + 2);
    const int nt = dp_threads(B);
    if (ring + tabs <= 150 * 1024) {
        const size_t smem = ring + tabs;
        if (smem > 64 * 1024)
            SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sparse_dp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(k_sparse_dp<true>, dim3(1), dim3(nt), smem, ctx->stream, g, types);
    } else {
        hipLaunchKernelGGL(k_sparse_dp<false>, dim3(1), dim3(nt), tabs, ctx->stream, g, types);
    }
    SVX_LAUNCH_CHECK(ctx, "k_sparse_dp");
    return SVX_OK;
}

int svxl_sparse_dp_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int depth, const SvxTypes& types, int B) {
    if (n_pairs <= 0) return SVX_OK;
    const int CH = dpf_choose_ch(types.n, B, types.maxstep);
    if (CH > 0) {
        const size_t smem = dpf_smem_bytes(types.n, B, types.maxstep + 1, CH);
#define DPF_LAUNCH(TPLT, GG)                                                                                          \
    do {                                                                                                              \
        if (smem > 64 * 1024)                                                                                         \
            SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sparse_dp_fast_batch<TPLT, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((k_sparse_dp_fast_batch<TPLT, GG>), dim3(n_pairs), dim3(DPF_THREADS), smem, ctx->stream, pairs, depth, types, B, CH); \
    } while (0)
#define DPF_LAUNCH_G(TPLT)                                                                                            \
    do {                                                                                                              \
        if (G == 4) DPF_LAUNCH(TPLT, 4); else if (G == 2) DPF_LAUNCH(TPLT, 2); else DPF_LAUNCH(TPLT, 1);               \
    } while (0)
        const int G = dpf_groups(types.n, B);
        switch (dpf_tpl(types.n, B)) {
            case 1: DPF_LAUNCH_G(1); break;
            case 2: DPF_LAUNCH_G(2); break;
            case 3: DPF_LAUNCH_G(3); break;
            case 4: DPF_LAUNCH_G(4); break;
            case 6: DPF_LAUNCH_G(6); break;
            default: DPF_LAUNCH_G(0); break;
        }
#undef DPF_LAUNCH_G
#undef DPF_LAUNCH
        SVX_LAUNCH_CHECK(ctx, "k_sparse_dp_fast_batch");
        return SVX_OK;
    }
    const size_t ring = (((size_t)(types.maxstep + 1) * B * sizeof(double)) + 15) & ~(size_t)15;
    const size_t tabs = wide_tab_bytes(types.n + 2);
    const int nt = dp_threads(B);
    if (ring + tabs <= 150 * 1024) {
        const size_t smem = ring + tabs;
        if (smem > 64 * 1024)
            SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sparse_dp_batch<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(k_sparse_dp_batch<true>, dim3(n_pairs), dim3(nt), smem, ctx->stream, pairs, depth, types, B);
    } else {
        hipLaunchKernelGGL(k_sparse_dp_batch<false>, dim3(n_pairs), dim3(nt), tabs, ctx->stream, pairs, depth, types, B);
    }
    SVX_LAUNCH_CHECK(ctx, "k_sparse_dp_batch");
    return SVX_OK;
}


int svxl_sparse_traceback(svx_ctx* ctx, const double* csum, const int* xp, const int* yp, const int* boff, int a_out, int B,
                          int xs, int ys, int* align, double* scores, int* count) {
    TbArgs g;
    g.csum = csum; g.xp = xp; g.yp = yp; g.bpk = nullptr; g.boff = boff;
    g.Aout = a_out; g.B = B; g.xs = xs; g.ys = ys;
    g.align = align; g.scores = scores; g.count = count; g.status = nullptr;
    g.chunk = tb_chunk(B, true, tb_win_kb());
    g.cw = 0;
    const size_t smem = g.chunk > 0 ? tb_smem_bytes(g.chunk, B, true) : 0;
    if (smem > 64 * 1024)
        SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sparse_traceback, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k_sparse_traceback, dim3(1), dim3(256), smem, ctx->stream, g);
    SVX_LAUNCH_CHECK(ctx, "k_sparse_traceback");
    return SVX_OK;
}

int svxl_sparse_traceback_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int depth, int B, int max_A, int packed) {
    if (n_pairs <= 0) return SVX_OK;
    int chunk = tb_chunk(B, !packed, tb_win_kb()), cw = 0;
    size_t smem = chunk > 0 ? tb_smem_bytes(chunk, B, !packed) : 0;
    if (chunk == 0 && packed && B > TB_COR_W && B % 16 == 0) {  // wide band: corridor windows
        chunk = TB_COR_CHUNK;
        cw = TB_COR_W;
        smem = 2 * tb_win_bytes(chunk, cw, false);
    }
    (void)max_A;
    if (smem > 64 * 1024)
        SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sparse_traceback_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k_sparse_traceback_batch, dim3(n_pairs), dim3(256), smem, ctx->stream, pairs, depth, B, chunk, cw);
    SVX_LAUNCH_CHECK(ctx, "k_sparse_traceback_batch");
    return SVX_OK;
}

static const size_t SP_LDS_LIMIT = 150 * 1024;

static int sp_lds_rows(int max_rows, size_t* smem) {
    int rows = max_rows;
    if (sp_smem_bytes(rows) > SP_LDS_LIMIT) rows = (int)(SP_LDS_LIMIT / (4 * sizeof(int))) - 3;
    *smem = sp_smem_bytes(rows);
    return rows;
}

int svxl_search_path(svx_ctx* ctx, const int* align, const int* n_align, int upsample, int size0, int size1, int* path,
                     int cap, int* path_len) {
    size_t smem;
    const int rows = sp_lds_rows(size0 + size1 + 2, &smem);
    if (smem > 64 * 1024)
        SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_search_path, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k_search_path, dim3(1), dim3(256), smem, ctx->stream, align, n_align, upsample, size0, size1, path, cap,
                       path_len, rows);
    SVX_LAUNCH_CHECK(ctx, "k_search_path");
    return SVX_OK;
}

// max_rows: path capacity of the longest pair at this depth; max_src_rows: alignment rows of the level the path is
// built from (half as many when it is the up-sampled coarser level) -- what the path kernel keeps in LDS.
int svxl_search_path_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int depth, int max_rows, int max_src_rows, int chunk_lim,
                           int chunk_tamax) {
    if (n_pairs <= 0) return SVX_OK;
    size_t smem;
    const int rows = sp_lds_rows(max_src_rows, &smem);
    if (smem > 64 * 1024)
        SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_search_path_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k_search_path_batch, dim3(n_pairs), dim3(256), smem, ctx->stream, pairs, depth, rows);
    SVX_LAUNCH_CHECK(ctx, "k_search_path_batch");
    if (chunk_tamax <= 0) return SVX_OK;  // (the tile sweep of wide straight bands has no use for band-cost chunks)
    int ints = max_rows + 4;  // path points of the longest pair (= its n + m + 4 bound)
    if (ints > 36 * 1024) ints = 36 * 1024;
    const size_t csmem = (size_t)ints * sizeof(int);
    if (csmem > 64 * 1024)
        SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_chunk_path, hipFuncAttributeMaxDynamicSharedMemorySize, (int)csmem));
    hipLaunchKernelGGL(k_chunk_path, dim3(n_pairs), dim3(256), csmem, ctx->stream, pairs, depth, ints, chunk_lim, chunk_tamax);
    SVX_LAUNCH_CHECK(ctx, "k_chunk_path");
    return SVX_OK;
}

int svxl_del_penalty(svx_ctx* ctx, const float* scores, int64_t n, double frac, double* out) {
    hipLaunchKernelGGL(k_del_penalty, dim3(1), dim3(256), 0, ctx->stream, scores, (long long)n, frac, out);
    SVX_LAUNCH_CHECK(ctx, "k_del_penalty");
    return SVX_OK;
}

int svxl_del_penalty_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int max_levels, double frac) {
    if (n_pairs <= 0) return SVX_OK;
    hipLaunchKernelGGL(k_del_penalty_batch, dim3(max_levels, n_pairs), dim3(256), 0, ctx->stream, pairs, frac);
    SVX_LAUNCH_CHECK(ctx, "k_del_penalty_batch");
    return SVX_OK;
}

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
#include "svx_common.h"

namespace {

// ------------------------------------------------------------------------------ dense DP
struct DenseDpArgs {
    const float* cost;  // [s0][s1]
    int s0, s1;
    float pen;
    double* csum;  // [s0+1][s1+1] or null
    int* bp;       // [s0+1][s1+1]
};

__device__ void dense_dp_block(const DenseDpArgs& g, double* ring) {
    const int rmax = g.s0 + 1, cmax = g.s1 + 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const double pen_d = (double)g.pen;
    for (int k = 0; k <= g.s0 + g.s1; k++) {
        const int rlo = k - g.s1 > 0 ? k - g.s1 : 0;
        const int rhi = k < g.s0 ? k : g.s0;
        double* cur = ring + (size_t)(k % 3) * rmax;
        const double* p1 = ring + (size_t)((k + 2) % 3) * rmax;  // diagonal k-1
        const double* p2 = ring + (size_t)((k + 1) % 3) * rmax;  // diagonal k-2
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
                const double cost0 = p2[r - 1] + (double)g.cost[(size_t)(r - 1) * g.s1 + (c - 1)];
                const double cost1 = p1[r] + pen_d;
                const double cost2 = p1[r - 1] + pen_d;
                v = cost0;
                b = 0;
                if (cost1 < v) { v = cost1; b = 1; }
                if (cost2 < v) { v = cost2; b = 2; }
            }
            cur[r] = v;
            g.bp[(size_t)r * cmax + c] = b;
            if (g.csum) g.csum[(size_t)r * cmax + c] = v;
        }
        __syncthreads();
    }
}

// Walks bp from (s0,s1) to (0,0); rows (x_start,x_len,y_start,y_len) land in document order.
// Single thread.  Returns the count or -SVX_ERR_BP.
__device__ int dense_traceback_thread(const int* bp, int s0, int s1, int* out) {
    const int cmax = s1 + 1;
    int xx = s0, yy = s1, n = 0;
    const int cap = s0 + s1;
    while (!(xx == 0 && yy == 0)) {
        if (n >= cap) return -SVX_ERR_BP;
        const int b = bp[(size_t)xx * cmax + yy];
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
    dense_dp_block(g, reinterpret_cast<double*>(smem));
    if (threadIdx.x == 0) {
        const int n = dense_traceback_thread(P.dbp, g.s0, g.s1, Lv.align);
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

// Generic kernel: any band width, cells strided over the workgroup, costs read straight from
// global memory.  RING: the last maxstep+1 diagonals of csum live in LDS, else they are re-read
// from the csum output array.
template <bool RING>
__device__ void sparse_dp_block(const SparseDpArgs& g, const SvxTypes& ty, double* ring) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int A = g.A, B = g.B, Aout = g.A + 2;
    const int T = ty.n, NTt = ty.n + 2;
    const int RD = ty.maxstep + 1;
    const int x_out = g.xs + 1, y_out = g.ys + 1;
    for (int a = tid; a < Aout; a += nt) g.boff_out[a] = a < 2 ? g.boff_in[0] : g.boff_in[a - 2] + 1;
    __syncthreads();
    const double inf = __builtin_inf();
    for (int a = 0; a < Aout; a++) {
        const int bo = g.boff_out[a];
        for (int b = tid; b < B; b += nt) {
            const size_t o = (size_t)a * B + b;
            const int yy = b + bo;
            const int xx = a - yy;
            double best;
            int bx, by;
            if (xx == 0 && 0 <= yy && yy < y_out) {
                best = g.pen * (double)yy; bx = 0; by = 1;
            } else if (yy == 0 && 0 <= xx && xx < x_out) {
                best = g.pen * (double)xx; bx = 1; by = 0;
            } else {
                best = inf; bx = -42; by = -42;
                const int xc = xx - 1, yc = yy - 1;
                if (0 <= xc && xc < g.xs && 0 <= yc && yc < g.ys) {
                    const int ac = xc + yc;
                    if (ac < A) {  // ac >= 0 here
                        const int bc = yc - g.boff_in[ac];
                        if (0 <= bc && bc < B) {
                            for (int t = 0; t < NTt; t++) {
                                const int xo = ty.x[t], yo = ty.y[t];
                                const int xpv = xx - xo, ypv = yy - yo;
                                if (0 <= xpv && xpv < x_out && 0 <= ypv && ypv < y_out) {
                                    const int ap = xpv + ypv;  // 0 <= ap < a
                                    const int bpv = ypv - g.boff_out[ap];
                                    if (0 <= bpv && bpv < B) {
                                        const double ac_cost = (t >= T) ? g.pen : (double)g.costs[cost_index(g, T, t, ac, bc)];
                                        const double prev = RING ? ring[(size_t)(ap % RD) * B + bpv] : g.csum[(size_t)ap * B + bpv];
                                        const double tot = prev + ac_cost;
                                        if (tot < best) { best = tot; bx = xo; by = yo; }
                                    }
                                }
                            }
                        }
                    }
                }
            }
            if (RING) ring[(size_t)(a % RD) * B + b] = best;
            store_node(g, o, best, bx, by);
        }
        __syncthreads();
    }
}

// Fast kernel for narrow bands (B <= 64): wave 0 sweeps the diagonals touching only LDS -- the
// csum ring, a ring of b_offset_out and a double-buffered chunk of CH diagonals of costs -- while
// waves 1-3 stage the next chunk from HBM, so no global load sits on the serial chain.
constexpr int DPF_THREADS = 256;

__host__ __device__ inline size_t dpf_smem_bytes(int T, int B, int RD, int CH) {
    return (size_t)RD * B * sizeof(double)                      // csum ring
           + (size_t)(SVX_MAX_TYPES + 2) * sizeof(int)          // packed transitions
           + 2 * (size_t)CH * sizeof(int)                       // b_offset_out of the chunk
           + 2 * (size_t)CH * (T + 2) * sizeof(int)             // lane shift per (diagonal, transition)
           + 2 * (size_t)CH * (T > 0 ? T : 1) * B * sizeof(float);  // costs
}

// Stage chunk c (node diagonals a0 .. a0+CH-1) into buffer c&1: b_offset_out, the lane shift of every
// transition and the cost rows a-2.  Everything is derived from global memory (b_offset_in is tiny
// and cached), so the staging waves need no hand-off between themselves.
__device__ __forceinline__ void dpf_stage(const SparseDpArgs& g, const int* tpk, int T, int CH, int c, int* bo_buf, int* sh_buf,
                                          float* cbuf, int tsub, int nsub) {
    const int A = g.A, B = g.B, Aout = A + 2, NTt = T + 2;
    const int a0 = c * CH;
    int* bo = bo_buf + (size_t)(c & 1) * CH;
    int* sh = sh_buf + (size_t)(c & 1) * CH * NTt;
    for (int i = tsub; i < CH; i += nsub) {
        const int a = a0 + i;
        bo[i] = (a < Aout) ? (a < 2 ? g.boff_in[0] : g.boff_in[a - 2] + 1) : 0;
    }
    for (int idx = tsub; idx < CH * NTt; idx += nsub) {
        const int i = idx / NTt, t = idx - i * NTt;
        const int a = a0 + i;
        const int pk = tpk[t];
        const int yo = (pk >> 8) & 255, st = pk >> 16;
        int v = 1 << 20;  // no predecessor diagonal: pushes the lane index out of the band
        if (a < Aout && a - st >= 0) {
            const int boa = a < 2 ? g.boff_in[0] : g.boff_in[a - 2] + 1;
            const int ap = a - st;
            const int bop = ap < 2 ? g.boff_in[0] : g.boff_in[ap - 2] + 1;
            v = boa - yo - bop;  // predecessor cell = lane + v on diagonal a - st
        }
        sh[idx] = v;
    }
    const int TB = T * B;
    const int total = CH * TB;
    float* dst = cbuf + (size_t)(c & 1) * total;
    if (g.atb) {
        // [A][T][B]: the chunk is one contiguous run of cost rows a0-2 .. a0-2+CH
        const long long first = (long long)(a0 - 2) * TB;
        const long long limit = (long long)A * TB;
        for (int idx = tsub; idx < total; idx += nsub) {
            const long long o = first + idx;
            dst[idx] = (o >= 0 && o < limit) ? g.costs[o] : 0.f;
        }
    } else {
        for (int idx = tsub; idx < total; idx += nsub) {
            const int i = idx / TB, rem = idx - i * TB;
            const int ac = a0 + i - 2;  // cost row of node diagonal a0 + i
            float v = 0.f;
            if (ac >= 0 && ac < A) {
                const int t = rem / B, b = rem - t * B;
                v = g.costs[cost_index(g, T, t, ac, b)];
            }
            dst[idx] = v;
        }
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

// G lane groups of LB = 64/G lanes share one diagonal: lane = grp*LB + b, group grp relaxes the
// transitions t = grp, grp+G, ...; the per-group winners are merged by (total, t) so that the
// reference's "first strictly smaller candidate wins" order is preserved exactly.
template <int TPLT>
__device__ void sparse_dp_fast(const SparseDpArgs& g, const SvxTypes& ty, int CH, char* smem) {
    constexpr int DPF_TPL = TPLT > 0 ? TPLT : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int A = g.A, B = g.B, Aout = A + 2, T = ty.n, NTt = ty.n + 2, RD = ty.maxstep + 1;
    const int G = B <= 16 ? 4 : (B <= 32 ? 2 : 1);
    const int LB = 64 / G;
    const int b = lane & (LB - 1), grp = lane / LB;
    constexpr bool pipelined = TPLT > 0;  // the launcher picks TPLT = ceil((T+2)/G) when it is <= 6
    double* ring = reinterpret_cast<double*>(smem);
    int* tpk = reinterpret_cast<int*>(ring + (size_t)RD * B);
    int* bo_buf = tpk + (SVX_MAX_TYPES + 2);
    int* sh_buf = bo_buf + 2 * CH;
    float* cbuf = reinterpret_cast<float*>(sh_buf + 2 * (size_t)CH * NTt);
    const int TB = T * B;
    for (int t = tid; t < NTt; t += DPF_THREADS) tpk[t] = (int)ty.x[t] | ((int)ty.y[t] << 8) | (((int)ty.x[t] + (int)ty.y[t]) << 16);
    for (int a = tid; a < Aout; a += DPF_THREADS) g.boff_out[a] = a < 2 ? g.boff_in[0] : g.boff_in[a - 2] + 1;
    __syncthreads();
    dpf_stage(g, tpk, T, CH, 0, bo_buf, sh_buf, cbuf, tid, DPF_THREADS);
    __syncthreads();
    const int nchunks = (Aout + CH - 1) / CH;
    const double inf = __builtin_inf();
    const double pen = g.pen;
    int slot = 0;  // a % RD, maintained incrementally (wave 0 only)
    for (int c = 0; c < nchunks; c++) {
        const int a0 = c * CH;
        if (wave >= 1) {
            if (c + 1 < nchunks) dpf_stage(g, tpk, T, CH, c + 1, bo_buf, sh_buf, cbuf, tid - 64, DPF_THREADS - 64);
        } else {
            const float* cb = cbuf + (size_t)(c & 1) * CH * TB;
            const int* bo = bo_buf + (size_t)(c & 1) * CH;
            const int* sh = sh_buf + (size_t)(c & 1) * CH * NTt;
            const int a_end = (a0 + CH) < Aout ? (a0 + CH) : Aout;
            if (pipelined) {
                // Software-pipelined sweep: everything of diagonal a+1 that does not depend on csum (band
                // offsets, lane shifts, ring addresses, costs) is prepared between issuing the ring reads
                // of diagonal a and consuming them, so the serial chain is ring read -> add/compare ->
                // group merge -> ring write.
                int ridx[DPF_TPL], key[DPF_TPL];
                double cst[DPF_TPL];
                unsigned okm = 0;
                int yy = 0, xx = 0;
                auto prep = [&](int a, int sl, int* r_, int* k_, double* c_, unsigned& ok_, int& yy_, int& xx_) {
                    const int i = a - a0;
                    yy_ = b + bo[i];
                    xx_ = a - yy_;
                    const bool general = (b < B) && 1 <= xx_ && xx_ <= g.xs && 1 <= yy_ && yy_ <= g.ys && a - 2 < A;
                    const float* crow = cb + i * TB + b;
                    const int* shr = sh + i * NTt;
                    ok_ = 0;
#pragma unroll
                    for (int j = 0; j < DPF_TPL; j++) {
                        const int t = grp + G * j;
                        const int tc = t < NTt ? t : NTt - 1;
                        const int pk = tpk[tc];
                        const int xo = pk & 255, yo = (pk >> 8) & 255, st = pk >> 16;
                        const int bpv = b + shr[tc];
                        const bool ok = general && t < NTt && xo <= xx_ && yo <= yy_ && 0 <= bpv && bpv < B;
                        int ps = sl - st;
                        ps = ps < 0 ? ps + RD : ps;
                        r_[j] = ps * B + (ok ? bpv : 0);
                        k_[j] = (t << 16) | (pk & 0xffff);  // ordered by t; carries (xo, yo) through the merge
                        c_[j] = (tc >= T) ? pen : (double)crow[tc * B];
                        ok_ |= ok ? (1u << j) : 0u;
                    }
                };
                prep(a0, slot, ridx, key, cst, okm, yy, xx);
                for (int a = a0; a < a_end; a++) {
                    double pv[DPF_TPL];
#pragma unroll
                    for (int j = 0; j < DPF_TPL; j++) pv[j] = ring[ridx[j]];
                    const int cyy = yy, cxx = xx;
                    const unsigned cok = okm;
                    double ccst[DPF_TPL];
                    int ckey[DPF_TPL];
#pragma unroll
                    for (int j = 0; j < DPF_TPL; j++) { ccst[j] = cst[j]; ckey[j] = key[j]; }
                    const int nslot = (slot + 1 == RD) ? 0 : slot + 1;
                    if (a + 1 < a_end) prep(a + 1, nslot, ridx, key, cst, okm, yy, xx);
                    double best = inf;
                    int bk = 0x7fffffff;
#pragma unroll
                    for (int j = 0; j < DPF_TPL; j++) {
                        const double tot = pv[j] + ccst[j];
                        if (((cok >> j) & 1u) && tot < best) { best = tot; bk = ckey[j]; }
                    }
                    if (G == 4) {
                        const double ob = xchg16_f64(best, lane);
                        const int ok2 = (int)xchg16_u32((unsigned)bk, lane);
                        if (ob < best || (ob == best && ok2 < bk)) { best = ob; bk = ok2; }
                    }
                    if (G >= 2) {
                        const double ob = xchg32_f64(best, lane);
                        const int ok2 = (int)xchg32_u32((unsigned)bk, lane);
                        if (ob < best || (ob == best && ok2 < bk)) { best = ob; bk = ok2; }
                    }
                    if (b < B && grp == 0) {
                        int bx, by;
                        if (cxx == 0 && 0 <= cyy && cyy <= g.ys) {
                            best = pen * (double)cyy; bx = 0; by = 1;
                        } else if (cyy == 0 && 0 <= cxx && cxx <= g.xs) {
                            best = pen * (double)cxx; bx = 1; by = 0;
                        } else if (bk != 0x7fffffff) {
                            bx = bk & 255; by = (bk >> 8) & 255;
                        } else {
                            best = inf; bx = -42; by = -42;
                        }
                        ring[slot * B + b] = best;
                        store_node(g, (size_t)((unsigned)a * (unsigned)B + (unsigned)b), best, bx, by);
                    }
                    slot = nslot;
                    __builtin_amdgcn_wave_barrier();
                }
            } else
            for (int a = a0; a < a_end; a++) {
                const int i = a - a0;
                const bool active = b < B;
                const int yy = b + bo[i];
                const int xx = a - yy;
                double best = inf;
                int bt = 1 << 20;
                // general node: cost cell (a-2, b) exists (b_offset_out[a] = b_offset_in[a-2] + 1, so bc == b)
                const bool general = active && 1 <= xx && xx <= g.xs && 1 <= yy && yy <= g.ys && a - 2 < A;
                if (general) {
                    const float* crow = cb + i * TB + b;
                    const int* shr = sh + i * NTt;
                    for (int t = grp; t < NTt; t += G) {
                        const int pk = tpk[t];
                        const int xo = pk & 255, yo = (pk >> 8) & 255, st = pk >> 16;
                        const int bpv = b + shr[t];
                        const bool ok = xo <= xx && yo <= yy && 0 <= bpv && bpv < B;
                        int ps = slot - st;
                        ps = ps < 0 ? ps + RD : ps;
                        const double prev = ring[ps * B + (ok ? bpv : 0)];
                        const double ac_cost = (t >= T) ? pen : (double)crow[t * B];
                        const double tot = prev + ac_cost;
                        if (ok && tot < best) { best = tot; bt = t; }
                    }
                }
                // merge the groups: smaller total wins, equal totals -> smaller transition index
                if (G == 4) {
                    const double ob = xchg16_f64(best, lane);
                    const int ot = (int)xchg16_u32((unsigned)bt, lane);
                    if (ob < best || (ob == best && ot < bt)) { best = ob; bt = ot; }
                }
                if (G >= 2) {
                    const double ob = xchg32_f64(best, lane);
                    const int ot = (int)xchg32_u32((unsigned)bt, lane);
                    if (ob < best || (ob == best && ot < bt)) { best = ob; bt = ot; }
                }
                if (active && grp == 0) {
                    int bx, by;
                    if (xx == 0 && 0 <= yy && yy <= g.ys) {
                        best = pen * (double)yy; bx = 0; by = 1;
                    } else if (yy == 0 && 0 <= xx && xx <= g.xs) {
                        best = pen * (double)xx; bx = 1; by = 0;
                    } else if (bt < NTt) {
                        const int pk = tpk[bt];
                        bx = pk & 255; by = (pk >> 8) & 255;
                    } else {
                        best = inf; bx = -42; by = -42;
                    }
                    ring[slot * B + b] = best;
                    store_node(g, (size_t)((unsigned)a * (unsigned)B + (unsigned)b), best, bx, by);
                }
                slot = (slot + 1 == RD) ? 0 : slot + 1;
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
    }
}

template <bool RING>
__global__ __launch_bounds__(1024) void k_sparse_dp(SparseDpArgs g, SvxTypes ty) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sparse_dp_block<RING>(g, ty, reinterpret_cast<double*>(smem));
}

template <int TPLT>
__global__ __launch_bounds__(DPF_THREADS) void k_sparse_dp_fast(SparseDpArgs g, SvxTypes ty, int CH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sparse_dp_fast<TPLT>(g, ty, CH, smem);
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
    sparse_dp_block<RING>(g, ty, reinterpret_cast<double*>(smem));
}

template <int TPLT>
__global__ __launch_bounds__(DPF_THREADS) void k_sparse_dp_fast_batch(const SvxPairDev* __restrict__ pairs, int depth,
                                                                       SvxTypes ty, int B, int CH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SparseDpArgs g;
    if (!batch_dp_args(pairs[blockIdx.x], depth, B, &g)) return;
    sparse_dp_fast<TPLT>(g, ty, CH, smem);
}

// ------------------------------------------------------------------------------ band traceback
// Thread 0 walks the back-pointers from (xs,ys) to (0,0) and writes the alignment rows from the
// back of the buffer; when they fit, b_offset_out and the packed back-pointers are first copied
// into LDS so that the pointer chase never leaves the CU.  Then the whole workgroup moves the rows
// to the front in document order and computes the scores from csum (process_scores).
// cap = xs + ys + 2 rows / doubles.
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
    int use_lds;               // LDS holds boff [Aout] ints, then bp [Aout*B]: bytes (from bpk) or 16-bit (from xp/yp)
};

__host__ __device__ inline size_t tb_bp_bytes(int Aout, int B, bool wide) {
    return (((size_t)Aout * B * (wide ? 2 : 1)) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t tb_smem_bytes(int Aout, int B, bool wide) {
    return tb_bp_bytes(Aout, B, wide) + (size_t)Aout * sizeof(int);  // back-pointers first (16-byte aligned), then boff
}

__device__ void sparse_traceback_block(const TbArgs& g, char* smem) {
    __shared__ int sh_n;
    const int cap = g.xs + g.ys + 2;
    const int Aout = g.Aout, B = g.B;
    unsigned char* lbp = reinterpret_cast<unsigned char*>(smem);
    unsigned short* lbw = reinterpret_cast<unsigned short*>(lbp);
    int* lbo = reinterpret_cast<int*>(smem + tb_bp_bytes(Aout, B, g.bpk == nullptr));
    if (g.use_lds) {
        for (int i = threadIdx.x; i < Aout; i += blockDim.x) lbo[i] = g.boff[i];
        const size_t nb = (size_t)Aout * B;
        if (g.bpk) {
            // 16-byte copies (the arena keeps bpk 256-byte aligned; the LDS copy starts at offset 0)
            const size_t nv = nb / 16;
            if ((reinterpret_cast<size_t>(g.bpk) & 15) == 0) {
                const uint4* src = reinterpret_cast<const uint4*>(g.bpk);
                uint4* dst = reinterpret_cast<uint4*>(lbp);
                for (size_t i = threadIdx.x; i < nv; i += blockDim.x) dst[i] = src[i];
                for (size_t i = nv * 16 + threadIdx.x; i < nb; i += blockDim.x) lbp[i] = g.bpk[i];
            } else {
                for (size_t i = threadIdx.x; i < nb; i += blockDim.x) lbp[i] = g.bpk[i];
            }
        } else {
            for (size_t i = threadIdx.x; i < nb; i += blockDim.x) {
                const int px = g.xp[i], py = g.yp[i];
                lbw[i] = (px < 0 || py < 0 || px > 255 || py > 255) ? (unsigned short)0xFFFF : (unsigned short)((px << 8) | py);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int xx = g.xs, yy = g.ys, n = 0, err = 0;
        for (;;) {
            if (xx == 0 && yy == 0) break;
            const int aa = xx + yy;
            if (aa < 0 || aa >= Aout || n >= cap - 1) { err = SVX_ERR_TRACEBACK; break; }
            const int bb = yy - (g.use_lds ? lbo[aa] : g.boff[aa]);
            if (bb < 0 || bb >= B) { err = SVX_ERR_TRACEBACK; break; }
            const size_t o = (size_t)aa * B + bb;
            int px, py;
            if (g.bpk) {
                const unsigned char v = g.use_lds ? lbp[o] : g.bpk[o];
                px = v == 0xFF ? -42 : (v >> 4);
                py = v == 0xFF ? -42 : (v & 15);
            } else if (g.use_lds) {
                const unsigned short v = lbw[o];
                px = v == 0xFFFF ? -42 : (v >> 8);
                py = v == 0xFFFF ? -42 : (v & 255);
            } else {
                px = g.xp[o];
                py = g.yp[o];
            }
            if (px < 0 || py < 0 || (px == 0 && py == 0) || px > xx || py > yy) { err = SVX_ERR_TRACEBACK; break; }
            int* r = g.align + 4 * (size_t)(cap - 1 - n);
            r[0] = xx - px; r[1] = px; r[2] = yy - py; r[3] = py;
            xx -= px;
            yy -= py;
            n++;
        }
        if (!err) {  // the end node itself must be inside the band (the reference indexes it first)
            const int aa = g.xs + g.ys;
            const int bb = (aa >= 0 && aa < Aout) ? g.ys - g.boff[aa] : -1;
            if (bb < 0 || bb >= B) err = SVX_ERR_TRACEBACK;
        }
        sh_n = err ? -err : n;
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
                                                                int lds_cap_aout) {
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
    g.use_lds = (A + 2) <= lds_cap_aout;
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
    DenseDpArgs g{cost, s0, s1, pen, csum, bp};
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
static int dpf_tpl(int T, int B) {
    const int G = B <= 16 ? 4 : (B <= 32 ? 2 : 1);
    const int t = (T + 2 + G - 1) / G;
    return t <= 4 ? t : (t <= 6 ? 6 : 0);
}

static int dpf_choose_ch(int T, int B, int maxstep) {
    if (B > 64 || maxstep > 120) return 0;  // transitions are packed into 8-bit fields
    const int opts[4] = {64, 32, 16, 8};
    for (int i = 0; i < 4; i++)
        if (dpf_smem_bytes(T, B, maxstep + 1, opts[i]) <= 120 * 1024) return opts[i];
    return 0;
}

int svxl_sparse_dp(svx_ctx* ctx, const float* costs, const int* boff_in, int A, int B, const SvxTypes& types, double pen,
                   int xs, int ys, double* csum, int* xp, int* yp, int* boff_out) {
    if (A <= 0 || B <= 0) return SVX_OK;
    SparseDpArgs g{costs, boff_in, A, B, pen, xs, ys, csum, xp, yp, nullptr, boff_out, 0};
    const int CH = dpf_choose_ch(types.n, B, types.maxstep);
    if (CH > 0) {
        const size_t smem = dpf_smem_bytes(types.n, B, types.maxstep + 1, CH);
#define DPF_LAUNCH(TPLT)                                                                                              \
    do {                                                                                                              \
        if (smem > 64 * 1024)                                                                                         \
            SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sparse_dp_fast<TPLT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL(k_sparse_dp_fast<TPLT>, dim3(1), dim3(DPF_THREADS), smem, ctx->stream, g, types, CH);      \
    } while (0)
        switch (dpf_tpl(types.n, B)) {
            case 1: DPF_LAUNCH(1); break;
            case 2: DPF_LAUNCH(2); break;
            case 3: DPF_LAUNCH(3); break;
            case 4: DPF_LAUNCH(4); break;
            case 6: DPF_LAUNCH(6); break;
            default: DPF_LAUNCH(0); break;
        }
#undef DPF_LAUNCH
        SVX_LAUNCH_CHECK(ctx, "k_sparse_dp_fast");
        return SVX_OK;
    }
    const size_t smem = (size_t)(types.maxstep + 1) * B * sizeof(double);
    const int nt = dp_threads(B);
    if (smem <= 150 * 1024) {
        if (smem > 64 * 1024)
            SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sparse_dp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(k_sparse_dp<true>, dim3(1), dim3(nt), smem, ctx->stream, g, types);
    } else {
        hipLaunchKernelGGL(k_sparse_dp<false>, dim3(1), dim3(nt), 0, ctx->stream, g, types);
    }
    SVX_LAUNCH_CHECK(ctx, "k_sparse_dp");
    return SVX_OK;
}

int svxl_sparse_dp_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int depth, const SvxTypes& types, int B) {
    if (n_pairs <= 0) return SVX_OK;
    const int CH = dpf_choose_ch(types.n, B, types.maxstep);
    if (CH > 0) {
        const size_t smem = dpf_smem_bytes(types.n, B, types.maxstep + 1, CH);
#define DPF_LAUNCH(TPLT)                                                                                              \
    do {                                                                                                              \
        if (smem > 64 * 1024)                                                                                         \
            SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sparse_dp_fast_batch<TPLT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL(k_sparse_dp_fast_batch<TPLT>, dim3(n_pairs), dim3(DPF_THREADS), smem, ctx->stream, pairs, depth, types, B, CH); \
    } while (0)
        switch (dpf_tpl(types.n, B)) {
            case 1: DPF_LAUNCH(1); break;
            case 2: DPF_LAUNCH(2); break;
            case 3: DPF_LAUNCH(3); break;
            case 4: DPF_LAUNCH(4); break;
            case 6: DPF_LAUNCH(6); break;
            default: DPF_LAUNCH(0); break;
        }
#undef DPF_LAUNCH
        SVX_LAUNCH_CHECK(ctx, "k_sparse_dp_fast_batch");
        return SVX_OK;
    }
    const size_t smem = (size_t)(types.maxstep + 1) * B * sizeof(double);
    const int nt = dp_threads(B);
    if (smem <= 150 * 1024) {
        if (smem > 64 * 1024)
            SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sparse_dp_batch<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(k_sparse_dp_batch<true>, dim3(n_pairs), dim3(nt), smem, ctx->stream, pairs, depth, types, B);
    } else {
        hipLaunchKernelGGL(k_sparse_dp_batch<false>, dim3(n_pairs), dim3(nt), 0, ctx->stream, pairs, depth, types, B);
    }
    SVX_LAUNCH_CHECK(ctx, "k_sparse_dp_batch");
    return SVX_OK;
}

static const size_t TB_LDS_LIMIT = 150 * 1024;

int svxl_sparse_traceback(svx_ctx* ctx, const double* csum, const int* xp, const int* yp, const int* boff, int a_out, int B,
                          int xs, int ys, int* align, double* scores, int* count) {
    TbArgs g;
    g.csum = csum; g.xp = xp; g.yp = yp; g.bpk = nullptr; g.boff = boff;
    g.Aout = a_out; g.B = B; g.xs = xs; g.ys = ys;
    g.align = align; g.scores = scores; g.count = count; g.status = nullptr;
    size_t smem = tb_smem_bytes(a_out, B, true);
    g.use_lds = smem <= TB_LDS_LIMIT;
    if (!g.use_lds) smem = 0;
    if (smem > 64 * 1024)
        SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sparse_traceback, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k_sparse_traceback, dim3(1), dim3(256), smem, ctx->stream, g);
    SVX_LAUNCH_CHECK(ctx, "k_sparse_traceback");
    return SVX_OK;
}

int svxl_sparse_traceback_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int depth, int B, int max_A, int packed) {
    if (n_pairs <= 0) return SVX_OK;
    // LDS for the largest pair that still fits; larger pairs in the batch chase pointers in global memory
    // (pairs whose types do not pack into bytes hold 16-bit entries: budget for those)
    int cap_aout = (int)(TB_LDS_LIMIT / (sizeof(int) + 2 * (size_t)B));
    if (packed) cap_aout = (int)(TB_LDS_LIMIT / (sizeof(int) + (size_t)B));
    if (cap_aout > max_A + 2) cap_aout = max_A + 2;
    const size_t smem = tb_smem_bytes(cap_aout, B, !packed);
    if (smem > 64 * 1024)
        SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_sparse_traceback_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k_sparse_traceback_batch, dim3(n_pairs), dim3(256), smem, ctx->stream, pairs, depth, B, cap_aout);
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

int svxl_search_path_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int depth, int max_rows) {
    if (n_pairs <= 0) return SVX_OK;
    size_t smem;
    const int rows = sp_lds_rows(max_rows, &smem);
    if (smem > 64 * 1024)
        SVX_HIP(ctx, hipFuncSetAttribute((const void*)k_search_path_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k_search_path_batch, dim3(n_pairs), dim3(256), smem, ctx->stream, pairs, depth, rows);
    SVX_LAUNCH_CHECK(ctx, "k_search_path_batch");
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

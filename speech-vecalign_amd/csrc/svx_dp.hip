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
    const float* costs;  // [T][A][B]
    const int* boff_in;  // [A]
    int A, B;
    double pen;
    int xs, ys;    // x_in_size, y_in_size
    double* csum;  // [A+2][B]
    int* xp;
    int* yp;
    int* boff_out;  // [A+2]
};

template <bool RING>
__device__ void sparse_dp_block(const SparseDpArgs& g, const SvxTypes& ty, double* ring) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int A = g.A, B = g.B, Aout = g.A + 2;
    const int NTt = ty.n + 2;
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
                                        const double ac_cost =
                                            (xo == 0 || yo == 0) ? g.pen : (double)g.costs[((size_t)t * A + ac) * B + bc];
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
            g.csum[o] = best;
            g.xp[o] = bx;
            g.yp[o] = by;
        }
        __syncthreads();
    }
}

template <bool RING>
__global__ __launch_bounds__(1024) void k_sparse_dp(SparseDpArgs g, SvxTypes ty) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sparse_dp_block<RING>(g, ty, reinterpret_cast<double*>(smem));
}

template <bool RING>
__global__ __launch_bounds__(1024) void k_sparse_dp_batch(const SvxPairDev* __restrict__ pairs, int depth, SvxTypes ty, int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SvxPairDev& P = pairs[blockIdx.x];
    if (depth > P.L || (depth == P.L && P.L > 0)) return;
    if (*P.status != 0) return;
    const SvxLevel& Lv = P.lev[depth];
    SparseDpArgs g;
    g.A = *Lv.path_len;
    if (g.A <= 0) return;
    g.costs = Lv.costs;
    g.boff_in = Lv.boff;
    g.B = B;
    g.pen = *Lv.pen;
    g.xs = Lv.n[0];
    g.ys = Lv.n[1];
    g.csum = Lv.csum;
    g.xp = Lv.xp;
    g.yp = Lv.yp;
    g.boff_out = Lv.boff_out;
    sparse_dp_block<RING>(g, ty, reinterpret_cast<double*>(smem));
}

// ------------------------------------------------------------------------------ band traceback
// Thread 0 walks the back-pointers from (xs,ys) to (0,0) writing rows and cumulative costs from
// the back of the buffers; then the whole workgroup moves them to the front in document order
// and turns cumulative costs into scores (process_scores).  cap = xs + ys + 2 rows / doubles.
__device__ void sparse_traceback_block(const double* csum, const int* xp, const int* yp, const int* boff, int Aout, int B,
                                       int xs, int ys, int* align, double* scores, int* count, int* status) {
    __shared__ int sh_n;
    const int cap = xs + ys + 2;
    if (threadIdx.x == 0) {
        int xx = xs, yy = ys, n = 0, err = 0;
        for (;;) {
            const int aa = xx + yy;
            if (aa < 0 || aa >= Aout) { err = SVX_ERR_TRACEBACK; break; }
            const int bb = yy - boff[aa];
            if (bb < 0 || bb >= B) { err = SVX_ERR_TRACEBACK; break; }
            const size_t o = (size_t)aa * B + bb;
            scores[cap - 1 - n] = csum[o];
            if (xx == 0 && yy == 0) break;
            if (n >= cap - 1) { err = SVX_ERR_TRACEBACK; break; }
            const int px = xp[o], py = yp[o];
            if (px < 0 || py < 0 || (px == 0 && py == 0)) { err = SVX_ERR_TRACEBACK; break; }
            int* r = align + 4 * (size_t)(cap - 1 - n);
            r[0] = xx - px; r[1] = px; r[2] = yy - py; r[3] = py;
            xx -= px;
            yy -= py;
            n++;
        }
        sh_n = err ? -err : n;
    }
    __syncthreads();
    const int n = sh_n;
    if (n < 0) {
        if (threadIdx.x == 0) {
            *count = n;
            if (status) *status = -n;
        }
        return;
    }
    // rows: alignment j (document order) sits at row cap-n+j.  cum: c_j sits at scores[cap-1-n+j],
    // j = 0..n (c_0 at node (0,0)); cost_j = c_{j+1} - c_j.
    const int nt = blockDim.x;
    for (int base = 0; base < n; base += nt) {
        const int j = base + threadIdx.x;
        int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        double s = 0.0;
        if (j < n) {
            const int* r = align + 4 * (size_t)(cap - n + j);
            r0 = r[0]; r1 = r[1]; r2 = r[2]; r3 = r[3];
            const double cost = scores[cap - 1 - n + j + 1] - scores[cap - 1 - n + j];
            s = cost < 0.0 ? 0.0 : cost;  // np.clip(a_min=0)
            if (r1 == 0 || r3 == 0) s = 0.0;
            else s = s / (double)r1 / (double)r3;
        }
        __syncthreads();
        if (j < n) {
            int* w = align + 4 * (size_t)j;
            w[0] = r0; w[1] = r1; w[2] = r2; w[3] = r3;
            scores[j] = s;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = n;
}

__global__ __launch_bounds__(64) void k_sparse_traceback(const double* csum, const int* xp, const int* yp, const int* boff,
                                                         int Aout, int B, int xs, int ys, int* align, double* scores,
                                                         int* count) {
    sparse_traceback_block(csum, xp, yp, boff, Aout, B, xs, ys, align, scores, count, nullptr);
}

__global__ __launch_bounds__(64) void k_sparse_traceback_batch(const SvxPairDev* __restrict__ pairs, int depth, int B) {
    const SvxPairDev& P = pairs[blockIdx.x];
    if (depth > P.L || (depth == P.L && P.L > 0)) return;
    if (*P.status != 0) return;
    const SvxLevel& Lv = P.lev[depth];
    const int A = *Lv.path_len;
    if (A <= 0) return;
    sparse_traceback_block(Lv.csum, Lv.xp, Lv.yp, Lv.boff_out, A + 2, B, Lv.n[0], Lv.n[1], Lv.align, Lv.scores, Lv.n_align,
                           P.status);
}

// ------------------------------------------------------------------------------ search path
// append_slant (dp_utils.py:177-196); python round() is round-half-even = rint in the default mode
__device__ int slant(int* path, int n, int cap, int xw, int yw) {
    const int NN = xw + yw;
    const int xs = path[2 * (n - 1)], ys = path[2 * (n - 1) + 1];
    int lx = xs, ly = ys;
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
        path[2 * n] = nx;
        path[2 * n + 1] = ny;
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
    path[0] = 0;
    path[1] = 0;
    for (int i = 0; i < n_align; i++) {
        const int* r = align + 4 * (size_t)i;
        const int p = r[1] * f, q = r[3] * f;
        if (r[1] > 0) { const int m = (r[0] + r[1]) * f - 1; if (m > xmax) xmax = m; }
        if (r[3] > 0) { const int m = (r[2] + r[3]) * f - 1; if (m > ymax) ymax = m; }
        if (p > 0 && q > 0) {
            n = slant(path, n, cap, xdel, ydel);
            if (n < 0) return n;
            xdel = 0; ydel = 0;
            n = slant(path, n, cap, p, q);
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
            n = slant(path, n, cap, xdel, ydel);
            if (n < 0) return n;
            xdel = 0; ydel = 0;
            n = slant(path, n, cap, ex, ey);
            if (n < 0) return n;
        }
    }
    return slant(path, n, cap, xdel, ydel);
}

__global__ void k_search_path(const int* align, const int* n_align, int upsample, int size0, int size1, int* path,
                              int cap, int* path_len) {
    if (threadIdx.x == 0) {
        const int na = *n_align;
        *path_len = na < 0 ? na : search_path_thread(align, na, upsample, size0, size1, path, cap);
    }
}

__global__ void k_search_path_batch(const SvxPairDev* __restrict__ pairs, int depth) {
    const SvxPairDev& P = pairs[blockIdx.x];
    if (threadIdx.x != 0) return;
    if (depth > P.L || (depth == P.L && P.L > 0)) return;
    if (*P.status != 0) return;
    const SvxLevel& dst = P.lev[depth];
    const SvxLevel& src = P.lev[P.L == 0 ? 0 : depth + 1];
    const int na = *src.n_align;
    int n = na < 0 ? na : search_path_thread(src.align, na, P.L > 0, dst.n[0], dst.n[1], dst.path, dst.path_cap);
    *dst.path_len = n;
    if (n < 0) *P.status = -n;
}

// ------------------------------------------------------------------------------ deletion penalty
// DeletionKnob (dp_utils.py:50-79) with numpy's arithmetic: float32 bin edges i*(max/1000),
// density histogram, float64 cdf, 27 interior knots k/28, np.interp at `frac`.
__device__ void del_penalty_block(const float* scores, long long n, double frac, double* out) {
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

int svxl_sparse_dp(svx_ctx* ctx, const float* costs, const int* boff_in, int A, int B, const SvxTypes& types, double pen,
                   int xs, int ys, double* csum, int* xp, int* yp, int* boff_out) {
    if (A <= 0 || B <= 0) return SVX_OK;
    SparseDpArgs g{costs, boff_in, A, B, pen, xs, ys, csum, xp, yp, boff_out};
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

int svxl_sparse_traceback(svx_ctx* ctx, const double* csum, const int* xp, const int* yp, const int* boff, int a_out, int B,
                          int xs, int ys, int* align, double* scores, int* count) {
    hipLaunchKernelGGL(k_sparse_traceback, dim3(1), dim3(64), 0, ctx->stream, csum, xp, yp, boff, a_out, B, xs, ys, align,
                       scores, count);
    SVX_LAUNCH_CHECK(ctx, "k_sparse_traceback");
    return SVX_OK;
}

int svxl_sparse_traceback_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int depth, int B) {
    if (n_pairs <= 0) return SVX_OK;
    hipLaunchKernelGGL(k_sparse_traceback_batch, dim3(n_pairs), dim3(64), 0, ctx->stream, pairs, depth, B);
    SVX_LAUNCH_CHECK(ctx, "k_sparse_traceback_batch");
    return SVX_OK;
}

int svxl_search_path(svx_ctx* ctx, const int* align, const int* n_align, int upsample, int size0, int size1, int* path,
                     int cap, int* path_len) {
    hipLaunchKernelGGL(k_search_path, dim3(1), dim3(64), 0, ctx->stream, align, n_align, upsample, size0, size1, path, cap,
                       path_len);
    SVX_LAUNCH_CHECK(ctx, "k_search_path");
    return SVX_OK;
}

int svxl_search_path_batch(svx_ctx* ctx, const SvxPairDev* pairs, int n_pairs, int depth) {
    if (n_pairs <= 0) return SVX_OK;
    hipLaunchKernelGGL(k_search_path_batch, dim3(n_pairs), dim3(64), 0, ctx->stream, pairs, depth);
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

/*
 * svx_oracle.c -- CPU restatement of the Speech-Vecalign segment-alignment hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product (speech-vecalign_amd/) never links,
 * imports or calls it and has no CPU fallback.
 *
 * Every function restates one function of the reference and cites the file:line it follows
 * (paths relative to /root/reference).  Arithmetic types and evaluation order follow the C that
 * Cython generates from dp_core.pyx (checked by reading the generated C in the build container):
 *   - dot products: float accumulator, one float multiply then one float add per element,
 *     j ascending (no FMA contraction: build with -ffp-contract=off);
 *   - cost formula in double, stored to float;
 *   - DP sums in double, strict '<' tie-breaking in transition order.
 * The numpy-side helpers (dp_utils.py) are restated with numpy's own reduction orders
 * (pairwise float32 sum for contiguous rows, row-sequential float32 sum for axis=0).
 *
 * Pinning: tests/test_oracle_vs_reference.py runs every function here against the real
 * reference (oracle/ref_loader.py) in the build container, bit-exact for all integer, float32
 * and float64 outputs; tests/golden/ holds fixtures generated from the real reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared svx_oracle.c -o liborc.so -lm   (oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_OVERLAPS 1   /* alignment_types need more overlap layers than vecs provide */
#define ORC_ERR_TRACEBACK 2  /* 'traceback bug' / walked off the band */
#define ORC_ERR_BP 3         /* 'got unknown value' */
#define ORC_ERR_EXTEND 4     /* 'asked to extend alignments but already bigger than requested' */
#define ORC_ERR_PATH 5       /* search path index outside the cost array */

/* ------------------------------------------------------------------------------------------
 * numpy float32 pairwise summation of a contiguous vector (numpy/_core/src/umath/loops_utils.h.src,
 * FLOAT_pairwise_sum; PW_BLOCKSIZE = 128).  Used by ndarray.sum() on a contiguous row, which is
 * what dp_utils.py:39 `np.square(v).sum()` calls.
 * ---------------------------------------------------------------------------------------- */
static float np_pairwise_sum_f32(const float *a, long n)
{
    if (n < 8) {
        float res = 0.0f;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum_f32(a, n2) + np_pairwise_sum_f32(a + n2, n - n2);
    }
}

/* dp_utils.py:32-40 make_norm1: v[row,:] /= (sqrt(sum(v*v)) + 1e-5), all float32, in place. */
void orc_make_norm1(float *v, long rows, int d)
{
    float *sq = (float *)malloc(sizeof(float) * (size_t)(d > 0 ? d : 1));
    const float eps = (float)1e-5; /* NEP-50: python float is weak, the add happens in float32 */
    for (long r = 0; r < rows; r++) {
        float *p = v + r * (long)d;
        for (int j = 0; j < d; j++) sq[j] = p[j] * p[j];
        float norm = sqrtf(np_pairwise_sum_f32(sq, d));
        float den = norm + eps;
        for (int j = 0; j < d; j++) p[j] = p[j] / den;
    }
    free(sq);
}

/* dp_utils.py:362-378 downsample_vectors: pair sums (odd tail dropped), minus per-layer
 * column mean (np.mean axis=0: row-sequential float32 sum, then / count), then make_norm1. */
void orc_downsample(const float *v, int K, int n, int d, float *half)
{
    int h = n / 2;
    float *mean = (float *)malloc(sizeof(float) * (size_t)(d > 0 ? d : 1));
    for (int k = 0; k < K; k++) {
        const float *src = v + (long)k * n * d;
        float *dst = half + (long)k * h * d;
        for (int j = 0; j < h; j++)
            for (int c = 0; c < d; c++)
                dst[(long)j * d + c] = src[(long)(2 * j) * d + c] + src[(long)(2 * j + 1) * d + c];
        for (int c = 0; c < d; c++) mean[c] = 0.0f;
        for (int j = 0; j < h; j++)
            for (int c = 0; c < d; c++) mean[c] += dst[(long)j * d + c];
        for (int c = 0; c < d; c++) mean[c] = mean[c] / (float)h;
        for (int j = 0; j < h; j++)
            for (int c = 0; c < d; c++) dst[(long)j * d + c] = dst[(long)j * d + c] - mean[c];
    }
    free(mean);
    orc_make_norm1(half, (long)K * h, d);
}

/* dp_core.pyx:36-77 make_dense_costs */
void orc_dense_costs(const float *v0, int s0, const float *v1, int s1, int d,
                     const float *n0, const float *n1, int off0, int off1, float *costs)
{
    const float *a0 = v0 + (long)off0 * s0 * d;
    const float *b0 = v1 + (long)off1 * s1 * d;
    const float *na = n0 + (long)off0 * s0;
    const float *nb = n1 + (long)off1 * s1;
    for (int xi = 0; xi < s0; xi++) {
        for (int yi = 0; yi < s1; yi++) {
            const float *a = a0 + (long)xi * d, *b = b0 + (long)yi * d;
            float sumx = 0.0f;
            for (int j = 0; j < d; j++) sumx = sumx + a[j] * b[j];
            float c = (float)((2.0 * (1.0 - (double)sumx)) / ((1e-6 + (double)na[xi]) + (double)nb[yi]));
            c = (c * (float)(off0 + 1)) * (float)(off1 + 1); /* float * int -> float (pyx:75) */
            costs[(long)xi * s1 + yi] = c;
        }
    }
}

/* dp_core.pyx:79-141 dense_dp.  pen is a C float in the reference signature. */
void orc_dense_dp(const float *cost, int s0, int s1, float pen, double *csum, int32_t *bp)
{
    int rmax = s0 + 1, cmax = s1 + 1;
#define CS(r, c) csum[(long)(r) * cmax + (c)]
#define BP(r, c) bp[(long)(r) * cmax + (c)]
    for (int c = 0; c < cmax; c++) { CS(0, c) = (double)((float)c * pen); BP(0, c) = 1; }
    for (int r = 0; r < rmax; r++) { CS(r, 0) = (double)((float)r * pen); BP(r, 0) = 2; }
    CS(0, 0) = 0.0;
    BP(0, 0) = 4;
    for (int c = 1; c < cmax; c++) {
        for (int r = 1; r < rmax; r++) {
            double cost0 = CS(r - 1, c - 1) + (double)cost[(long)(r - 1) * s1 + (c - 1)];
            double cost1 = CS(r, c - 1) + (double)pen;
            double cost2 = CS(r - 1, c) + (double)pen;
            double best = cost0;
            int b = 0;
            if (cost1 < best) { best = cost1; b = 1; }
            if (cost2 < best) { best = cost2; b = 2; }
            CS(r, c) = best;
            BP(r, c) = b;
        }
    }
#undef CS
#undef BP
}

/* dp_core.pyx:143-161 score_path (denominator is a float add, no epsilon) */
void orc_score_path(const int32_t *xx, const int32_t *yy, long n, const float *norm1, const float *norm2,
                    const float *v1, const float *v2, int d, float *out)
{
    for (long i = 0; i < n; i++) {
        const float *a = v1 + (long)xx[i] * d, *b = v2 + (long)yy[i] * d;
        float outx = 0.0f;
        for (int j = 0; j < d; j++) outx = outx + a[j] * b[j];
        float den = norm1[xx[i]] + norm2[yy[i]];
        out[i] = (float)((2.0 * (1.0 - (double)outx)) / (double)den);
    }
}

/* dp_core.pyx:165-267 make_sparse_costs.  path = A x 2 int32 (x,y); types = T x 2 (x,y >= 1).
 * feats[T][A][B] with B = 2W; b_offset[A].  Rows are indexed by aa = x + y as in the reference. */
int orc_sparse_costs(const float *v0, int k0, int xsize, const float *v1, int k1, int ysize, int d,
                     const float *n0, const float *n1, const int32_t *path, int A,
                     const int32_t *types, int T, int W, float *feats, int32_t *b_offset)
{
    int maxx = 0, maxy = 0;
    for (int t = 0; t < T; t++) {
        if (types[2 * t] > maxx) maxx = types[2 * t];
        if (types[2 * t + 1] > maxy) maxy = types[2 * t + 1];
    }
    if (maxx > k0 || maxy > k1) return ORC_ERR_OVERLAPS;
    int B = 2 * W;
    for (int ii = 0; ii < A; ii++) {
        int x = path[2 * ii], y = path[2 * ii + 1];
        int aa = x + y;
        if (aa < 0 || aa >= A) return ORC_ERR_PATH;
        b_offset[aa] = y - W;
        for (int b = 0; b < B; b++) {
            int yy = y - W + b;
            int xx = aa - yy;
            for (int t = 0; t < T; t++) {
                int xo = types[2 * t], yo = types[2 * t + 1];
                float feat;
                if (0 <= xx && xx < xsize && 0 <= yy && yy < ysize) {
                    const float *a = v0 + ((long)(xo - 1) * xsize + xx) * d;
                    const float *bv = v1 + ((long)(yo - 1) * ysize + yy) * d;
                    float sumx = 0.0f;
                    for (int j = 0; j < d; j++) sumx = sumx + a[j] * bv[j];
                    feat = (float)((((2.0 * xo) * yo) * (1.0 - (double)sumx)) /
                                   ((1e-6 + (double)n0[(long)(xo - 1) * xsize + xx]) +
                                    (double)n1[(long)(yo - 1) * ysize + yy]));
                } else {
                    feat = INFINITY;
                }
                feats[((long)t * A + aa) * B + b] = feat;
            }
        }
    }
    return ORC_OK;
}

/* dp_core.pyx:269-404 sparse_dp.  Outputs csum/xp/yp are (A+2) x B, b_offset_out is A+2. */
void orc_sparse_dp(const float *costs, const int32_t *boff_in, int A, int B, const int32_t *types, int T,
                   double pen, int x_in, int y_in, double *csum, int32_t *xp, int32_t *yp, int32_t *boff_out)
{
    int NT = T + 2;
    int32_t *xo = (int32_t *)malloc(sizeof(int32_t) * (size_t)NT);
    int32_t *yo = (int32_t *)malloc(sizeof(int32_t) * (size_t)NT);
    for (int t = 0; t < T; t++) { xo[t] = types[2 * t]; yo[t] = types[2 * t + 1]; }
    xo[T] = 0; yo[T] = 1;
    xo[T + 1] = 1; yo[T + 1] = 0;
    int Aout = A + 2;
    int x_out = x_in + 1, y_out = y_in + 1;
    boff_out[0] = boff_in[0];
    boff_out[1] = boff_in[0];
    for (int a = 0; a < A; a++) boff_out[a + 2] = boff_in[a] + 1;
    for (int a = 0; a < Aout; a++) {
        for (int b = 0; b < B; b++) {
            long o = (long)a * B + b;
            int yy = b + boff_out[a];
            int xx = a - yy;
            if (xx == 0 && 0 <= yy && yy < y_out) {
                csum[o] = pen * yy; xp[o] = 0; yp[o] = 1;
            } else if (yy == 0 && 0 <= xx && xx < x_out) {
                csum[o] = pen * xx; xp[o] = 1; yp[o] = 0;
            } else {
                double best = INFINITY;
                int bx = -42, by = -42;
                for (int t = 0; t < NT; t++) {
                    int xc = xx - 1, yc = yy - 1;
                    int xpv = xx - xo[t], ypv = yy - yo[t];
                    if (0 <= xc && xc < x_in && 0 <= yc && yc < y_in && 0 <= xpv && xpv < x_out && 0 <= ypv && ypv < y_out) {
                        int ac = xc + yc;
                        int ap = xpv + ypv;
                        /* the reference indexes b_offset before its range check (boundscheck off);
                           guard the read, the result is the same because the check below fails */
                        if (!(0 <= ac && ac < A && 0 <= ap && ap < Aout)) continue;
                        int bc = yc - boff_in[ac];
                        int bpv = ypv - boff_out[ap];
                        if (0 <= bc && bc < B && 0 <= bpv && bpv < B) {
                            double ac_cost = (xo[t] == 0 || yo[t] == 0) ? pen : (double)costs[((long)t * A + ac) * B + bc];
                            double tot = csum[(long)ap * B + bpv] + ac_cost;
                            if (tot < best) { best = tot; bx = xo[t]; by = yo[t]; }
                        }
                    }
                }
                csum[o] = best; xp[o] = bx; yp[o] = by;
            }
        }
    }
    free(xo);
    free(yo);
}

/* dp_utils.py:146-174 dense_traceback.  bp is (s0+1) x (s1+1).  Alignments are written in
 * document order as rows (x_start, x_len, y_start, y_len); returns the count or -err. */
int orc_dense_traceback(const int32_t *bp, int s0, int s1, int32_t *out)
{
    int cmax = s1 + 1;
    int xx = s0, yy = s1, n = 0;
    while (!(xx == 0 && yy == 0)) {
        int b = bp[(long)xx * cmax + yy];
        int32_t *o = out + 4 * (long)n;
        if (b == 0)      { o[0] = xx - 1; o[1] = 1; o[2] = yy - 1; o[3] = 1; xx--; yy--; }
        else if (b == 1) { o[0] = xx;     o[1] = 0; o[2] = yy - 1; o[3] = 1; yy--; }
        else if (b == 2) { o[0] = xx - 1; o[1] = 1; o[2] = yy;     o[3] = 0; xx--; }
        else return -ORC_ERR_BP;
        n++;
    }
    for (int i = 0, j = n - 1; i < j; i++, j--)
        for (int c = 0; c < 4; c++) { int32_t t = out[4 * i + c]; out[4 * i + c] = out[4 * j + c]; out[4 * j + c] = t; }
    return n;
}

/* dp_utils.py:105-143 sparse_traceback + :89-102 process_scores + :82-86 xy2ab_w_offset.
 * Returns the number of alignments (document order) or -err. */
int orc_sparse_traceback(const double *csum, const int32_t *xp, const int32_t *yp, const int32_t *boff,
                         int Aout, int B, int xsize, int ysize, int32_t *out, double *scores)
{
    int xx = xsize, yy = ysize, n = 0;
    int cap = xsize + ysize + 2;
    double *cum = (double *)malloc(sizeof(double) * (size_t)(cap + 1));
    for (;;) {
        int aa = xx + yy;
        if (aa < 0 || aa >= Aout) { free(cum); return -ORC_ERR_TRACEBACK; }
        int bb = yy - boff[aa];
        if (bb < 0 || bb >= B) { free(cum); return -ORC_ERR_TRACEBACK; }
        long o = (long)aa * B + bb;
        cum[n] = csum[o];
        if (xx == 0 && yy == 0) break;
        if (xx < 0 || yy < 0 || n >= cap) { free(cum); return -ORC_ERR_TRACEBACK; }
        int px = xp[o], py = yp[o];
        if (px < 0 || py < 0 || (px == 0 && py == 0)) { free(cum); return -ORC_ERR_TRACEBACK; }
        int32_t *r = out + 4 * (long)n;
        r[0] = xx - px; r[1] = px; r[2] = yy - py; r[3] = py;
        xx -= px; yy -= py;
        n++;
    }
    /* reverse; cost_i = cum_after - cum_before */
    for (int i = 0, j = n - 1; i < j; i++, j--)
        for (int c = 0; c < 4; c++) { int32_t t = out[4 * i + c]; out[4 * i + c] = out[4 * j + c]; out[4 * j + c] = t; }
    for (int i = 0; i < n; i++) {
        /* cum[] is in traceback order: cum[0] at the end node, cum[n] at (0,0) */
        double cost = cum[n - 1 - i] - cum[n - i];
        double s = cost < 0.0 ? 0.0 : cost; /* np.clip(a_min=0) */
        const int32_t *r = out + 4 * (long)i;
        if (r[1] == 0 || r[3] == 0) s = 0.0;
        else s = s / (double)r[1] / (double)r[3];
        scores[i] = s;
    }
    free(cum);
    return n;
}

/* dp_utils.py:177-196 append_slant (python round() = round-half-even = rint) */
static int slant(int32_t *path, int n, int xw, int yw)
{
    int NN = xw + yw;
    int xs = path[2 * (n - 1)], ys = path[2 * (n - 1) + 1];
    for (int ii = 1; ii <= NN; ii++) {
        int x = xs + (int)rint((double)((long)xw * ii) / (double)NN);
        int y = ys + (int)rint((double)((long)yw * ii) / (double)NN);
        int lx = path[2 * (n - 1)], ly = path[2 * (n - 1) + 1];
        int delta = x + y - lx - ly;
        if (delta == 1)      { path[2 * n] = x;     path[2 * n + 1] = y; n++; }
        else if (delta == 2) { path[2 * n] = x - 1; path[2 * n + 1] = y; n++; }
        else if (delta == 0) { path[2 * n] = x + 1; path[2 * n + 1] = y; n++; }
    }
    return n;
}

/* dp_utils.py:261-275 upsample_alignment, :228-258 extend_alignments, :199-225
 * alignment_to_search_path, fused: only block lengths and the running maxima matter.
 * align = n_align rows (x_start,x_len,y_start,y_len) of the COARSER level when upsample != 0
 * (then size0/size1 are the finer level's sizes), or of the same level when upsample == 0
 * (dp_utils.py:485-486, no extension).  path must hold 2*(size0+size1+4) ints.
 * Returns the path length or -err. */
int orc_search_path(const int32_t *align, int n_align, int upsample, int size0, int size1, int32_t *path)
{
    int n = 1, xdel = 0, ydel = 0;
    int f = upsample ? 2 : 1;
    int xmax = 0, ymax = 0;
    path[0] = 0; path[1] = 0;
    for (int i = 0; i < n_align; i++) {
        const int32_t *r = align + 4 * (long)i;
        int p = r[1] * f, q = r[3] * f;
        if (r[1] > 0) { int m = (r[0] + r[1]) * f - 1; if (m > xmax) xmax = m; }
        if (r[3] > 0) { int m = (r[2] + r[3]) * f - 1; if (m > ymax) ymax = m; }
        if (p > 0 && q > 0) {
            n = slant(path, n, xdel, ydel);
            xdel = 0; ydel = 0;
            n = slant(path, n, p, q);
        } else if (p > 0) xdel += p;
        else if (q > 0) ydel += q;
    }
    if (upsample) {
        if (xmax > size0 || ymax > size1) return -ORC_ERR_EXTEND;
        int ex = size0 - xmax; /* len(range(xmax+1, size0+1)) */
        int ey = size1 - ymax;
        if (ex == 0) ydel += ey;
        else if (ey == 0) xdel += ex;
        else {
            n = slant(path, n, xdel, ydel);
            xdel = 0; ydel = 0;
            n = slant(path, n, ex, ey);
        }
    }
    n = slant(path, n, xdel, ydel);
    return n;
}

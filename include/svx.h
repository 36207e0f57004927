/*
 * svx.h -- C ABI of libsvx: the MI355X (gfx950) implementation of Speech-Vecalign's
 * segment-alignment hot path (svecalign/vecalign + svecalign/seg_align).
 *
 * The reference has exactly one native module on this path, svecalign/vecalign/dp_core.pyx
 * (Cython, 5 functions), called from svecalign/vecalign/dp_utils.py:vecalign().  Each entry
 * point below names the reference interface it replaces (paths relative to the reference
 * repository).  All array arguments are DEVICE pointers (HIP) unless marked "host"; the caller
 * owns every input and output buffer, the library owns only scratch inside the context.
 * Everything is asynchronous on the context's stream; call svx_synchronize() (or synchronise
 * the stream yourself) before reading results.  No function falls back to the CPU.
 *
 * Return value: 0 (SVX_OK) or an SVX_ERR_* code; svx_last_error() has the message.
 */
#ifndef SVX_H
#define SVX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svx_ctx svx_ctx;

enum svx_dtype { SVX_F32 = 0, SVX_F16 = 1, SVX_BF16 = 2 };

enum svx_status {
    SVX_OK = 0,
    SVX_ERR_ARG = 1,       /* bad shape / null pointer / unsupported size (asserts of dp_core.pyx:48-60,186-190,216) */
    SVX_ERR_OVERLAPS = 2,  /* "%d x overlaps requrested (via alignment_types), but vecs0 only has %d" dp_core.pyx:204-209 */
    SVX_ERR_HIP = 3,       /* a HIP runtime call failed */
    SVX_ERR_TRACEBACK = 4, /* 'traceback bug' dp_utils.py:123-124 / walked off the band */
    SVX_ERR_NOMEM = 5,     /* scratch allocation failed */
    SVX_ERR_EXTEND = 6,    /* 'asked to extend alignments but already bigger than requested' dp_utils.py:242-243 */
    SVX_ERR_PATH = 7,      /* search path is not a unit-step lattice path from (0,0) */
    SVX_ERR_BP = 8         /* 'got unknown value' dp_utils.py:166-167 */
};

#define SVX_MAX_LEVELS 16 /* pyramid depth limit (N*M <= max_size_full_dp^2 * 4^15) */
#define SVX_MAX_TYPES 128 /* alignment types per call (a=10 -> 45) */
#define SVX_MAX_DIM 2048  /* embedding dimension limit; d must be a multiple of 8 */

/* ---- context -------------------------------------------------------------------------- */
/* One context per (process, device): owns a HIP stream reference and a grow-only scratch arena.
 * Replaces nothing in the reference (its module is stateless); see SURVEY.md section 8(b). */
int svx_create(int device_id, svx_ctx **out);
int svx_destroy(svx_ctx *ctx);
/* Use an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream); NULL = default stream. */
int svx_set_stream(svx_ctx *ctx, void *hip_stream);
int svx_synchronize(svx_ctx *ctx);
/* Message of the last failure on this context (ctx may be NULL for svx_create failures). */
const char *svx_last_error(const svx_ctx *ctx);
const char *svx_version(void);
/* Bytes of device scratch currently held by the context. */
int64_t svx_scratch_bytes(const svx_ctx *ctx);

/* ---- the five native functions of dp_core.pyx (float32 buffers, like the reference) ----- */

/* make_dense_costs(vecs0, vecs1, norm0, norm1, offset0, offset1) -> costs      dp_core.pyx:36-77
 * vecs0 [k0][s0][d], vecs1 [k1][s1][d], norm0 [k0][s0], norm1 [k1][s1]; costs [s0][s1]. */
int svx_dense_costs(svx_ctx *ctx, const float *vecs0, int k0, int s0, const float *vecs1, int k1, int s1, int d,
                    const float *norm0, const float *norm1, int offset0, int offset1, float *costs);

/* dense_dp(alignment_cost, pen) -> (csum, bp)                                   dp_core.pyx:79-141
 * cost [s0][s1]; csum [s0+1][s1+1] float64 (may be NULL); bp [s0+1][s1+1] int32. */
int svx_dense_dp(svx_ctx *ctx, const float *cost, int s0, int s1, float pen, double *csum, int32_t *bp);

/* score_path(xx, yy, norm1, norm2, vecs1, vecs2, out)                            dp_core.pyx:143-161
 * xx,yy [n] int32; norm1 [rows1], norm2 [rows2]; vecs1 [rows1][d], vecs2 [rows2][d]; out [n]. */
int svx_score_path(svx_ctx *ctx, const int32_t *xx, const int32_t *yy, int64_t n, const float *norm1,
                   const float *norm2, const float *vecs1, int rows1, const float *vecs2, int rows2, int d, float *out);

/* make_sparse_costs(vecs0, vecs1, norms0, norms1, x_y_path, alignment_types, width_over2)
 *   -> (a_b_feats, b_offset)                                                    dp_core.pyx:165-267
 * path [A][2] int32 (device), types [T][2] int32 (HOST); costs [T][A][2W]; b_offset [A]. */
int svx_sparse_costs(svx_ctx *ctx, const float *vecs0, int k0, int xsize, const float *vecs1, int k1, int ysize,
                     int d, const float *norms0, const float *norms1, const int32_t *path, int A,
                     const int32_t *types_host, int T, int width_over2, float *costs, int32_t *b_offset);

/* sparse_dp(a_b_costs, b_offset_in, alignment_types, del_penalty, x_in_size, y_in_size)
 *   -> (a_b_csum, a_b_xp, a_b_yp, b_offset_out)                                 dp_core.pyx:269-404
 * costs [T][A][B]; csum [A+2][B] float64; xp, yp [A+2][B] int32; b_offset_out [A+2]. */
int svx_sparse_dp(svx_ctx *ctx, const float *costs, const int32_t *b_offset_in, int A, int B,
                  const int32_t *types_host, int T, double del_penalty, int x_in_size, int y_in_size,
                  double *csum, int32_t *xp, int32_t *yp, int32_t *b_offset_out);

/* ---- device versions of the numpy-side hot spots of dp_utils.py ------------------------ */

/* make_norm1(vecs): rows /= (||row|| + 1e-5), in place                          dp_utils.py:32-40 */
int svx_make_norm1(svx_ctx *ctx, float *vecs, int64_t rows, int d);

/* downsample_vectors(vecs) -> half [k][n/2][d]                                  dp_utils.py:362-378 */
int svx_downsample(svx_ctx *ctx, const float *vecs, int k, int n, int d, float *half);

/* compute_norms(vecs0, vecs1, num_samples) -> norms0 [k0][n0]                   dp_utils.py:326-359
 * The random row indices (one np.random.choice per overlap layer of vecs1, dp_utils.py:345-348)
 * are drawn by the caller: idx [k1][samples_per_overlap] int32, indices into vecs1's rows. */
int svx_compute_norms(svx_ctx *ctx, const float *vecs0, int k0, int n0, const float *vecs1, int k1, int n1, int d,
                      const int32_t *idx, int samples_per_overlap, float *norms0);

/* DeletionKnob(samp, 0, max(samp)).percentile_frac_to_del_penalty(frac)  dp_utils.py:43-79,312-313,321
 * scores [n] float32 -> *del_penalty (device double). */
int svx_del_penalty(svx_ctx *ctx, const float *scores, int64_t n, double frac, double *del_penalty);

/* dense_traceback(bp) -> alignments                                            dp_utils.py:146-174
 * bp [s0+1][s1+1]; align [s0+s1][4] int32 rows (x_start, x_len, y_start, y_len) in document
 * order; *count = number of rows, or -SVX_ERR_* on failure. */
int svx_dense_traceback(svx_ctx *ctx, const int32_t *bp, int s0, int s1, int32_t *align, int32_t *count);

/* sparse_traceback(csum, xp, yp, b_offset, xsize, ysize) -> (alignments, scores)  dp_utils.py:105-143
 * (+ process_scores :89-102).  align [xsize+ysize+2][4], scores [xsize+ysize+2] float64 (both are also
 * used as scratch by the walk), *count as above. */
int svx_sparse_traceback(svx_ctx *ctx, const double *csum, const int32_t *xp, const int32_t *yp,
                         const int32_t *b_offset_out, int a_out, int B, int xsize, int ysize, int32_t *align,
                         double *scores, int32_t *count);

/* upsample_alignment + extend_alignments + alignment_to_search_path  dp_utils.py:261-275,228-258,199-225
 * align [*n_align][4] rows of the coarser level (upsample != 0; size0/size1 = finer sizes) or of
 * the same level (upsample == 0, dp_utils.py:485-486).  path [size0+size1+4][2]; *path_len =
 * number of points or -SVX_ERR_*. */
int svx_search_path(svx_ctx *ctx, const int32_t *align, const int32_t *n_align, int upsample, int size0,
                    int size1, int32_t *path, int32_t *path_len);

/* make_doc_embedding's gather (svecalign/utils/embedding_utils.py:164-201): out[r][:] = table[idx[r]][:],
 * or zeros where idx[r] < 0 (PAD / ignored / missing candidate) or where the source row holds a NaN
 * (embedding_utils.py:183-190).  table [n_rows][d], out [n_out][d] of `dtype`; idx [n_out] int32 is the
 * flattened [overlaps][segments] row table (svx_candidate_table, or the Python mirror of make_overlap). */
int svx_gather_rows(svx_ctx *ctx, const void *table, int64_t n_rows, int d, int dtype, const int32_t *idx, int64_t n_out,
                    void *out);

/* ---- host-side helpers (HOST pointers, no device work, callable from any thread) ---------
 * What a driver needs besides the kernels to keep a GPU fed: the random row indices, the candidate index
 * table of a document and the text of an alignment file. */

/* np.random.RandomState.choice(n, size, replace=True) (the call of dp_utils.py:301-302,346), bit for bit: MT19937
 * 32-bit outputs with masked rejection.  key[624] / *pos are RandomState.get_state()[1] / [2], advanced in place. */
int svx_mt19937_choice(uint32_t *key, int32_t *pos, int64_t n, int64_t size, int32_t *out);

/* Lengths of svx_pair.norm_idx / svx_pair.knob_idx for a pair of these sizes. */
int64_t svx_norm_index_count(int n, int m, int k0, int k1, int max_size_full_dp, int num_samps_for_norm, int have_norms0,
                             int have_norms1);
int64_t svx_knob_index_count(int n, int m, int max_size_full_dp, int costs_sample_size);

/* All random row indices of one vecalign() call, in the reference's draw order (dp_utils.py:423-444 then
 * :450-456), laid out as svx_pair.norm_idx / knob_idx document.  have_norms0/1: the depth-0
 * normalisers of that side are given (dp_utils.py:428-444), so its draws are skipped. */
int svx_draw_indices(uint32_t *key, int32_t *pos, int n, int m, int k0, int k1, int max_size_full_dp, int costs_sample_size,
                     int num_samps_for_norm, int have_norms0, int have_norms1, int32_t *norm_idx, int32_t *knob_idx);

/* make_doc_embedding's table for speech segments (overlap_segments=True; embedding_utils.py:106-132, 164-201,
 * read_in_embeddings :93-99, load_ignore_index_file vecalign.py:187-195): table[o][i] = first line of `cat_path`
 * equal to "<start of segment i-o> <end of segment i>", or -1 (slot i < o, ignored from that overlap on, or no
 * such candidate).  table [max_overlaps][cap_lines] (NULL: only *n_lines / *n_candidates are returned).
 * ignore_path may be NULL.  err: optional message buffer. */
int svx_candidate_table(const char *seg_path, const char *cat_path, const char *ignore_path, int max_overlaps, int32_t *table,
                        int cap_lines, int32_t *n_lines, int64_t *n_candidates, char *err, int err_cap);

/* print_alignments (vecalign.py:174-184): "[x ids]:[y ids]:%.6f\n" per row (x_start, x_len, y_start, y_len), Python
 * list syntax; scores may be NULL ("[..]:[..]\n").  Returns the number of bytes needed; they are written to out
 * when they fit in cap. */
int64_t svx_format_alignments(const int32_t *rows, const double *scores, int64_t n, char *out, int64_t cap);

/* ---- global margin scoring of the mined alignments (next row after the alignment path) ---
 * svecalign/postprocess/score_align.py:118-161 (compute_sim_with_nonflat_idx) and the index side of
 * svecalign/postprocess/prep_index.py:153-185 (populate_index), for an exact ("Flat") database held in
 * HBM as unit-norm fp16 / bf16 rows -- the storage of the reference's faiss GPU index with gpu_type
 * "fp16-shard" (score_align.py:48-50). */

#define SVX_MARGIN_RATIO 0
#define SVX_MARGIN_DISTANCE 1

/* populate_index: out[i] = rows[i] / |rows[i]| (faiss.normalize_L2, prep_index.py:180) stored as
 * out_dtype (SVX_F16 | SVX_BF16).  rows [n][d] of `dtype`; d a multiple of 32, at most 1024. */
int svx_unit_rows(svx_ctx *ctx, const void *rows, int dtype, int64_t n, int d, void *out, int out_dtype);

/* index.search(x, k) + the mean over the k neighbours (score_align.py:137-148), exact search:
 * mean_sim[i] = mean of the k largest <q_i / |q_i|, db_j>, j < n_db  (= (2 - mean L2^2) / 2 of the
 * reference for unit rows).  queries [n][d] of q_dtype are normalised on the fly (not in place) and
 * rounded to db_dtype for the MFMA, accumulation in fp32.  db [n_db][d] fp16 / bf16, n_db >= k,
 * 1 <= k <= 64. */
int svx_knn_mean_sim(svx_ctx *ctx, const void *queries, int q_dtype, int64_t n, const void *db, int db_dtype,
                     int64_t n_db, int d, int k, float *mean_sim);

/* The same search with the database arriving shard by shard (one rank's rows at a time on a ring of GPUs, or a
 * corpus larger than one allocation): topk [n][k] floats holds every query's k largest similarities so far, in no
 * particular order (-inf = none yet).  first != 0: the lists start empty; otherwise they continue from `topk`.
 * After the call `topk` covers the shards seen so far; mean_sim (nullable) receives their mean, which after the
 * last shard is svx_knn_mean_sim's result over the concatenated database (score_align.py:137-148, where the
 * reference searches one index holding the whole corpus).  n_db may be smaller than k, or 0. */
int svx_knn_topk_merge(svx_ctx *ctx, const void *queries, int q_dtype, int64_t n, const void *db, int db_dtype,
                       int64_t n_db, int d, int k, float *topk, int first, float *mean_sim);

/* score_align.py:151-160: scores[i] = <x_i/|x_i|, y_i/|y_i|> / ((mean_xy[i] + mean_yx[i]) / 2)
 * (SVX_MARGIN_RATIO) or minus it (SVX_MARGIN_DISTANCE).  x, y [n][d] of `dtype`. */
int svx_margin_scores(svx_ctx *ctx, const void *x, const void *y, int dtype, int64_t n, int d, const float *mean_xy,
                      const float *mean_yx, int margin, float *scores);

/* ---- the whole of dp_utils.vecalign() for a batch of document pairs -------------------- */

typedef struct svx_align_params {
    int32_t dtype;            /* svx_dtype of vecs0/vecs1 */
    int32_t d;                /* embedding dimension */
    int32_t n_types;          /* final_alignment_types, in the order of vecalign.py:154-162 */
    int32_t types[2 * SVX_MAX_TYPES];
    int32_t width_over2;      /* dp_utils.py:391-393: values < 3 are raised to 3 */
    int32_t max_size_full_dp; /* dp_utils.py:403-408 */
    int32_t costs_sample_size;
    int32_t num_samps_for_norm;
    double del_percentile_frac;
    int32_t search_mode;      /* SVX_SEARCH_* (0 = the reference's coarse-to-fine recursion) */
    int32_t reserved0;
} svx_align_params;

/* search_mode.  COARSE_TO_FINE: dp_utils.vecalign() as the reference runs it.  STRAIGHT: the same recurrence
 * (make_sparse_costs + sparse_dp + sparse_traceback, final alignment types, depth-0 normalisers and deletion
 * penalty, same random draws as a vecalign() call without pyramid levels) in a band of half-width width_over2
 * around the straight line from (0,0) to (N,M) (append_slant, dp_utils.py:177-196) instead of around an up-sampled
 * coarse alignment: Sakoe-Chiba search (BASELINE configs[3]), and with width_over2 > max(N, M) every cell of
 * the lattice (what vecalign() evaluates with max_size_full_dp = infinity; SURVEY.md 8a, Modes B / C).  Bands wider
 * than 64 cells run as a wavefront of 32 x 32 tiles over all CUs, the costs of a tile computed on the matrix cores
 * and consumed from LDS (no [T][A][B] cost tensor in memory). */
#define SVX_SEARCH_COARSE_TO_FINE 0
#define SVX_SEARCH_STRAIGHT 1

typedef struct svx_pair {
    const void *vecs0, *vecs1; /* [k0][n][d], [k1][m][d] of params.dtype; NOT modified */
    int32_t n, m, k0, k1;
    /* Random row indices in the reference's draw order.  norm_idx: for each depth
     * 0..L: [k1][S1] indices into side 1 (used for n0), then [k0][S0] into side 0 (used for n1),
     * S1 = ceil(num_samps/k1), S0 = ceil(num_samps/k0); a side whose norms are overridden at depth 0
     * contributes no indices at depth 0.  knob_idx: for each depth: x[c] then y[c], c =
     * svx_knob_count(n_l, m_l, costs_sample_size) (full enumeration when n_l*m_l < sample size). */
    const int32_t *norm_idx;
    const int32_t *knob_idx;
    const float *norms0, *norms1; /* optional depth-0 overrides [k0][n], [k1][m] (dp_utils.py:428-444) */
    /* outputs */
    int32_t *align;   /* [n+m+2][4] rows (x_start, x_len, y_start, y_len) */
    double *scores;   /* [n+m+2] */
    int32_t *info;    /* [2]: info[0] = number of alignments, info[1] = 0 or SVX_ERR_* */
    double *del_pen;  /* optional [L+1]: deletion penalty per depth */
} svx_pair;

/* Number of halvings dp_utils.py:403-408 performs (max_depth). */
int svx_num_levels(int n, int m, int max_size_full_dp);
/* Length of each of the two knob index arrays at a level of sizes (n_l, m_l): dp_utils.py:286-302. */
int64_t svx_knob_count(int n_l, int m_l, int costs_sample_size);

/* vecalign(vecs0, vecs1, final_alignment_types, del_percentile_frac, width_over2,
 *          max_size_full_dp, costs_sample_size, num_samps_for_norm, norms0, norms1) -> stack
 *                                                                              dp_utils.py:381-537
 * for n_pairs independent document pairs (pairs: HOST array).  Produces stack[0]
 * ['final_alignments'] and ['alignment_scores'] per pair. */
int svx_align_batch(svx_ctx *ctx, const svx_align_params *params, const svx_pair *pairs, int n_pairs);

/* Per-level intermediates of the LAST svx_align_batch call on this context -- the entries of the `stack` the reference
 * returns (dp_utils.py:412-537) -- as device pointers into the context's scratch arena (valid until the next call;
 * the scalar fields are read back, which synchronises the stream).  Pointers are NULL for what a level does not have
 * (the coarsest level of a pyramid only has normalisers, penalty and alignments; the tile sweep keeps no cost array). */
typedef struct svx_level_view {
    int32_t size0, size1, k0, k1;   /* rows and overlap layers of the two sides at this depth */
    int32_t n_types, band;          /* alignment types of this depth, cells per diagonal (2 * width_over2) */
    int32_t path_len, n_align;      /* search path points (= len(b_offset)); alignments of this depth */
    const float *n0, *n1;           /* [k0][size0], [k1][size1]   stack[d]['n0'], ['n1'] */
    const double *del_penalty;      /* [1]                        stack[d]['del_penalty'] */
    const int32_t *searchpath;      /* [path_len][2]              stack[d]['searchpath'] */
    const float *a_b_costs;         /* [path_len][n_types][band]  stack[d]['a_b_costs'], diagonal-major ([T][A][B] in the reference) */
    const int32_t *b_offset;        /* [path_len]                 stack[d]['b_offset'] */
    const double *a_b_csum;         /* [path_len + 2][band]       stack[d]['a_b_csum'] */
    const uint8_t *a_b_bp;          /* [path_len + 2][band] xp << 4 | yp, 0xFF = -42, or NULL when ... */
    const int32_t *a_b_xp, *a_b_yp; /* ... the types do not pack into 4 bits (then these are set) */
    const int32_t *new_b_offset;    /* [path_len + 2]             stack[d]['new_b_offset'] */
    const int32_t *alignments;      /* [n_align][4] rows (x_start, x_len, y_start, y_len) */
    const double *alignment_scores; /* [n_align] (refined levels) */
    /* the coarsest level of a pyramid (dp_utils.py:465-473), NULL elsewhere: */
    const float *costs_1to1;        /* [size0][size1]             stack[max_depth]['costs_1to1'] (make_dense_costs) */
    const int32_t *x_y_tb_diag;     /* stack[max_depth]['x_y_tb'] (dense_dp back-pointers 0 / 1 / 2, 4 at the origin), stored by
                                     * anti-diagonal: entry (x, y) of the reference's [size0+1][size1+1] array is at
                                     * [(x + y) * (size0 + 1) + x] */
    /* levels >= 1: layer 0 of the level's normalised vectors, [size0][d] / [size1][d] float32 = stack[d]['v0'][0], ['v1'][0]
     * (the other layers of a level are consumed by the pass that forms them and are not kept); NULL at level 0, where the
     * vectors are the caller's inputs */
    const float *v0_l0, *v1_l0;
    /* the sampled 1-1 costs the deletion penalty of this level was estimated from (make_del_knob, dp_utils.py:278-323),
     * in the order of the caller's knob_idx for this level */
    const float *knob_scores;       /* [n_knob] */
    int32_t n_knob, reserved;
} svx_level_view;
int svx_debug_level(svx_ctx *ctx, int pair, int level, svx_level_view *out);
/* Synchronous device -> host copy behind the context's stream (for reading svx_level_view arrays without a tensor library). */
int svx_copy_to_host(svx_ctx *ctx, void *dst_host, const void *src_device, int64_t bytes);

/* Device time of the named stage of the last svx_align_batch call, in milliseconds, measured with
 * HIP events on the context's stream when profiling is on (svx_set_profiling); one name per kernel:
 * "pyr0" "pyr1" "pyrN" "pyr_aux" "knob_sort" "knob_scores0" "knob_scoresN" "knob" "dense_costs" "dense_dp" "path"
 * "band_costs0" "band_costsN" "band_dp0" "band_dpN" "path0" "traceback0" "traceback" "setup" "total" (0 = level 0, N = the
 * deeper levels; "path" and "traceback" without a digit are the deeper levels),
 * "tiles" (the wide-band tile sweep of SVX_SEARCH_STRAIGHT: costs + DP);
 * "host_plan"/"host_launch" are host wall-clock.  -1 if unknown.
 * svx_set_profiling(ctx, 2): accumulate -- the events are recorded as with 1 but a call neither reads them nor waits
 * for its stream, so calls keep queueing behind one another; svx_stage_ms / svx_stage_launches then return the
 * totals over all svx_align_batch calls since profiling was set to 2 (the first query synchronises the stream). */
int svx_set_profiling(svx_ctx *ctx, int on);
/* Software pipeline over consecutive svx_align_batch calls (default off).  The path of one document pair is a
 * streaming front (pyramid, sampled scores: HBM-bound) followed by a refinement chain whose DP / traceback / search-path
 * kernels are serial per pair and leave the memory system idle.  With the pipeline on, a call cuts its batch into two
 * halves; every streaming kernel runs on the context's stream in one fixed order, the latency-bound kernels of one
 * half run beside the streaming kernels of the other on an internal stream, and the refinement chain of the call's
 * SECOND half is held back to run beside the next call's front.  Contract while it is on: the outputs (and the info
 * words) of a call are complete, in the order of the context's stream, only after svx_flush() (or svx_synchronize /
 * svx_debug_level, which flush); inputs and outputs of a call must stay alive until then.  Results do not depend on
 * the setting.  Wide straight bands (the tile sweep, a persistent kernel over all CUs) always run unpipelined. */
int svx_set_pipeline(svx_ctx *ctx, int on);
/* Launch whatever the pipeline still holds back and make the context's stream wait for it (no host synchronisation).
 * A no-op when nothing is in flight. */
int svx_flush(svx_ctx *ctx);
double svx_stage_ms(svx_ctx *ctx, const char *stage);
/* Number of launches of the named stage in the last batch. */
int svx_stage_launches(svx_ctx *ctx, const char *stage);

#ifdef __cplusplus
}
#endif
#endif /* SVX_H */

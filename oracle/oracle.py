"""CPU oracle for the Speech-Vecalign segment-alignment hot path (numpy + oracle/liborc.so).

TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; the product package never does.

It restates, with the reference's arithmetic types and evaluation order, the functions of
  /root/reference/svecalign/vecalign/dp_core.pyx   (C: svx_oracle.c)
  /root/reference/svecalign/vecalign/dp_utils.py   (here + svx_oracle.c)
Each function cites the reference lines it follows.  Parity is PINNED: tests/
test_oracle_vs_reference.py compares every function and the whole pipeline bit-for-bit with the
real reference imported in the build container (oracle/ref_loader.py), and tests/golden/*.npz
hold outputs of the real reference for the GPU box, where the reference does not exist.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_ci, _cl, _cf, _cd = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double


def build():
    """Compile oracle/liborc.so (gcc, seconds).  Called by __graft_entry__.build()."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.orc_make_norm1.argtypes = [_f32p, _cl, _ci]
        L.orc_make_norm1.restype = None
        L.orc_downsample.argtypes = [_f32p, _ci, _ci, _ci, _f32p]
        L.orc_downsample.restype = None
        L.orc_dense_costs.argtypes = [_f32p, _ci, _f32p, _ci, _ci, _f32p, _f32p, _ci, _ci, _f32p]
        L.orc_dense_costs.restype = None
        L.orc_dense_dp.argtypes = [_f32p, _ci, _ci, _cf, _f64p, _i32p]
        L.orc_dense_dp.restype = None
        L.orc_score_path.argtypes = [_i32p, _i32p, _cl, _f32p, _f32p, _f32p, _f32p, _ci, _f32p]
        L.orc_score_path.restype = None
        L.orc_sparse_costs.argtypes = [_f32p, _ci, _ci, _f32p, _ci, _ci, _ci, _f32p, _f32p, _i32p, _ci,
                                       _i32p, _ci, _ci, _f32p, _i32p]
        L.orc_sparse_costs.restype = _ci
        L.orc_sparse_dp.argtypes = [_f32p, _i32p, _ci, _ci, _i32p, _ci, _cd, _ci, _ci, _f64p, _i32p, _i32p, _i32p]
        L.orc_sparse_dp.restype = None
        L.orc_dense_traceback.argtypes = [_i32p, _ci, _ci, _i32p]
        L.orc_dense_traceback.restype = _ci
        L.orc_sparse_traceback.argtypes = [_f64p, _i32p, _i32p, _i32p, _ci, _ci, _ci, _ci, _i32p, _f64p]
        L.orc_sparse_traceback.restype = _ci
        L.orc_search_path.argtypes = [_i32p, _ci, _ci, _ci, _ci, _i32p]
        L.orc_search_path.restype = _ci
        _LIB = L
    return _LIB


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _types_arr(alignment_types):
    for x, y in alignment_types:  # dp_core.pyx:24-34
        assert x > 0 and y > 0
    return np.array(list(alignment_types), dtype=np.int32).reshape(-1, 2)


# ----------------------------------------------------------------------------- dp_core.pyx
def make_dense_costs(vecs0, vecs1, norm0, norm1, offset0=0, offset1=0):
    """dp_core.pyx:36-77"""
    assert vecs0.shape[0] > offset0 and vecs1.shape[0] > offset1
    assert norm0.shape[0] > offset0 and norm1.shape[0] > offset1
    s0, s1, d = vecs0.shape[1], vecs1.shape[1], vecs0.shape[2]
    assert norm0.shape[1] == s0 and norm1.shape[1] == s1 and vecs1.shape[2] == d
    costs = np.empty((s0, s1), dtype=np.float32)
    lib().orc_dense_costs(_c(vecs0, np.float32), s0, _c(vecs1, np.float32), s1, d,
                          _c(norm0, np.float32), _c(norm1, np.float32), offset0, offset1, costs)
    return costs


def dense_dp(alignment_cost, pen):
    """dp_core.pyx:79-141 (pen is rounded to C float by the reference signature)"""
    s0, s1 = alignment_cost.shape
    csum = np.empty((s0 + 1, s1 + 1), dtype=np.float64)
    bp = np.empty((s0 + 1, s1 + 1), dtype=np.int32)
    lib().orc_dense_dp(_c(alignment_cost, np.float32), s0, s1, float(np.float32(pen)), csum, bp)
    return csum, bp


def score_path(xx, yy, norm1, norm2, vecs1, vecs2, out):
    """dp_core.pyx:143-161 (fills `out`)"""
    lib().orc_score_path(_c(xx, np.int32), _c(yy, np.int32), len(xx), _c(norm1, np.float32),
                         _c(norm2, np.float32), _c(vecs1, np.float32), _c(vecs2, np.float32),
                         vecs1.shape[1], out)


def make_sparse_costs(vecs0, vecs1, norms0, norms1, x_y_path, alignment_types, width_over2):
    """dp_core.pyx:165-267"""
    path = np.array(x_y_path).astype(np.int32).reshape(-1, 2)
    assert vecs0.shape[0] == norms0.shape[0] and vecs1.shape[0] == norms1.shape[0]
    assert vecs0.shape[1] == norms0.shape[1] and vecs1.shape[1] == norms1.shape[1]
    assert vecs0.shape[2] == vecs1.shape[2]
    types = _types_arr(alignment_types)
    T, A, B = len(types), path.shape[0], 2 * width_over2
    feats = np.empty((T, A, B), dtype=np.float32)
    boff = np.empty(A, dtype=np.int32)
    rc = lib().orc_sparse_costs(_c(vecs0, np.float32), vecs0.shape[0], vecs0.shape[1],
                                _c(vecs1, np.float32), vecs1.shape[0], vecs1.shape[1], vecs0.shape[2],
                                _c(norms0, np.float32), _c(norms1, np.float32), _c(path, np.int32), A,
                                _c(types, np.int32), T, width_over2, feats, boff)
    if rc == 1:
        mx = max([0] + [x for x, y in alignment_types])
        my = max([0] + [y for x, y in alignment_types])
        if mx > vecs0.shape[0]:
            raise Exception('%d x overlaps requrested (via alignment_types), but vecs0 only has %d' % (mx, vecs0.shape[0]))
        raise Exception('%d y overlaps requrested (via alignment_types), but vecs1 only has %d' % (my, vecs1.shape[0]))
    if rc != 0:
        raise Exception('search path leaves the cost array (rc=%d)' % rc)
    return feats, boff


def sparse_dp(a_b_costs, b_offset_in, alignment_types, del_penalty, x_in_size, y_in_size):
    """dp_core.pyx:269-404"""
    types = _types_arr(alignment_types)
    T, A, B = a_b_costs.shape
    assert T == len(types)
    csum = np.empty((A + 2, B), dtype=np.float64)
    xp = np.empty((A + 2, B), dtype=np.int32)
    yp = np.empty((A + 2, B), dtype=np.int32)
    bout = np.empty(A + 2, dtype=np.int32)
    lib().orc_sparse_dp(_c(a_b_costs, np.float32), _c(b_offset_in, np.int32), A, B, _c(types, np.int32), T,
                        float(del_penalty), x_in_size, y_in_size, csum, xp, yp, bout)
    return csum, xp, yp, bout


# ----------------------------------------------------------------------------- dp_utils.py
def make_norm1(vecs):
    """dp_utils.py:32-40 (in place; vecs float32 C-contiguous [K, n, d])"""
    assert vecs.dtype == np.float32 and vecs.flags.c_contiguous
    lib().orc_make_norm1(vecs, vecs.shape[0] * vecs.shape[1], vecs.shape[2])


def downsample_vectors(vecs):
    """dp_utils.py:362-378"""
    K, n, d = vecs.shape
    half = np.empty((K, n // 2, d), dtype=np.float32)
    lib().orc_downsample(_c(vecs, np.float32), K, n, d, half)
    return half


def _rows_to_alignments(rows):
    return [(list(range(r[0], r[0] + r[1])), list(range(r[2], r[2] + r[3]))) for r in rows.tolist()]


def _alignments_to_rows(algn):
    rows = np.zeros((len(algn), 4), dtype=np.int32)
    for i, (x, y) in enumerate(algn):
        rows[i] = (x[0] if len(x) else 0, len(x), y[0] if len(y) else 0, len(y))
    # a deletion's start index is irrelevant to everything downstream except printing (empty list)
    return rows


def dense_traceback(x_y_tb):
    """dp_utils.py:146-174"""
    s0, s1 = x_y_tb.shape[0] - 1, x_y_tb.shape[1] - 1
    out = np.empty((s0 + s1 + 1, 4), dtype=np.int32)
    n = lib().orc_dense_traceback(_c(x_y_tb, np.int32), s0, s1, out)
    if n < 0:
        raise Exception('got unknown value')
    return _rows_to_alignments(out[:n])


def sparse_traceback(a_b_csum, a_b_xp, a_b_yp, b_offset, xsize, ysize):
    """dp_utils.py:105-143 (+ process_scores :89-102)"""
    out = np.empty((xsize + ysize + 2, 4), dtype=np.int32)
    scores = np.empty(xsize + ysize + 2, dtype=np.float64)
    n = lib().orc_sparse_traceback(_c(a_b_csum, np.float64), _c(a_b_xp, np.int32), _c(a_b_yp, np.int32),
                                   _c(b_offset, np.int32), a_b_csum.shape[0], a_b_csum.shape[1],
                                   xsize, ysize, out, scores)
    if n < 0:
        raise Exception('traceback bug')
    return _rows_to_alignments(out[:n]), scores[:n].copy()


def search_path(alignments, upsample, size0, size1):
    """dp_utils.py:261-275 (upsample) + :228-258 (extend) + :199-225 (to search path), fused."""
    rows = _alignments_to_rows(alignments)
    path = np.empty((size0 + size1 + 8 + 2 * len(rows), 2), dtype=np.int32)
    n = lib().orc_search_path(rows.reshape(-1) if len(rows) else np.zeros(4, np.int32), len(rows),
                              1 if upsample else 0, size0, size1, path.reshape(-1))
    if n < 0:
        raise Exception('asked to extend alignments but already bigger than requested')
    return [tuple(p) for p in path[:n].tolist()]


def sample_norm_indices(size_other, overlaps_other, num_samples, rng):
    """The draws of dp_utils.py:340-348, one choice() per overlap layer of the OTHER side."""
    spo = math.ceil(num_samples / overlaps_other)
    return [rng.choice(size_other, size=spo, replace=True) for _ in range(overlaps_other)]


def compute_norms(vecs0, vecs1, num_samples, rng):
    """dp_utils.py:326-359 (np.matmul + mean kept in numpy, like the reference)"""
    overlaps1, size1, dim = vecs1.shape
    overlaps0, size0, _ = vecs0.shape
    spo = math.ceil(num_samples / overlaps1)
    if size1 and spo:
        idx = sample_norm_indices(size1, overlaps1, num_samples, rng)
        samp = np.empty((spo * overlaps1, dim), dtype=np.float32)
        for k in range(overlaps1):
            samp[k * spo:(k + 1) * spo, :] = vecs1[k, idx[k], :]
        norms0 = np.empty((overlaps0, size0), dtype=np.float32)
        for k in range(overlaps0):
            sim = np.matmul(vecs0[k, :, :], samp.T)
            norms0[k, :] = 1.0 - sim.mean(axis=1)
    else:
        norms0 = np.ones((overlaps0, size0)).astype(np.float32)
    return norms0


def sample_knob_indices(e_size, f_size, sample_size, rng):
    """The index arrays of dp_utils.py:286-302 (full enumeration when e*f < sample_size)."""
    if e_size * f_size < sample_size:
        x = np.repeat(np.arange(e_size, dtype=np.int32), f_size)
        y = np.tile(np.arange(f_size, dtype=np.int32), e_size)
    else:
        x = rng.choice(e_size, size=sample_size, replace=True).astype(np.int32)
        y = rng.choice(f_size, size=sample_size, replace=True).astype(np.int32)
    return x, y


def del_penalty_from_scores(random_scores, min_score, max_score, knob_val):
    """DeletionKnob, dp_utils.py:43-79: histogram -> cdf -> 29-knot percentile map -> interp."""
    res_min, res_max = min_score, max_score
    if res_min >= res_max:
        res_max = res_min + 1e-4
    hist, edges = np.histogram(random_scores, bins=1000, range=[res_min, res_max], density=True)
    dx = edges[1] - edges[0]
    cdf = np.cumsum(hist) * dx
    xs, ys = [0], [res_min]
    for kv in np.linspace(0, 1, 30 - 1)[1:-1]:
        ci = np.searchsorted(cdf, kv)
        xs.append(kv)
        ys.append(res_min + ci / float(1000) * (res_max - res_min))
    xs.append(1)
    ys.append(res_max)
    return np.interp([knob_val], xs, ys)[0]


def make_del_penalty(e_laser, f_laser, e_norms, f_norms, sample_size, frac, rng):
    """make_del_knob dp_utils.py:278-323 + percentile_frac_to_del_penalty :77-79"""
    e_size, f_size = e_laser.shape[0], f_laser.shape[0]
    if e_size > 0 and f_size > 0 and sample_size > 0:
        x, y = sample_knob_indices(e_size, f_size, sample_size, rng)
        scores = np.empty(len(x), dtype=np.float32)
        score_path(x, y, e_norms, f_norms, e_laser, f_laser, scores)
        lo, hi = 0, max(scores)
    else:
        scores, lo, hi = np.array([0.0, 0.5, 1.0]), 0, 1
    return del_penalty_from_scores(scores, lo, hi, frac), scores


def vecalign(vecs0, vecs1, final_alignment_types, del_percentile_frac, width_over2, max_size_full_dp,
             costs_sample_size, num_samps_for_norm, norms0=None, norms1=None, rng=None, del_penalties=None):
    """dp_utils.py:381-537.  `rng` defaults to numpy's global legacy stream, like the reference;
    vecs0/vecs1 are normalised IN PLACE, like the reference (dp_utils.py:396-397).
    del_penalties (test hook, not in the reference): {depth: value} replaces the estimated deletion penalty of those
    depths AFTER it was estimated (the random stream is consumed as always; stack[d]['del_penalty_estimated'] keeps
    the estimate) -- used to separate a percentile knife-edge of the estimate from everything downstream of it."""
    rng = np.random if rng is None else rng
    if width_over2 < 3:
        width_over2 = 3
    make_norm1(vecs0)
    make_norm1(vecs1)
    s0, s1 = vecs0.shape[1], vecs1.shape[1]
    max_depth = 0
    while s0 * s1 > max_size_full_dp ** 2:
        max_depth += 1
        s0, s1 = s0 // 2, s1 // 2
    stack = {0: {'v0': vecs0, 'v1': vecs1}}
    for depth in range(1, max_depth + 1):
        stack[depth] = {'v0': downsample_vectors(stack[depth - 1]['v0']),
                        'v1': downsample_vectors(stack[depth - 1]['v1'])}
    for depth in stack:
        st = stack[depth]
        st['size0'], st['size1'] = st['v0'].shape[1], st['v1'].shape[1]
        st['alignment_types'] = final_alignment_types if depth == 0 else [(1, 1)]
        if depth == 0 and norms0 is not None:
            if norms0.shape != vecs0.shape[:2]:
                raise Exception('norms0 wrong shape')
            st['n0'] = norms0
        else:
            st['n0'] = compute_norms(st['v0'], st['v1'], num_samps_for_norm, rng)
        if depth == 0 and norms1 is not None:
            if norms1.shape != vecs1.shape[:2]:
                raise Exception('norms1 wrong shape')
            st['n1'] = norms1
        else:
            st['n1'] = compute_norms(st['v1'], st['v0'], num_samps_for_norm, rng)
    for depth in stack:
        st = stack[depth]
        st['del_penalty'], st['knob_scores'] = make_del_penalty(
            st['v0'][0], st['v1'][0], st['n0'][0], st['n1'][0], costs_sample_size, del_percentile_frac, rng)
        if del_penalties is not None and depth in del_penalties:
            st['del_penalty_estimated'] = st['del_penalty']
            st['del_penalty'] = float(del_penalties[depth])
    top = stack[max_depth]
    top['costs_1to1'] = make_dense_costs(top['v0'], top['v1'], top['n0'], top['n1'])
    _, top['x_y_tb'] = dense_dp(top['costs_1to1'], top['del_penalty'])
    top['alignments'] = dense_traceback(top['x_y_tb'])
    depths = [0] if max_depth == 0 else list(reversed(range(0, max_depth)))
    for depth in depths:
        st = stack[depth]
        if max_depth > 0:
            st['searchpath'] = search_path(stack[depth + 1]['alignments'], True, st['size0'], st['size1'])
        else:
            st['searchpath'] = search_path(stack[0]['alignments'], False, st['size0'], st['size1'])
        st['a_b_costs'], st['b_offset'] = make_sparse_costs(st['v0'], st['v1'], st['n0'], st['n1'],
                                                            st['searchpath'], st['alignment_types'], width_over2)
        st['a_b_csum'], st['a_b_xp'], st['a_b_yp'], st['new_b_offset'] = sparse_dp(
            st['a_b_costs'], st['b_offset'], st['alignment_types'], st['del_penalty'], st['size0'], st['size1'])
        akey = 'final_alignments' if depth == 0 else 'alignments'
        st[akey], st['alignment_scores'] = sparse_traceback(st['a_b_csum'], st['a_b_xp'], st['a_b_yp'],
                                                            st['new_b_offset'], st['size0'], st['size1'])
    return stack


# ---------------------------------------------------------------------------------------------
# Margin scoring of the mined alignments (the row after the alignment path).
#   /root/reference/svecalign/postprocess/score_align.py:118-161  compute_sim_with_nonflat_idx
# The reference searches faiss indexes (third-party, absent here: faiss-gpu, no version pinned in the
# reference's README); for the Flat index type the search is exact brute force over the stored
# unit-norm rows, which is what this restates.  Pinned by the reference's own shipped example:
# tests/golden/margin_example.npz holds the rows of example/voxpopuli/*_embed_indexes/en-de/*/
# Flat.populate.idx and the scores of example/voxpopuli/*_margin/en-de/*.txt; this function reproduces
# them to 1.3e-4 (the example was produced by faiss' fp16 GPU search, tests/test_oracle_golden.py).
def round_storage(a, storage):
    """fp32 -> the database's storage type -> fp32 (round to nearest even)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if storage == "fp32":
        return a
    if storage == "fp16":
        return a.astype(np.float16).astype(np.float32)
    if storage == "bf16":
        u = a.view(np.uint32).astype(np.uint64)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return u.astype(np.uint32).view(np.float32)
    raise ValueError(storage)


def normalize_l2(x):
    """faiss.normalize_L2 (score_align.py:133-134): x * (1 / sqrt(sum x^2)) in fp32, zero rows untouched."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    ss = np.einsum("ij,ij->i", x, x, dtype=np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(ss > 0, np.float32(1.0) / np.sqrt(ss, dtype=np.float32), np.float32(0.0)).astype(np.float32)
    return x * inv[:, None]


def knn_mean_sim(queries, db, k, storage="fp16"):
    """index.search + mean (score_align.py:137-148) for unit rows: mean of the k largest <q, db_j>.
    `db` holds stored (already rounded) unit rows; the queries are normalised and rounded the same way."""
    q = round_storage(normalize_l2(queries), storage).astype(np.float64)
    sims = q @ np.asarray(db, dtype=np.float64).T
    top = np.partition(sims, sims.shape[1] - k, axis=1)[:, sims.shape[1] - k:]
    return top.mean(axis=1).astype(np.float32)


def margin_scores(x, y, db_x, db_y, k=16, margin="ratio", storage="fp16"):
    """score_align.py:124-161.  db_x / db_y: the stored rows of the source / target index."""
    xn, yn = normalize_l2(x), normalize_l2(y)
    mean_xy = knn_mean_sim(x, db_y, k, storage)
    mean_yx = knn_mean_sim(y, db_x, k, storage)
    a = np.einsum("ij,ij->i", xn, yn, dtype=np.float32)
    b = (mean_xy + mean_yx) / np.float32(2)
    if margin == "ratio":
        return (a / b).astype(np.float32)
    if margin == "distance":
        return (a - b).astype(np.float32)
    raise ValueError(f"Wrong margin type: {margin}")

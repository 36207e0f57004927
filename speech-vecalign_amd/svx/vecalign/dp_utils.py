"""Drop-in for the reference's svecalign/vecalign/dp_utils.py on MI355X.

`vecalign()` keeps the reference signature (dp_utils.py:381-390) and returns a `stack` dict whose
level 0 holds 'final_alignments' and 'alignment_scores' (what every caller of the reference reads,
vecalign.py:276-285).  The whole recursion -- unit-normalisation, pyramid, normalisers, deletion
penalties, coarse dense DP, path up-sampling, band costs, band DP, tracebacks -- runs on the
device as one batched launch sequence (svx_align_batch, include/svx.h); the only host work is
drawing the random row indices, which uses numpy's legacy global stream in the reference's exact
call order so that `np.random.seed(s)` selects the same samples as the reference.

Differences from the reference, by design:
  * vecs0/vecs1 are NOT normalised in place (the reference mutates its inputs, dp_utils.py:396-397);
    pass `normalize_inputs_inplace=True` to get that side effect for float32 numpy inputs.
  * inputs may be float32, float16 or bfloat16 (numpy or torch CUDA tensors); the reference
    only takes float32 arrays that were up-cast from fp16 files.
  * `align_batch()` aligns many document pairs per call (the reference loops in Python,
    seg_align/align.py:206-230).
"""
import ctypes
import logging
from collections import OrderedDict
from math import ceil

import numpy as np

from .. import _lib

logger = logging.getLogger('vecalign')  # same logger name as the reference (dp_utils.py:29)


# ------------------------------------------------------------------------------------- helpers
_pool = None


def _draw_pool():
    """Threads for the GIL-free native host helpers (index drawing, file parsing)."""
    global _pool
    if _pool is None:
        import os
        from multiprocessing.pool import ThreadPool
        _pool = ThreadPool(max(2, min(16, (os.cpu_count() or 4))))
    return _pool


def _ctx():
    return _lib.context()


def _p(x):
    return ctypes.c_void_p(x.data_ptr()) if x is not None else ctypes.c_void_p(0)


def _dev(ctx, a):
    t = ctx.torch
    if hasattr(a, "data_ptr"):
        return a.to(ctx.tdev).contiguous()
    return t.from_numpy(np.ascontiguousarray(a)).to(ctx.tdev)


def _svx_dtype(ctx, x):
    t = ctx.torch
    if x.dtype == t.float32:
        return _lib.SVX_F32
    if x.dtype == t.float16:
        return _lib.SVX_F16
    if x.dtype == t.bfloat16:
        return _lib.SVX_BF16
    raise ValueError("Buffer dtype mismatch, expected 'float' (or float16/bfloat16) but got '%s'" % x.dtype)


def rows_to_alignments(rows):
    """[(x_start, x_len, y_start, y_len)] -> the reference's list of ([x ids], [y ids])."""
    return [(list(range(r[0], r[0] + r[1])), list(range(r[2], r[2] + r[3]))) for r in np.asarray(rows).tolist()]


def alignments_to_rows(alignments):
    rows = np.zeros((max(1, len(alignments)), 4), dtype=np.int32)
    for i, (x, y) in enumerate(alignments):
        rows[i] = (x[0] if len(x) else 0, len(x), y[0] if len(y) else 0, len(y))
    return rows


def level_sizes(n, m, max_size_full_dp):
    """Sizes per depth, dp_utils.py:403-408."""
    L = _lib.load().svx_num_levels(int(n), int(m), int(max_size_full_dp))
    return [(n >> l, m >> l) for l in range(L + 1)]


# ------------------------------------------------------------------------------------- sampling
def draw_indices(n, m, k0, k1, max_size_full_dp, costs_sample_size, num_samps_for_norm, rng=None,
                 have_norms0=False, have_norms1=False):
    """All random row indices of one vecalign() call, drawn in the reference's order from `rng`
    (default: numpy's global legacy stream, like the reference).

    Order (dp_utils.py:423-444 then :450-456): for every depth, compute_norms(v0, v1) draws one
    choice(range(size1), ceil(num/k1)) per overlap layer of side 1, then compute_norms(v1, v0) draws
    per layer of side 0; after all depths, make_del_knob draws x then y per depth, or enumerates
    all pairs when size0*size1 < sample size.  Returns (norm_idx, knob_idx) int32 arrays laid out
    as include/svx.h:svx_pair documents."""
    rng = np.random if rng is None else rng
    sizes = level_sizes(n, m, max_size_full_dp)
    norm_parts, knob_parts = [], []
    for depth, (s0, s1) in enumerate(sizes):
        for size_other, k_other, skip in ((s1, k1, depth == 0 and have_norms0), (s0, k0, depth == 0 and have_norms1)):
            spo = ceil(num_samps_for_norm / k_other)
            if skip or not (size_other and spo):
                continue
            for _ in range(k_other):
                norm_parts.append(rng.choice(size_other, size=spo, replace=True).astype(np.int32))
    for s0, s1 in sizes:
        if s0 * s1 < costs_sample_size:
            knob_parts.append(np.repeat(np.arange(s0, dtype=np.int32), s1))
            knob_parts.append(np.tile(np.arange(s1, dtype=np.int32), s0))
        else:
            knob_parts.append(rng.choice(s0, size=costs_sample_size, replace=True).astype(np.int32))
            knob_parts.append(rng.choice(s1, size=costs_sample_size, replace=True).astype(np.int32))
    norm_idx = np.concatenate(norm_parts) if norm_parts else np.zeros(1, np.int32)
    knob_idx = np.concatenate(knob_parts)
    return norm_idx, knob_idx


def index_counts(n, m, k0, k1, max_size_full_dp, costs_sample_size, num_samps_for_norm, have_norms0=False, have_norms1=False):
    """(len(norm_idx), len(knob_idx)) of draw_indices for a pair of these sizes."""
    lib = _lib.load()
    return (int(lib.svx_norm_index_count(n, m, k0, k1, max_size_full_dp, num_samps_for_norm, int(have_norms0), int(have_norms1))),
            int(lib.svx_knob_index_count(n, m, max_size_full_dp, costs_sample_size)))


def draw_indices_into(norm_out, knob_out, n, m, k0, k1, max_size_full_dp, costs_sample_size, num_samps_for_norm, rng=None,
                      have_norms0=False, have_norms1=False):
    """draw_indices() written straight into int32 buffers (numpy views, e.g. of pinned staging memory), by the
    native MT19937 restatement of RandomState.choice (svx_draw_indices): the same stream, bit for bit, at a
    fraction of the cost, and without the GIL, so a thread pool can draw for many pairs at once.  `rng` is a
    numpy RandomState (advanced like the numpy calls would) or None for numpy's global legacy stream."""
    lib = _lib.load()
    st = np.random.get_state() if rng is None else rng.get_state()
    if st[0] != 'MT19937':
        raise ValueError("draw_indices_into needs a legacy MT19937 RandomState")
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = ctypes.c_int32(int(st[2]))
    assert norm_out.dtype == np.int32 and knob_out.dtype == np.int32 and norm_out.flags.c_contiguous and knob_out.flags.c_contiguous
    rc = lib.svx_draw_indices(ctypes.c_void_p(key.ctypes.data), ctypes.byref(pos), int(n), int(m), int(k0), int(k1),
                              int(max_size_full_dp), int(costs_sample_size), int(num_samps_for_norm), int(have_norms0),
                              int(have_norms1), ctypes.c_void_p(norm_out.ctypes.data if norm_out.size else 0),
                              ctypes.c_void_p(knob_out.ctypes.data))
    if rc != 0:
        raise ValueError("svx_draw_indices: bad argument")
    new = ('MT19937', key, int(pos.value), st[3], st[4])
    if rng is None:
        np.random.set_state(new)
    else:
        rng.set_state(new)


# ------------------------------------------------------------------------------------- batch
class PreparedBatch:
    """A batch of document pairs resident on the device, ready for `run()` (svx_align_batch).
    Everything the timed path touches -- embeddings, sampled indices, output buffers, descriptors --
    is built here once."""

    def __init__(self, pairs, final_alignment_types, del_percentile_frac, width_over2, max_size_full_dp,
                 costs_sample_size, num_samps_for_norm, rngs=None, norms=None, device=None, search="coarse_to_fine"):
        """search: "coarse_to_fine" (the reference's recursion) or "straight" (band of half-width `width_over2`
        around the straight line from (0,0) to (N,M): Sakoe-Chiba / dense search, include/svx.h SVX_SEARCH_STRAIGHT;
        no pyramid, so `max_size_full_dp` is ignored)."""
        if search not in ("coarse_to_fine", "straight"):
            raise ValueError("search must be 'coarse_to_fine' or 'straight'")
        if search == "straight":
            max_size_full_dp = 1 << 30  # depth 0 only: the random draws of a vecalign() call without pyramid levels
        ctx = _lib.context(device)
        self.ctx = ctx
        t = ctx.torch
        if width_over2 < 3:
            logger.warning('width_over2 was set to %d, which does not make sense. increasing to 3.', width_over2)
            width_over2 = 3
        self.width_over2 = width_over2
        self.types = [tuple(int(v) for v in xy) for xy in final_alignment_types]
        prm = _lib.AlignParams()
        self.vecs = [(_dev(ctx, a), _dev(ctx, b)) for a, b in pairs]
        if not self.vecs:
            raise ValueError("empty batch")
        dts = {_svx_dtype(ctx, v) for ab in self.vecs for v in ab}
        if len(dts) != 1:
            raise ValueError("all embeddings of a batch must share one dtype")
        for a, b in self.vecs:
            if a.dim() != 3 or b.dim() != 3:
                raise ValueError("Buffer has wrong number of dimensions (expected 3, got %d)" % a.dim())
            assert a.shape[2] == b.shape[2]
        prm.dtype = dts.pop()
        prm.d = int(self.vecs[0][0].shape[2])
        prm.n_types = len(self.types)
        if prm.n_types > _lib.SVX_MAX_TYPES:
            raise ValueError("too many alignment types")
        for i, (x, y) in enumerate(self.types):
            assert (x > 0)
            assert (y > 0)
            prm.types[2 * i], prm.types[2 * i + 1] = x, y
        prm.width_over2 = int(width_over2)
        prm.max_size_full_dp = int(max_size_full_dp)
        prm.costs_sample_size = int(costs_sample_size)
        prm.num_samps_for_norm = int(num_samps_for_norm)
        prm.del_percentile_frac = float(del_percentile_frac)
        prm.search_mode = _lib.SVX_SEARCH_STRAIGHT if search == "straight" else _lib.SVX_SEARCH_COARSE_TO_FINE
        self.prm = prm
        npairs = len(self.vecs)
        self.cpairs = (_lib.Pair * npairs)()
        self.keep = []
        self.levels = []
        # one allocation per output kind, sliced per pair
        caps = [int(a.shape[1] + b.shape[1] + 2) for a, b in self.vecs]
        offs = np.concatenate([[0], np.cumsum(caps)]).astype(np.int64)
        self.offs = offs
        self.align = t.zeros((int(offs[-1]), 4), dtype=t.int32, device=ctx.tdev)
        self.scores = t.zeros(int(offs[-1]), dtype=t.float64, device=ctx.tdev)
        self.info = t.zeros((npairs, 2), dtype=t.int32, device=ctx.tdev)
        self.del_pen = t.zeros((npairs, _lib.SVX_MAX_LEVELS), dtype=t.float64, device=ctx.tdev)
        # sampled indices: sizes are a function of the shapes, so one pinned staging buffer per kind is laid out
        # first and filled in place (native generator; per-pair streams are independent, so they fill in parallel)
        norm_offs, knob_offs = [0], [0]
        for i, (a, b) in enumerate(self.vecs):
            k0, n = int(a.shape[0]), int(a.shape[1])
            k1, m = int(b.shape[0]), int(b.shape[1])
            nrm = norms[i] if norms is not None else (None, None)
            nn, kn = index_counts(n, m, k0, k1, max_size_full_dp, costs_sample_size, num_samps_for_norm,
                                  nrm[0] is not None, nrm[1] is not None)
            norm_offs.append(norm_offs[-1] + nn)
            knob_offs.append(knob_offs[-1] + kn)
            self.levels.append(len(level_sizes(n, m, max_size_full_dp)))
        h_norm = t.empty(max(1, norm_offs[-1]), dtype=t.int32, pin_memory=True)
        h_knob = t.empty(max(1, knob_offs[-1]), dtype=t.int32, pin_memory=True)
        hn, hk = h_norm.numpy(), h_knob.numpy()

        def draw(i):
            a, b = self.vecs[i]
            nrm = norms[i] if norms is not None else (None, None)
            draw_indices_into(hn[norm_offs[i]:norm_offs[i + 1]], hk[knob_offs[i]:knob_offs[i + 1]], int(a.shape[1]), int(b.shape[1]),
                              int(a.shape[0]), int(b.shape[0]), max_size_full_dp, costs_sample_size, num_samps_for_norm,
                              rngs[i] if rngs is not None else None, nrm[0] is not None, nrm[1] is not None)
        if rngs is not None and npairs >= 8:
            _draw_pool().map(draw, range(npairs))
        else:  # the global stream is consumed pair after pair, like the reference's serial loop
            for i in range(npairs):
                draw(i)
        self.norm_idx = h_norm.to(ctx.tdev, non_blocking=True)
        self.knob_idx = h_knob.to(ctx.tdev, non_blocking=True)
        self.keep.append((h_norm, h_knob))
        for i, (a, b) in enumerate(self.vecs):
            c = self.cpairs[i]
            c.vecs0, c.vecs1 = a.data_ptr(), b.data_ptr()
            c.k0, c.n = int(a.shape[0]), int(a.shape[1])
            c.k1, c.m = int(b.shape[0]), int(b.shape[1])
            c.norm_idx = self.norm_idx.data_ptr() + 4 * norm_offs[i]
            c.knob_idx = self.knob_idx.data_ptr() + 4 * knob_offs[i]
            nrm = norms[i] if norms is not None else (None, None)
            for j, name in enumerate(("norms0", "norms1")):
                if nrm[j] is not None:
                    side = (a, b)[j]
                    if tuple(nrm[j].shape) != tuple(side.shape[:2]):
                        raise Exception('%s wrong shape' % name)  # dp_utils.py:429-432,438-441
                    dn = _dev(ctx, nrm[j]).to(t.float32).contiguous()
                    self.keep.append(dn)
                    setattr(c, name, dn.data_ptr())
            c.align = self.align.data_ptr() + 16 * int(offs[i])
            c.scores = self.scores.data_ptr() + 8 * int(offs[i])
            c.info = self.info.data_ptr() + 8 * i
            c.del_pen = self.del_pen.data_ptr() + 8 * _lib.SVX_MAX_LEVELS * i

    def run(self):
        """Launch the whole pipeline for the batch (asynchronous on the current torch stream)."""
        ctx = self.ctx
        ctx.use_current_stream()
        ctx.check(ctx.lib.svx_align_batch(ctx.h, ctypes.byref(self.prm), self.cpairs, len(self.vecs)))
        ctx.hold(self)

    def flush(self):
        """With the context's pipeline on (Context.set_pipeline): launch what run() held back and order it in front of
        whatever follows on the current stream."""
        self.ctx.flush()

    def fetch_async(self):
        """Queue the device -> pinned-host copies of the outputs behind run() on the current stream and return
        an event that fires when they have landed (the host pipeline formats batch i while batch i+1 computes)."""
        t = self.ctx.torch
        self.ctx.flush()  # (a no-op unless the pipeline is on)
        self.h_out = tuple(t.empty(x.shape, dtype=x.dtype, pin_memory=True) for x in (self.info, self.align, self.scores, self.del_pen))
        for h, d in zip(self.h_out, (self.info, self.align, self.scores, self.del_pen)):
            h.copy_(d, non_blocking=True)
        ev = t.cuda.Event()
        ev.record(t.cuda.current_stream(self.ctx.tdev))
        return ev

    def raw_results(self):
        """-> (info [P][2], rows [sum][4], scores [sum], del_pen [P][levels], offsets) as numpy arrays, after
        checking every pair's status.  Uses the copies of fetch_async() when there are any."""
        if getattr(self, "h_out", None) is not None:
            info, align, scores, pens = (h.numpy() for h in self.h_out)
        else:
            self.ctx.sync()
            info, align, scores, pens = (x.cpu().numpy() for x in (self.info, self.align, self.scores, self.del_pen))
        bad = np.nonzero((info[:, 1] != 0) | (info[:, 0] < 0))[0]
        if len(bad):
            cnt, err = int(info[bad[0], 0]), int(info[bad[0], 1])
            code = err if err != 0 else -cnt
            raise Exception(_lib.DEVICE_ERRORS.get(code, "device failure %d" % code))
        return info, align, scores, pens, self.offs

    def level_stack(self, i, depth):
        """The reference's stack[depth] entries (dp_utils.py:412-537) of pair `i` of the last run(), copied from
        the device (svx_debug_level): n0, n1, del_penalty, and for refined levels searchpath, a_b_costs [T][A][B],
        b_offset, a_b_csum, a_b_xp, a_b_yp, new_b_offset, alignments, alignment_scores; the coarsest level of a
        pyramid has costs_1to1 and x_y_tb (dp_utils.py:465-473); levels >= 1 have v0_layer0 / v1_layer0 = v0[0] / v1[0]."""
        ctx = self.ctx
        t = ctx.torch
        v = _lib.LevelView()
        ctx.check(ctx.lib.svx_debug_level(ctx.h, int(i), int(depth), ctypes.byref(v)))

        npdt = {t.float32: np.float32, t.float64: np.float64, t.int32: np.int32, t.uint8: np.uint8}

        def arr(ptr, shape, dt):
            if not ptr:
                return None
            out = np.empty(shape, dtype=npdt[dt])
            if out.size:
                ctx.check(ctx.lib.svx_copy_to_host(ctx.h, ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(ptr), out.nbytes))
            return out
        d = {'size0': v.size0, 'size1': v.size1,
             'alignment_types': list(self.types) if depth == 0 else [(1, 1)],
             'n0': arr(v.n0, (v.k0, v.size0), t.float32), 'n1': arr(v.n1, (v.k1, v.size1), t.float32),
             'del_penalty': float(arr(v.del_penalty, (1,), t.float64)[0])}
        if v.alignments and v.n_align >= 0:
            d['alignments'] = rows_to_alignments(arr(v.alignments, (v.n_align, 4), t.int32))
        if v.knob_scores and v.n_knob > 0:
            d['knob_scores'] = arr(v.knob_scores, (v.n_knob,), t.float32)
        if v.costs_1to1:  # the coarsest level's dense stage (dp_utils.py:465-473)
            s0, s1 = v.size0, v.size1
            d['costs_1to1'] = arr(v.costs_1to1, (s0, s1), t.float32)
            diag = arr(v.x_y_tb_diag, (s0 + s1 + 1, s0 + 1), t.int32)
            xs, ys = np.meshgrid(np.arange(s0 + 1), np.arange(s1 + 1), indexing='ij')
            d['x_y_tb'] = np.ascontiguousarray(diag[xs + ys, xs])
        if v.v0_l0:
            dim = int(self.prm.d)
            d['v0_layer0'] = arr(v.v0_l0, (v.size0, dim), t.float32)
            d['v1_layer0'] = arr(v.v1_l0, (v.size1, dim), t.float32)
        if v.searchpath:
            A, B, T = v.path_len, v.band, v.n_types
            d['searchpath'] = [tuple(p) for p in arr(v.searchpath, (A, 2), t.int32).tolist()]
            d['b_offset'] = arr(v.b_offset, (A,), t.int32)
            if v.a_b_costs:
                d['a_b_costs'] = np.ascontiguousarray(arr(v.a_b_costs, (A, T, B), t.float32).transpose(1, 0, 2))
            d['a_b_csum'] = arr(v.a_b_csum, (A + 2, B), t.float64)
            if v.a_b_bp:
                bp = arr(v.a_b_bp, (A + 2, B), t.uint8).astype(np.int32)
                d['a_b_xp'] = np.where(bp == 255, -42, bp >> 4).astype(np.int32)
                d['a_b_yp'] = np.where(bp == 255, -42, bp & 15).astype(np.int32)
            else:
                d['a_b_xp'], d['a_b_yp'] = arr(v.a_b_xp, (A + 2, B), t.int32), arr(v.a_b_yp, (A + 2, B), t.int32)
            d['new_b_offset'] = arr(v.new_b_offset, (A + 2,), t.int32)
            if v.alignment_scores and v.n_align >= 0:
                d['alignment_scores'] = arr(v.alignment_scores, (v.n_align,), t.float64)
        return d

    def results(self):
        """-> list of (alignments, scores, del_penalties) per pair; raises on a device-side failure."""
        info, align, scores, pens, offs = self.raw_results()
        out = []
        for i in range(len(self.vecs)):
            cnt, o = int(info[i, 0]), int(offs[i])
            out.append((rows_to_alignments(align[o:o + cnt]), scores[o:o + cnt].copy(), pens[i, :self.levels[i]].copy()))
        return out


def align_batch(pairs, final_alignment_types, del_percentile_frac, width_over2, max_size_full_dp,
                costs_sample_size, num_samps_for_norm, rngs=None, norms=None, device=None, search="coarse_to_fine"):
    """vecalign() for a list of (vecs0, vecs1) pairs in one device pass.
    rngs: optional per-pair numpy RandomState objects (shard-invariant sampling); default = the
    global numpy stream consumed pair after pair, exactly like the reference's serial loop."""
    pb = PreparedBatch(pairs, final_alignment_types, del_percentile_frac, width_over2, max_size_full_dp,
                       costs_sample_size, num_samps_for_norm, rngs=rngs, norms=norms, device=device, search=search)
    pb.run()
    return pb.results()


def vecalign(vecs0, vecs1, final_alignment_types, del_percentile_frac, width_over2, max_size_full_dp,
             costs_sample_size, num_samps_for_norm, norms0=None, norms1=None, normalize_inputs_inplace=False,
             full_stack=False):
    """dp_utils.py:381-537.  Returns {0: {'final_alignments', 'alignment_scores', 'del_penalty',
    'size0', 'size1', 'alignment_types'}, d: {'del_penalty', 'size0', 'size1'} ...}; with full_stack=True every
    depth also carries the intermediates the reference keeps (n0, n1, searchpath, a_b_costs, b_offset, a_b_csum,
    a_b_xp, a_b_yp, new_b_offset, alignments: PreparedBatch.level_stack), copied back from the device."""
    pb = PreparedBatch([(vecs0, vecs1)], final_alignment_types, del_percentile_frac, width_over2, max_size_full_dp,
                       costs_sample_size, num_samps_for_norm, norms=[(norms0, norms1)])
    timed = logger.isEnabledFor(logging.INFO)
    if timed:
        pb.ctx.check(pb.ctx.lib.svx_set_profiling(pb.ctx.h, 1))
    try:
        pb.run()
        alignments, scores, pens = pb.results()[0]
        if timed:
            log_phase_times(pb.ctx)
    finally:
        if timed:
            pb.ctx.lib.svx_set_profiling(pb.ctx.h, 0)
    sizes = level_sizes(vecs0.shape[1], vecs1.shape[1], max_size_full_dp)
    stack = {}
    for depth, (s0, s1) in enumerate(sizes):
        stack[depth] = {'size0': s0, 'size1': s1, 'del_penalty': float(pens[depth]),
                        'alignment_types': list(final_alignment_types) if depth == 0 else [(1, 1)]}
        if full_stack:
            stack[depth].update(pb.level_stack(0, depth))
    stack[0]['final_alignments'] = alignments
    stack[0]['alignment_scores'] = scores
    if normalize_inputs_inplace:
        make_norm1(vecs0)
        make_norm1(vecs1)
    return stack


# the reference's phases (dp_utils.py:400,415-419,446,457,462-475,518-528) and the device stages that do their work;
# the fused pyramid pass of level 0 also produces that level's row norms and n0 / n1, so it is the "normalize" phase
PHASES = (('Downsample embeddings', ('pyr1', 'pyrN', 'pyr_aux')),
          ('Normalize embeddings', ('pyr0',)),
          ('Compute deletion penalties', ('knob_sort', 'knob_scores0', 'knob_scoresN', 'knob')),
          ('Full DP make features', ('dense_costs',)),
          ('Full DP', ('dense_dp',)),
          ('Upsample DP compute costs', ('path', 'band_costsN')),
          ('Upsample DP', ('band_dpN', 'traceback')),
          ('Final DP compute costs', ('path0', 'band_costs0', 'tiles')),
          ('Final DP', ('band_dp0', 'traceback0')))


def log_phase_times(ctx):
    """The reference's closing timing log (dp_utils.py:530-535: 'key took ....... 0.1234s' at INFO on logger
    'vecalign', phases above 5e-5 s only), from the device stage timers of the last call (svx_stage_ms; needs
    svx_set_profiling(1) around the call)."""
    runtimes = OrderedDict()
    for key, stages in PHASES:
        runtimes[key] = sum(max(0.0, float(ctx.lib.svx_stage_ms(ctx.h, s.encode()))) for s in stages) * 1e-3
    max_key_str_len = max(len(key) for key in runtimes)
    for key in runtimes:
        if runtimes[key] > 5e-5:
            logger.info(key + ' took ' + '.' * (max_key_str_len + 5 - len(key)) + ('%.4fs' % runtimes[key]).rjust(7))
    return runtimes


# ------------------------------------------------------------------------------------- per-function mirrors
def make_norm1(vecs0):
    """dp_utils.py:32-40, in place on a float32 numpy array [K, n, d] (or torch CUDA tensor)."""
    ctx = _ctx()
    x = _dev(ctx, vecs0)
    if x.dtype != ctx.torch.float32:
        raise ValueError("make_norm1 needs float32")
    ctx.check(ctx.lib.svx_make_norm1(ctx.h, _p(x), int(x.shape[0] * x.shape[1]), int(x.shape[2])))
    if not hasattr(vecs0, "data_ptr"):
        vecs0[...] = x.cpu().numpy()
    elif x.data_ptr() != vecs0.data_ptr():
        vecs0.copy_(x)


def downsample_vectors(vecs1):
    """dp_utils.py:362-378"""
    ctx = _ctx()
    t = ctx.torch
    x = _dev(ctx, vecs1)
    a, b, c = x.shape
    half = t.empty((a, b // 2, c), dtype=t.float32, device=ctx.tdev)
    ctx.check(ctx.lib.svx_downsample(ctx.h, _p(x), int(a), int(b), int(c), _p(half)))
    return half if hasattr(vecs1, "data_ptr") else half.cpu().numpy()


def compute_norms(vecs0, vecs1, num_samples, overlaps_to_use=None):
    """dp_utils.py:326-359 (draws from numpy's global stream like the reference)"""
    overlaps1, size1, dim = vecs1.shape
    overlaps0, size0, dim0 = vecs0.shape
    assert (dim == dim0)
    if overlaps_to_use is not None:
        if overlaps_to_use > overlaps1:
            raise Exception('Cannot use more overlaps than provided. You may want to re-run make_verlaps.py with a larger -n value')
    else:
        overlaps_to_use = overlaps1
    samps_per_overlap = ceil(num_samples / overlaps_to_use)
    if not (size1 and samps_per_overlap):
        return np.ones((overlaps0, size0)).astype(np.float32)
    idx = np.stack([np.random.choice(size1, size=samps_per_overlap, replace=True) for _ in range(overlaps_to_use)]).astype(np.int32)
    ctx = _ctx()
    t = ctx.torch
    norms0 = t.empty((overlaps0, size0), dtype=t.float32, device=ctx.tdev)
    v0, v1, di = _dev(ctx, vecs0), _dev(ctx, vecs1), _dev(ctx, idx)  # keep alive across the call
    ctx.check(ctx.lib.svx_compute_norms(ctx.h, _p(v0), overlaps0, size0, _p(v1), overlaps_to_use,
                                        size1, dim, _p(di), samps_per_overlap, _p(norms0)))
    return norms0.cpu().numpy()


class DeletionKnob(object):
    """dp_utils.py:43-79; the histogram/cdf/interp run on the device (svx_del_penalty)."""

    def __init__(self, samp, res_min, res_max):
        self.samp = np.ascontiguousarray(samp, dtype=np.float32)
        self.res_min, self.res_max = res_min, res_max

    def percentile_frac_to_del_penalty(self, knob_val):
        ctx = _ctx()
        t = ctx.torch
        out = t.zeros(1, dtype=t.float64, device=ctx.tdev)
        s = _dev(ctx, self.samp)
        ctx.check(ctx.lib.svx_del_penalty(ctx.h, _p(s), int(s.numel()), float(knob_val), _p(out)))
        return float(out.cpu().numpy()[0])


def make_del_knob(e_laser, f_laser, e_laser_norms, f_laser_norms, sample_size):
    """dp_utils.py:278-323"""
    from .dp_core import score_path
    e_size, f_size = e_laser.shape[0], f_laser.shape[0]
    if e_size > 0 and f_size > 0 and sample_size > 0:
        if e_size * f_size < sample_size:
            x_idxs = np.repeat(np.arange(e_size, dtype=np.int32), f_size)
            y_idxs = np.tile(np.arange(f_size, dtype=np.int32), e_size)
        else:
            x_idxs = np.random.choice(e_size, size=sample_size, replace=True).astype(np.int32)
            y_idxs = np.random.choice(f_size, size=sample_size, replace=True).astype(np.int32)
        random_scores = np.empty(len(x_idxs), dtype=np.float32)
        score_path(x_idxs, y_idxs, e_laser_norms, f_laser_norms, e_laser, f_laser, random_scores)
        return DeletionKnob(random_scores, 0, max(random_scores))
    return DeletionKnob(np.array([0.0, 0.5, 1.0]), 0, 1)


def dense_traceback(x_y_tb):
    """dp_utils.py:146-174"""
    ctx = _ctx()
    t = ctx.torch
    bp = _dev(ctx, np.ascontiguousarray(x_y_tb, dtype=np.int32))
    s0, s1 = bp.shape[0] - 1, bp.shape[1] - 1
    rows = t.zeros((max(1, s0 + s1), 4), dtype=t.int32, device=ctx.tdev)
    cnt = t.zeros(1, dtype=t.int32, device=ctx.tdev)
    ctx.check(ctx.lib.svx_dense_traceback(ctx.h, _p(bp), s0, s1, _p(rows), _p(cnt)))
    n = int(cnt.cpu().numpy()[0])
    if n < 0:
        raise Exception('got unknown value')
    return rows_to_alignments(rows.cpu().numpy()[:n])


def sparse_traceback(a_b_csum, a_b_xp, a_b_yp, b_offset, xsize, ysize):
    """dp_utils.py:105-143"""
    ctx = _ctx()
    t = ctx.torch
    csum = _dev(ctx, np.ascontiguousarray(a_b_csum, dtype=np.float64))
    cap = xsize + ysize + 2
    rows = t.zeros((cap, 4), dtype=t.int32, device=ctx.tdev)
    scores = t.zeros(cap, dtype=t.float64, device=ctx.tdev)
    cnt = t.zeros(1, dtype=t.int32, device=ctx.tdev)
    xp = _dev(ctx, np.ascontiguousarray(a_b_xp, dtype=np.int32))
    yp = _dev(ctx, np.ascontiguousarray(a_b_yp, dtype=np.int32))
    bo = _dev(ctx, np.ascontiguousarray(b_offset, dtype=np.int32))
    ctx.check(ctx.lib.svx_sparse_traceback(ctx.h, _p(csum), _p(xp), _p(yp), _p(bo),
                                           int(csum.shape[0]), int(csum.shape[1]), int(xsize), int(ysize),
                                           _p(rows), _p(scores), _p(cnt)))
    n = int(cnt.cpu().numpy()[0])
    if n < 0:
        raise Exception('traceback bug')
    return rows_to_alignments(rows.cpu().numpy()[:n]), scores.cpu().numpy()[:n].copy()


def make_search_path(alignments, size0=0, size1=0, upsample=False):
    """upsample_alignment + extend_alignments + alignment_to_search_path (dp_utils.py:261-275,
    228-258, 199-225) in one device call; with upsample=False it is alignment_to_search_path."""
    ctx = _ctx()
    t = ctx.torch
    rows = alignments_to_rows(alignments)
    if not upsample:
        size0 = max(size0, int(sum(r[1] for r in rows)))
        size1 = max(size1, int(sum(r[3] for r in rows)))
    path = t.zeros((size0 + size1 + 4, 2), dtype=t.int32, device=ctx.tdev)
    plen = t.zeros(1, dtype=t.int32, device=ctx.tdev)
    na = _dev(ctx, np.array([len(alignments)], dtype=np.int32))
    drows = _dev(ctx, rows)
    ctx.check(ctx.lib.svx_search_path(ctx.h, _p(drows), _p(na), 1 if upsample else 0, int(size0), int(size1),
                                      _p(path), _p(plen)))
    n = int(plen.cpu().numpy()[0])
    if n == -_lib.SVX_ERR_EXTEND:
        raise Exception('asked to extend alignments but already bigger than requested')
    if n < 0:
        raise Exception('search path failure %d' % -n)
    return [tuple(p) for p in path.cpu().numpy()[:n].tolist()]


def alignment_to_search_path(algn):
    """dp_utils.py:199-225"""
    return make_search_path(algn, upsample=False)


# ------------------------------------------------------------------------------------- Sakoe-Chiba mode
def align_band(vecs0, vecs1, final_alignment_types, del_percentile_frac, width_over2, costs_sample_size,
               num_samps_for_norm, rng=None):
    """Banded alignment around the straight diagonal (Sakoe-Chiba, BASELINE.json configs[3]; with
    width_over2 > max(N, M) the dense mode): the same recurrence as the refinement step of vecalign() --
    make_sparse_costs + sparse_dp + sparse_traceback (dp_core.pyx:165-404, dp_utils.py:105-143) with the final
    alignment types -- but the search path is the straight line from (0,0) to (N,M) (append_slant,
    dp_utils.py:177-196) with band half-width `width_over2` instead of an up-sampled coarse alignment.  Norms and
    the deletion penalty are those of depth 0 of vecalign() (same random draws, same order).  One svx_align_batch
    call (SVX_SEARCH_STRAIGHT): bands wider than 64 cells run as a wavefront of MFMA cost tiles + DP over all
    CUs.  Inputs of any storage type ([K, N, d] float32 / float16 / bfloat16).  Returns (alignments, scores)."""
    res = align_band_batch([(vecs0, vecs1)], final_alignment_types, del_percentile_frac, width_over2, costs_sample_size,
                           num_samps_for_norm, rngs=None if rng is None else [rng])
    return res[0][0], res[0][1]


def align_band_batch(pairs, final_alignment_types, del_percentile_frac, width_over2, costs_sample_size, num_samps_for_norm,
                     rngs=None, device=None):
    """align_band() for a list of pairs in one device pass -> [(alignments, scores, [del_penalty])]."""
    return align_batch(pairs, final_alignment_types, del_percentile_frac, width_over2, 1 << 30, costs_sample_size,
                       num_samps_for_norm, rngs=rngs, device=device, search="straight")

"""Whole-path parity on the GPU (svx_align_batch through svx.vecalign.dp_utils.vecalign):
  * golden vectors produced by the REAL reference (tests/golden/pipeline.npz) -- identical alignment
    spans, scores within 1e-4 (BASELINE.json north_star tolerance), deletion penalties within 5e-5;
  * the CPU oracle at the benchmark size (4096 x 4096, d = 1024, bf16);
  * size-independent properties: batch invariance, ragged batches, idempotence, seed handling.
"""
import os

import numpy as np
import pytest

from cases import PIPELINE_CASES, pipeline_inputs
from synth import alignment_types, make_pair, round_bf16

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "pipeline.npz")
SCORE_TOL = 1e-4


def to_dev(v, dt):
    import torch
    t = torch.from_numpy(v).cuda()
    return {"f32": t, "f16": t.half(), "bf16": t.bfloat16()}[dt]


def rows(al):
    out = np.zeros((len(al), 4), np.int32)
    for i, (x, y) in enumerate(al):
        out[i] = (x[0] if x else 0, len(x), y[0] if y else 0, len(y))
    return out


@pytest.mark.parametrize("name", list(PIPELINE_CASES))
def test_vecalign_vs_reference_golden(name):
    from svx.vecalign import dp_utils
    gold = np.load(GOLD)
    c = PIPELINE_CASES[name]
    v0, v1, types, W, kw = pipeline_inputs(c)
    np.random.seed(c["rng_seed"])
    stack = dp_utils.vecalign(to_dev(v0, kw["dtype"]), to_dev(v1, kw["dtype"]), types, c.get("frac", 0.2), W,
                              c.get("max_full", 300), c.get("sample", 20000), c.get("nsamp", 100), full_stack=True)
    # intermediates of the fused kernels (svx_debug_level) against what the reference kept in its stack
    assert np.abs(stack[0]['n0'] - gold[name + "/n0_l0"]).max() < 2e-6
    assert np.array(stack[0]['searchpath'], np.int64)[:, 1].sum() == gold[name + "/searchpath_sum"][0]
    pens = np.array([stack[d]['del_penalty'] for d in sorted(stack)])
    assert len(pens) == len(gold[name + "/del_pen"])
    assert np.abs(pens - gold[name + "/del_pen"]).max() < 5e-5
    got = rows(stack[0]['final_alignments'])
    assert got.shape == gold[name + "/align"].shape and np.array_equal(got[:, [1, 3]], gold[name + "/align"][:, [1, 3]])
    keep = (got[:, 1] > 0) & (got[:, 3] > 0)  # a deletion's start index is not defined by the reference (empty list)
    assert np.array_equal(got[keep], gold[name + "/align"][keep])
    assert np.abs(stack[0]['alignment_scores'] - gold[name + "/scores"]).max() < SCORE_TOL


def test_full_stack_vs_oracle(orc):
    """Every per-level intermediate the reference keeps (dp_utils.py:412-537), read back from the fused pipeline
    (svx_debug_level) and compared with the oracle's stack: integer arrays exact, costs 4e-6, float64 sums 1e-4."""
    from svx.vecalign import dp_utils
    v0, v1 = make_pair(1101, 1003, 4, 64, 4)
    types = alignment_types(5)
    np.random.seed(5)
    ref = orc.vecalign(v0.copy(), v1.copy(), types, 0.2, 7, 300, 20000, 100)
    np.random.seed(5)
    got = dp_utils.vecalign(v0, v1, types, 0.2, 7, 300, 20000, 100, full_stack=True)
    assert len(got) == len(ref) == 3
    for d in range(3):
        assert np.abs(got[d]['n0'] - ref[d]['n0']).max() < 4e-6 and np.abs(got[d]['n1'] - ref[d]['n1']).max() < 4e-6
        assert abs(got[d]['del_penalty'] - ref[d]['del_penalty']) < 5e-5
        if d < 2:
            assert got[d]['searchpath'] == [tuple(p) for p in ref[d]['searchpath']]
            assert np.array_equal(got[d]['b_offset'], ref[d]['b_offset']) and np.array_equal(got[d]['new_b_offset'], ref[d]['new_b_offset'])
            a, b = got[d]['a_b_costs'], ref[d]['a_b_costs']
            assert a.shape == b.shape and np.array_equal(np.isinf(a), np.isinf(b))
            assert np.abs(a[np.isfinite(a)] - b[np.isfinite(b)]).max() < 4e-6
            assert np.array_equal(got[d]['a_b_xp'], ref[d]['a_b_xp']) and np.array_equal(got[d]['a_b_yp'], ref[d]['a_b_yp'])
            fin = np.isfinite(ref[d]['a_b_csum'])
            assert np.array_equal(np.isfinite(got[d]['a_b_csum']), fin)
            assert np.abs(got[d]['a_b_csum'][fin] - ref[d]['a_b_csum'][fin]).max() < 1e-4 * (1 + ref[d]['a_b_csum'][fin].max())
        assert got[d]['alignments'] == ref[d]['final_alignments' if d == 0 else 'alignments']
        if d >= 1:  # the level's normalised layer 0 (downsample_vectors, dp_utils.py:362-378)
            assert np.abs(got[d]['v0_layer0'] - ref[d]['v0'][0]).max() < 2e-6
            assert np.abs(got[d]['v1_layer0'] - ref[d]['v1'][0]).max() < 2e-6
    # the coarsest level's dense stage (dp_utils.py:465-473): costs within the matrix-core tolerance; the back-pointer
    # array bit for bit what dense_dp makes of the SAME float32 costs and penalty, and (nearly) the reference's own
    top, rtop = got[2], ref[2]
    assert 'costs_1to1' not in got[0] and 'costs_1to1' not in got[1]
    assert top['costs_1to1'].shape == rtop['costs_1to1'].shape == (rtop['size0'], rtop['size1'])
    assert np.abs(top['costs_1to1'] - rtop['costs_1to1']).max() < 4e-6
    _, tb = orc.dense_dp(top['costs_1to1'], top['del_penalty'])
    assert top['x_y_tb'].dtype == np.int32 and np.array_equal(top['x_y_tb'], tb)
    assert (top['x_y_tb'] != rtop['x_y_tb']).mean() < 1e-3   # knife-edge nodes off the optimal path may differ
    assert orc.dense_traceback(top['x_y_tb']) == rtop['alignments']


def test_benchmark_size_vs_oracle(orc):
    """BASELINE configs[1]: 4096 x 4096, d = 1024, bf16, 4 overlap layers (10 types, band 14)."""
    import torch
    from svx.vecalign import dp_utils
    K, a = 4, 5
    types = alignment_types(a)
    W = int(np.ceil(K / 2.0)) + 5
    pairs, hosts = [], []
    for i in range(2):
        v0, v1 = make_pair(4096, 4096 - 37 * i, K, 1024, 100 + i, deletions=40 * i)
        v0, v1 = round_bf16(v0), round_bf16(v1)
        hosts.append((v0, v1))
        pairs.append((torch.from_numpy(v0).cuda().bfloat16(), torch.from_numpy(v1).cuda().bfloat16()))
    rngs = [np.random.RandomState(50 + i) for i in range(2)]
    res = dp_utils.align_batch(pairs, types, 0.2, W, 300, 20000, 100, rngs=rngs)
    for i, (v0, v1) in enumerate(hosts):
        ref = orc.vecalign(v0.copy(), v1.copy(), types, 0.2, W, 300, 20000, 100, rng=np.random.RandomState(50 + i))
        assert res[i][0] == ref[0]['final_alignments']
        assert np.abs(res[i][1] - ref[0]['alignment_scores']).max() < SCORE_TOL
        assert len(ref) == 5 and np.abs(res[i][2] - np.array([ref[d]['del_penalty'] for d in range(5)])).max() < 5e-5


def test_batch_invariance_and_idempotence():
    """A pair's result does not depend on which other pairs share its batch, nor on repetition."""
    from svx.vecalign import dp_utils
    types = alignment_types(5)
    shapes = [(700, 650), (90, 1300), (1500, 1400), (300, 280), (2049, 1999)]
    docs = [make_pair(n, m, 4, 64, 30 + i) for i, (n, m) in enumerate(shapes)]

    def rng(i):
        return np.random.RandomState(1000 + i)
    batch = dp_utils.align_batch(docs, types, 0.2, 7, 300, 20000, 100, rngs=[rng(i) for i in range(len(docs))])
    again = dp_utils.align_batch(docs, types, 0.2, 7, 300, 20000, 100, rngs=[rng(i) for i in range(len(docs))])
    for i in range(len(docs)):
        single = dp_utils.align_batch([docs[i]], types, 0.2, 7, 300, 20000, 100, rngs=[rng(i)])[0]
        assert single[0] == batch[i][0] == again[i][0]
        assert np.array_equal(single[1], batch[i][1]) and np.array_equal(again[i][1], batch[i][1])
        assert np.array_equal(single[2], batch[i][2])
        # every segment is covered exactly once, in order
        xs = [x for al in batch[i][0] for x in al[0]]
        ys = [y for al in batch[i][0] for y in al[1]]
        assert xs == list(range(shapes[i][0])) and ys == list(range(shapes[i][1]))


def test_pipeline_does_not_change_results():
    """svx_set_pipeline only changes WHEN kernels run: consecutive calls -- same batch again, a different batch, an
    odd number of pairs, a single pair -- give bit-identical outputs with the software pipeline on and off, and the
    stack of a pair in the held-back half reads back the same."""
    from svx import _lib
    from svx.vecalign import dp_utils
    types = alignment_types(4)
    docs = [make_pair(500 + 37 * i, 520 - 11 * i, 3, 64, 60 + i) for i in range(5)]
    more = [make_pair(900 - 50 * i, 700 + 90 * i, 3, 64, 80 + i) for i in range(4)]
    # enough pairs for the pipeline to cut its pyramid passes into slices (>= 32 per half), three levels each
    many = [make_pair(330 + (7 * i) % 90, 310 + (11 * i) % 70, 3, 32, 300 + i) for i in range(71)]
    ctx = _lib.context()

    def sequence():
        outs = []
        mk = lambda ds, s0: dp_utils.PreparedBatch(ds, types, 0.2, 7, 300, 20000, 100, rngs=[np.random.RandomState(s0 + i) for i in range(len(ds))])
        a, b, c, m = mk(docs, 0), mk(more, 100), mk(docs[:1], 200), mk(many, 400)
        for pb in (a, a, b, m, m, c, m, a):     # every run() overlaps the chain the one before it held back
            pb.run()
        a.flush()
        for pb in (a, b, c, m):
            outs.append(pb.results())
        outs.append(a.level_stack(4, 0)['a_b_csum'])   # a pair of the second (held-back) half
        return outs
    try:
        ctx.set_pipeline(False)
        plain = sequence()
        ctx.set_pipeline(True)
        piped = sequence()
    finally:
        ctx.set_pipeline(False)
    # svx_flush orders the held-back work in front of whatever follows ON THE STREAM: a device-side copy queued behind
    # the flush, without any host synchronisation in between, already sees the final rows of the held-back half
    import torch
    pb = dp_utils.PreparedBatch(many, types, 0.2, 7, 300, 20000, 100, rngs=[np.random.RandomState(400 + i) for i in range(len(many))])
    try:
        ctx.set_pipeline(True)
        pb.align.zero_()
        pb.info.zero_()
        pb.run()
        pb.flush()
        rows_after_flush, info_after_flush = pb.align.clone(), pb.info.clone()
        torch.cuda.synchronize()
    finally:
        ctx.set_pipeline(False)
    assert np.array_equal(info_after_flush.cpu().numpy(), pb.raw_results()[0])
    assert np.array_equal(rows_after_flush.cpu().numpy(), pb.raw_results()[1])
    assert [r[0] for r in pb.results()] == [r[0] for r in plain[3]]
    for o, q in zip(plain[:4], piped[:4]):
        assert len(o) == len(q)
        for x, y in zip(o, q):
            assert x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2])
    assert np.array_equal(plain[4], piped[4])


def test_phase_timing_log_like_the_reference():
    """The reference closes vecalign() with one INFO line per phase on logger 'vecalign' (dp_utils.py:530-535); here the
    same lines are written from the device stage timers when that logger is enabled for INFO, and cost nothing otherwise."""
    import logging
    import re
    from svx.vecalign import dp_utils
    v0, v1 = make_pair(1300, 1200, 4, 64, 9)
    types = alignment_types(5)
    lg = logging.getLogger('vecalign')
    seen = []

    class Grab(logging.Handler):
        def emit(self, record):
            seen.append(record.getMessage())
    h, old = Grab(), lg.level
    lg.addHandler(h)
    try:
        lg.setLevel(logging.INFO)
        np.random.seed(2)
        timed = dp_utils.vecalign(v0, v1, types, 0.2, 7, 300, 20000, 100)
        lines = [m for m in seen if ' took ' in m]
        lg.setLevel(logging.WARNING)
        del seen[:]
        np.random.seed(2)
        quiet = dp_utils.vecalign(v0, v1, types, 0.2, 7, 300, 20000, 100)
        assert not seen
    finally:
        lg.setLevel(old)
        lg.removeHandler(h)
    keys = [k for k, _ in dp_utils.PHASES]
    pat = re.compile(r'^(%s) took \.+ *\d+\.\d{4}s$' % '|'.join(re.escape(k) for k in keys))
    assert lines and all(pat.match(m) for m in lines), lines
    got = [pat.match(m).group(1) for m in lines]
    assert got == [k for k in keys if k in got]                      # the reference's order
    for must in ('Compute deletion penalties', 'Upsample DP', 'Final DP'):   # (like the reference, phases under 50 us are not logged)
        assert must in got, lines
    assert len({len(m) for m in lines}) == 1                         # right-aligned like the reference's
    assert timed[0]['final_alignments'] == quiet[0]['final_alignments'] and np.array_equal(timed[0]['alignment_scores'], quiet[0]['alignment_scores'])


def test_norm_overrides_and_global_stream(orc):
    """norms0/norms1 overrides (dp_utils.py:428-444) skip the corresponding random draws."""
    from svx.vecalign import dp_utils
    v0, v1 = make_pair(400, 380, 3, 64, 77)
    types = alignment_types(4)
    a, b = v0.copy(), v1.copy()
    orc.make_norm1(a)
    orc.make_norm1(b)
    n0 = orc.compute_norms(a, b, 100, np.random.RandomState(4))
    np.random.seed(8)
    ref = orc.vecalign(v0.copy(), v1.copy(), types, 0.2, 7, 300, 20000, 100, norms0=n0)
    np.random.seed(8)
    got = dp_utils.vecalign(v0, v1, types, 0.2, 7, 300, 20000, 100, norms0=n0)
    assert got[0]['final_alignments'] == ref[0]['final_alignments']
    assert np.abs(got[0]['alignment_scores'] - ref[0]['alignment_scores']).max() < SCORE_TOL
    with pytest.raises(Exception, match="norms0 wrong shape"):
        dp_utils.vecalign(v0, v1, types, 0.2, 7, 300, 20000, 100, norms0=n0[:, :-1])


def test_degenerate_documents(orc):
    from svx.vecalign import dp_utils
    types = alignment_types(3)
    for n, m in [(1, 1), (1, 9), (12, 1), (2, 3)]:
        v0, v1 = make_pair(n, m, 2, 64, 5)
        np.random.seed(1)
        ref = orc.vecalign(v0.copy(), v1.copy(), types, 0.2, 6, 300, 20000, 100)
        np.random.seed(1)
        got = dp_utils.vecalign(v0, v1, types, 0.2, 6, 300, 20000, 100)
        assert got[0]['final_alignments'] == ref[0]['final_alignments'], (n, m)
        assert np.abs(got[0]['alignment_scores'] - ref[0]['alignment_scores']).max() < SCORE_TOL
    with pytest.raises(Exception, match=r"4 x overlaps requrested \(via alignment_types\), but vecs0 only has 2"):
        dp_utils.vecalign(*make_pair(30, 30, 2, 64, 5), alignment_types(5), 0.2, 7, 300, 20000, 100)


def test_sakoe_chiba_band_vs_oracle(orc):
    """BASELINE configs[3] semantics at a size the oracle finishes in seconds: straight-diagonal search
    path, wide band (B = 96 > 64: generic DP kernel, six band-cell chunks per path chunk)."""
    from svx.vecalign import dp_utils
    N, M, K, W = 900, 840, 3, 48
    v0, v1 = make_pair(N, M, K, 64, 91, deletions=25)
    types = alignment_types(4)
    a, b = v0.copy(), v1.copy()
    orc.make_norm1(a)
    orc.make_norm1(b)
    rs = np.random.RandomState(17)
    n0, n1 = orc.compute_norms(a, b, 100, rs), orc.compute_norms(b, a, 100, rs)
    pen, _ = orc.make_del_penalty(a[0], b[0], n0[0], n1[0], 20000, 0.2, rs)
    path = orc.search_path([(list(range(N)), list(range(M)))], False, N, M)
    f, bo = orc.make_sparse_costs(a, b, n0, n1, path, types, W)
    al_o, sc_o = orc.sparse_traceback(*orc.sparse_dp(f, bo, types, pen, N, M), N, M)
    np.random.seed(17)
    al_g, sc_g = dp_utils.align_band(v0, v1, types, 0.2, W, 20000, 100)
    assert al_g == al_o
    assert np.abs(sc_g - sc_o).max() < SCORE_TOL


def test_dense_mode_vs_oracle(orc):
    """SURVEY 8(a) Mode B: the band is wider than the documents (W >= max(N, M) + 1), so every (x, y, type) cell
    of the lattice is evaluated -- the same kernels with B = 2W cells per diagonal.  Checked against the oracle's
    make_sparse_costs / sparse_dp / sparse_traceback on the same straight path, and against the coarse-to-fine
    result (the dense optimum is the global optimum; on this pair the banded search finds it too)."""
    from svx.vecalign import dp_utils
    N, M, K = 150, 137, 3
    W = max(N, M) + 1
    v0, v1 = make_pair(N, M, K, 64, 77, deletions=9)
    types = alignment_types(4)
    a, b = v0.copy(), v1.copy()
    orc.make_norm1(a)
    orc.make_norm1(b)
    rs = np.random.RandomState(23)
    n0, n1 = orc.compute_norms(a, b, 100, rs), orc.compute_norms(b, a, 100, rs)
    pen, _ = orc.make_del_penalty(a[0], b[0], n0[0], n1[0], 20000, 0.2, rs)
    path = orc.search_path([(list(range(N)), list(range(M)))], False, N, M)
    f, bo = orc.make_sparse_costs(a, b, n0, n1, path, types, W)
    al_o, sc_o = orc.sparse_traceback(*orc.sparse_dp(f, bo, types, pen, N, M), N, M)
    np.random.seed(23)
    al_g, sc_g = dp_utils.align_band(v0, v1, types, 0.2, W, 20000, 100)
    assert al_g == al_o
    assert np.abs(sc_g - sc_o).max() < SCORE_TOL
    np.random.seed(23)
    st = dp_utils.vecalign(v0, v1, types, 0.2, 7, 300, 20000, 100)
    assert st[0]['final_alignments'] == al_g


def _straight_oracle(orc, v0, v1, types, W, seed):
    """make_sparse_costs + sparse_dp + sparse_traceback on the straight path, depth-0 norms and penalty (oracle)."""
    N, M = v0.shape[1], v1.shape[1]
    a, b = v0.copy(), v1.copy()
    orc.make_norm1(a)
    orc.make_norm1(b)
    rs = np.random.RandomState(seed)
    n0, n1 = orc.compute_norms(a, b, 100, rs), orc.compute_norms(b, a, 100, rs)
    pen, _ = orc.make_del_penalty(a[0], b[0], n0[0], n1[0], 20000, 0.2, rs)
    path = orc.search_path([(list(range(N)), list(range(M)))], False, N, M)
    f, bo = orc.make_sparse_costs(a, b, n0, n1, path, types, W)
    return orc.sparse_traceback(*orc.sparse_dp(f, bo, types, pen, N, M), N, M)


@pytest.mark.parametrize("case", [
    dict(N=700, M=640, K=4, a=5, d=1024, W=40, dt="bf16", dels=20),     # tile sweep, bf16 d = 1024, 10 types
    dict(N=333, M=421, K=4, a=5, d=1024, W=422, dt="f16", dels=11),     # dense mode (band covers the lattice), fp16
    dict(N=260, M=250, K=5, a=6, d=64, W=50, dt="f32", dels=9),         # 15 types on 10 layers (second tile shape), fp32 rows
    dict(N=97, M=1000, K=3, a=4, d=64, W=45, dt="f32", dels=0),         # very uneven documents: steep straight path
    dict(N=500, M=480, K=4, a=5, d=96, W=33, dt="bf16", dels=15, zero=30),  # d not a multiple of 32 (tail slab), zero rows: exact ties
])
def test_straight_band_tiles_vs_oracle(orc, case):
    """SVX_SEARCH_STRAIGHT with bands wider than 64 cells (the tile sweep of csrc/svx_tiles.hip) against the oracle's
    make_sparse_costs / sparse_dp / sparse_traceback on the same straight path: identical spans, scores within 1e-4."""
    import torch
    from svx.vecalign import dp_utils
    c = case
    v0, v1 = make_pair(c["N"], c["M"], c["K"], c["d"], 400 + c["N"], deletions=c["dels"], zero_rows=c.get("zero", 0))
    if c["dt"] == "bf16":
        v0, v1 = round_bf16(v0), round_bf16(v1)
    elif c["dt"] == "f16":
        v0, v1 = v0.astype(np.float16).astype(np.float32), v1.astype(np.float16).astype(np.float32)
    types = alignment_types(c["a"])
    al_o, sc_o = _straight_oracle(orc, v0, v1, types, c["W"], 31)
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[c["dt"]]
    np.random.seed(31)
    al_g, sc_g = dp_utils.align_band(torch.from_numpy(v0).cuda().to(tdt), torch.from_numpy(v1).cuda().to(tdt), types, 0.2, c["W"], 20000, 100)
    assert al_g == al_o
    assert np.abs(sc_g - sc_o).max() < SCORE_TOL


def test_straight_band_batch_is_batch_invariant():
    """Several pairs of different sizes in one tile sweep (tickets run over pairs): every pair equals its own run."""
    from svx.vecalign import dp_utils
    types = alignment_types(5)
    shapes = [(300, 280), (90, 500), (640, 700), (33, 31)]
    docs = [make_pair(n, m, 4, 64, 900 + i, deletions=5) for i, (n, m) in enumerate(shapes)]
    mk = lambda i: np.random.RandomState(70 + i)
    batch = dp_utils.align_band_batch(docs, types, 0.2, 40, 20000, 100, rngs=[mk(i) for i in range(len(docs))])
    for i in range(len(docs)):
        single = dp_utils.align_band_batch([docs[i]], types, 0.2, 40, 20000, 100, rngs=[mk(i)])[0]
        assert single[0] == batch[i][0] and np.array_equal(single[1], batch[i][1])
        xs = [x for al in batch[i][0] for x in al[0]]
        ys = [y for al in batch[i][0] for y in al[1]]
        assert xs == list(range(shapes[i][0])) and ys == list(range(shapes[i][1]))


def test_sakoe_chiba_full_size_equals_coarse_to_fine():
    """BASELINE configs[3] at full size -- N = M = 32768, d = 1024, band 2048 around the straight diagonal
    (no CPU implementation finishes this in test time: 2.7e9 cost cells).  Size-independent property instead:
    both search modes evaluate the same recurrence, and for a pair whose true path stays inside both search
    regions the banded optimum and the coarse-to-fine optimum are the same alignment ("Modes")."""
    from svx.vecalign import dp_utils
    N = M = 32768
    v0, v1 = make_pair(N, M, 4, 1024, 5)
    types = alignment_types(5)
    np.random.seed(1)
    st = dp_utils.vecalign(v0, v1, types, 0.2, 7, 300, 20000, 100)
    np.random.seed(1)
    al_c, sc_c = dp_utils.align_band(v0, v1, types, 0.2, 1024, 20000, 100)
    al_a, sc_a = st[0]['final_alignments'], st[0]['alignment_scores']
    assert [x for al in al_c for x in al[0]] == list(range(N)) and [y for al in al_c for y in al[1]] == list(range(M))
    assert al_c == al_a
    assert np.abs(np.asarray(sc_c) - np.asarray(sc_a)).max() < SCORE_TOL


def test_ragged_batch_up_to_8192(orc):
    """BASELINE configs[2] shape: one batch of pairs with N, M drawn from [512, 8192]; every pair is
    checked for full monotone coverage, the largest one against the oracle."""
    import torch
    from svx.vecalign import dp_utils
    rs = np.random.RandomState(3)
    shapes = [(int(rs.randint(512, 8193)), int(rs.randint(512, 8193))) for _ in range(5)] + [(8192, 8000)]
    types = alignment_types(5)
    docs, hosts = [], []
    for i, (n, m) in enumerate(shapes):
        v0, v1 = make_pair(n, m, 4, 64, 200 + i)
        v0, v1 = round_bf16(v0), round_bf16(v1)
        hosts.append((v0, v1))
        docs.append((torch.from_numpy(v0).cuda().bfloat16(), torch.from_numpy(v1).cuda().bfloat16()))
    res = dp_utils.align_batch(docs, types, 0.2, 7, 300, 20000, 100, rngs=[np.random.RandomState(i) for i in range(len(docs))])
    for (n, m), r in zip(shapes, res):
        assert [x for al in r[0] for x in al[0]] == list(range(n)) and [y for al in r[0] for y in al[1]] == list(range(m))
    i = len(shapes) - 1
    ref = orc.vecalign(hosts[i][0].copy(), hosts[i][1].copy(), types, 0.2, 7, 300, 20000, 100, rng=np.random.RandomState(i))
    assert res[i][0] == ref[0]['final_alignments'] and np.abs(res[i][1] - ref[0]['alignment_scores']).max() < SCORE_TOL


def test_ragged_batch_c3_scale(orc):
    """BASELINE configs[2] at scale on one GPU: 256 document pairs, N, M ~ U{512..8192} i.i.d. (seed 1), d = 1024,
    bf16, 4 overlap layers (10 types, band 14), one svx_align_batch call.  Every pair: full monotone coverage
    of both documents (a size-independent property of any valid alignment); three pairs (the largest, the
    smallest and a random one) against the CPU oracle on the same rounded inputs and the same sampled indices."""
    import torch
    from synth import make_pair_device
    from svx.vecalign import dp_utils
    rs = np.random.RandomState(1)
    shapes = [(int(rs.randint(512, 8193)), int(rs.randint(512, 8193))) for _ in range(256)]
    types = alignment_types(5)
    dev = torch.device("cuda", torch.cuda.current_device())
    docs = [make_pair_device(n, m, 4, 1024, 7000 + i, dev, torch.bfloat16) for i, (n, m) in enumerate(shapes)]
    res = dp_utils.align_batch(docs, types, 0.2, 7, 300, 20000, 100, rngs=[np.random.RandomState(900 + i) for i in range(len(docs))])
    for (n, m), r in zip(shapes, res):
        assert [x for al in r[0] for x in al[0]] == list(range(n)) and [y for al in r[0] for y in al[1]] == list(range(m))
        assert np.isfinite(r[1]).all() and (r[1] >= 0).all()
    sizes = [n + m for n, m in shapes]
    picks = {int(np.argmax(sizes)), int(np.argmin(sizes)), int(rs.randint(0, 256))}
    for i in sorted(picks):
        h0, h1 = docs[i][0].float().cpu().numpy(), docs[i][1].float().cpu().numpy()
        ref = orc.vecalign(h0, h1, types, 0.2, 7, 300, 20000, 100, rng=np.random.RandomState(900 + i))
        assert res[i][0] == ref[0]['final_alignments'], shapes[i]
        assert np.abs(res[i][1] - ref[0]['alignment_scores']).max() < SCORE_TOL


def test_randomised_sweep_small():
    """A short run of tests/fuzz_gpu_vs_oracle.py (random sizes, layers, types, band widths, thresholds, storage
    types, deletions, zero rows; ragged batches): identical spans and scores within 1e-4 in every case that is not an
    exact tie or a percentile knife-edge of the reference itself (see that file)."""
    from fuzz_gpu_vs_oracle import run_sweep
    bad, ties, edges = run_sweep(160, 11, verbose=False)
    assert bad == 0
    assert ties + edges <= 3


def test_document_longer_than_the_sort_histogram(orc):
    """More than ~38 000 segments per side: the sampled-score pass cannot sort by source row in LDS and scores the
    samples in drawn order instead (k_knob_scores_unsorted); everything else is size-independent.  Against the oracle."""
    from svx.vecalign import dp_utils
    N, M, K, d = 40000, 39000, 2, 32
    v0, v1 = make_pair(N, M, K, d, 123, deletions=40)
    types = alignment_types(3)
    np.random.seed(9)
    ref = orc.vecalign(v0.copy(), v1.copy(), types, 0.2, 6, 300, 20000, 100)
    np.random.seed(9)
    got = dp_utils.vecalign(v0, v1, types, 0.2, 6, 300, 20000, 100)
    assert got[0]['final_alignments'] == ref[0]['final_alignments']
    assert np.abs(got[0]['alignment_scores'] - ref[0]['alignment_scores']).max() < SCORE_TOL


def test_mixed_depth_batch_with_long_search_paths(orc):
    """One svx_align_batch call over pairs of very different pyramid depths: a pair whose level-1 alignment has more
    rows (~11 000) than the search-path kernel keeps in LDS (9 597: that pair's up-sampling runs on the kernel's
    one-thread branch), a pair without any pyramid level (L = 0: dense DP at level 0) and two in between.  Every pair
    against the oracle on its own random stream."""
    import torch
    from svx.vecalign import dp_utils
    shapes = [(11500, 10800), (120, 141), (2300, 1700), (640, 9000)]
    K, d = 2, 32
    types = alignment_types(3)
    hosts = [make_pair(n, m, K, d, 50 + i, deletions=8 if min(n, m) > 200 else 0) for i, (n, m) in enumerate(shapes)]
    devs = [(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()) for a, b in hosts]
    res = dp_utils.align_batch(devs, types, 0.2, 6, 300, 20000, 100, rngs=[np.random.RandomState(70 + i) for i in range(len(shapes))])
    for i, (a, b) in enumerate(hosts):
        ref = orc.vecalign(a.copy(), b.copy(), types, 0.2, 6, 300, 20000, 100, rng=np.random.RandomState(70 + i))
        assert len(ref) == len(dp_utils.level_sizes(a.shape[1], b.shape[1], 300))   # (depths 0..L on both sides)
        assert res[i][0] == ref[0]['final_alignments'], shapes[i]
        assert np.abs(res[i][1] - ref[0]['alignment_scores']).max() < SCORE_TOL


def test_stage_timers_accumulate_without_synchronising():
    """svx_set_profiling(ctx, 2): stage times and launch counts are totals over the calls since, read back on demand
    (bench.py's timed loop queues its steps behind one another and reads the HIP events once at the end)."""
    from svx.vecalign import dp_utils
    pb = dp_utils.PreparedBatch([make_pair(700, 650, 3, 64, 5)], alignment_types(4), 0.2, 7, 300, 20000, 100,
                                rngs=[np.random.RandomState(1)])
    ctx, lib = pb.ctx, pb.ctx.lib
    pb.run()
    ctx.sync()
    lib.svx_set_profiling(ctx.h, 1)
    pb.run()
    one = {s: (lib.svx_stage_ms(ctx.h, s.encode()), lib.svx_stage_launches(ctx.h, s.encode())) for s in ("pyr0", "band_dp0", "traceback", "total")}
    lib.svx_set_profiling(ctx.h, 2)
    for _ in range(3):
        pb.run()
    tot = {s: (lib.svx_stage_ms(ctx.h, s.encode()), lib.svx_stage_launches(ctx.h, s.encode())) for s in one}
    lib.svx_set_profiling(ctx.h, 0)
    for s in one:
        assert tot[s][1] == 3 * one[s][1] and one[s][1] > 0
        assert 1.5 * one[s][0] < tot[s][0] < 6 * one[s][0]
    assert lib.svx_stage_ms(ctx.h, b"no such stage") == -1.0


def test_context_follows_current_device():
    """A rank that called torch.cuda.set_device(LOCAL_RANK) must compute on that GPU (seg_align.align under
    torchrun): contexts default to torch's current device.  Needs two visible devices for the non-trivial half."""
    import torch
    from svx import _lib
    from svx.vecalign import dp_utils
    cur = torch.cuda.current_device()
    assert _lib.context().device == cur
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device: only the default-device half can be checked")
    try:
        torch.cuda.set_device(1)
        pb = dp_utils.PreparedBatch([make_pair(300, 280, 3, 64, 5)], alignment_types(4), 0.2, 7, 300, 20000, 100,
                                    rngs=[np.random.RandomState(1)])
        assert pb.ctx.device == 1 and pb.align.device.index == 1 and pb.vecs[0][0].device.index == 1
        pb.run()
        assert len(pb.results()[0][0]) > 0
    finally:
        torch.cuda.set_device(cur)


@pytest.mark.parametrize("m2o", [8, 50])
def test_many_to_one_vs_oracle(orc, m2o):
    """--many_to_one M (vecalign.py:165-171: types (1,1) ... (M,1), M overlap layers on the source side, one on the
    target side; the CLI default is 50): the band-cost kernel stages the layers a pass of types needs, so the number
    of layers is not limited by LDS.  M = 50 also exercises unpacked (int32) back-pointers: steps above 15."""
    from svx.vecalign import dp_utils
    from svx.vecalign.vecalign import resolve_search_params
    types, sk, tk, W = resolve_search_params(10, m2o, 5)
    assert (sk, tk) == (m2o, 1) and len(types) == m2o
    v0, v1 = make_pair(420, 130, m2o, 64, 77)
    v1 = np.ascontiguousarray(v1[:1])
    np.random.seed(13)
    ref = orc.vecalign(v0.copy(), v1.copy(), types, 0.2, W, 300, 20000, 100)
    np.random.seed(13)
    got = dp_utils.vecalign(v0, v1, types, 0.2, W, 300, 20000, 100)
    assert got[0]['final_alignments'] == ref[0]['final_alignments']
    assert np.abs(got[0]['alignment_scores'] - ref[0]['alignment_scores']).max() < SCORE_TOL
    assert max(len(x) for x, _ in got[0]['final_alignments']) > 1


def test_straight_narrow_band_vs_oracle(orc):
    """SVX_SEARCH_STRAIGHT with a band of <= 64 cells takes the band-cost / fast-DP kernels of the coarse-to-fine path
    (no tile sweep): against the oracle on the same straight path."""
    from svx.vecalign import dp_utils
    N, M, K, W = 800, 760, 4, 20
    v0, v1 = make_pair(N, M, K, 64, 321, deletions=12)
    types = alignment_types(5)
    al_o, sc_o = _straight_oracle(orc, v0, v1, types, W, 44)
    np.random.seed(44)
    al_g, sc_g = dp_utils.align_band(v0, v1, types, 0.2, W, 20000, 100)
    assert al_g == al_o
    assert np.abs(sc_g - sc_o).max() < SCORE_TOL


@pytest.mark.parametrize("m2o", [None, 20])
def test_band_too_wide_for_the_traceback_window_vs_oracle(orc, m2o):
    """A band of 400 cells per diagonal leaves no room for a traceback window in LDS (tb_chunk() == 0): the walk then
    reads band offsets and back-pointers from global memory -- packed bytes for the 10-type shape, int32 pairs when the
    steps do not fit four bits (--many_to_one 20).  Same spans and scores as the oracle."""
    from svx.vecalign import dp_utils
    from svx.vecalign.vecalign import resolve_search_params
    W = 200
    if m2o is None:
        types = alignment_types(5)
        v0, v1 = make_pair(310, 280, 4, 64, 91, deletions=7)
    else:
        types, sk, tk, _ = resolve_search_params(10, m2o, 5)
        v0, v1 = make_pair(330, 90, m2o, 64, 92)
        v1 = np.ascontiguousarray(v1[:1])
    np.random.seed(17)
    ref = orc.vecalign(v0.copy(), v1.copy(), types, 0.2, W, 300, 20000, 100)
    np.random.seed(17)
    got = dp_utils.vecalign(v0, v1, types, 0.2, W, 300, 20000, 100)
    assert got[0]['final_alignments'] == ref[0]['final_alignments']
    assert np.abs(got[0]['alignment_scores'] - ref[0]['alignment_scores']).max() < SCORE_TOL

"""The CPU oracle against the golden vectors captured from the REAL reference
(tests/golden/*.npz, generator: tests/golden/make_golden.py).  Integer outputs, float32 costs and
float64 DP sums are bit-exact; quantities behind np.matmul (norms, and what follows from them in the
whole-pipeline cases) get 1e-6 because BLAS kernels may differ between hosts."""
import os

import numpy as np
import pytest

from cases import OPS_CASE, PIPELINE_CASES, pipeline_inputs
from synth import alignment_types, make_pair

GD = os.path.join(os.path.dirname(__file__), "golden")


def rows(al):
    out = np.zeros((len(al), 4), np.int32)
    for i, (x, y) in enumerate(al):
        out[i] = (x[0] if x else 0, len(x), y[0] if y else 0, len(y))
    return out


def test_ops_golden(orc):
    g = np.load(os.path.join(GD, "ops.npz"))
    c = OPS_CASE
    v0, v1 = make_pair(c["N"], c["M"], c["K"], c["d"], c["seed"])
    a, b = v0.copy(), v1.copy()
    orc.make_norm1(a)
    orc.make_norm1(b)
    assert np.array_equal(a[:, 0, :], g["norm_a_row0"]) and np.array_equal(b[:, -1, :], g["norm_b_last"])
    assert np.array_equal(orc.downsample_vectors(a), g["half"])
    rs = np.random.RandomState(c["norm_seed"])
    n0, n1 = orc.compute_norms(a, b, 100, rs), orc.compute_norms(b, a, 100, rs)
    assert np.abs(n0 - g["n0"]).max() < 1e-6 and np.abs(n1 - g["n1"]).max() < 1e-6
    n0, n1 = g["n0"], g["n1"]
    assert np.array_equal(orc.make_dense_costs(a, b, n0, n1, 1, 2), g["dense_costs_1_2"])
    costs = orc.make_dense_costs(a, b, n0, n1)
    assert np.array_equal(costs, g["dense_costs"])
    csum, bp = orc.dense_dp(costs, c["dense_pen"])
    assert np.array_equal(csum, g["dense_csum"]) and np.array_equal(bp, g["dense_bp"].astype(np.int32))
    al = orc.dense_traceback(bp)
    assert np.array_equal(rows(al), g["dense_align"])
    path = orc.search_path(al, False, c["N"], c["M"])
    assert np.array_equal(np.array(path, np.int32), g["path"])
    assert np.array_equal(np.array(orc.search_path(al, True, 2 * c["N"] + 1, 2 * c["M"]), np.int32), g["path_up"])
    types = alignment_types(c["a"])
    f, bo = orc.make_sparse_costs(a, b, n0, n1, path, types, c["W"])
    assert np.array_equal(f, g["sparse_costs"]) and np.array_equal(bo, g["b_offset"])
    r = orc.sparse_dp(f, bo, types, c["sparse_pen"], c["N"], c["M"])
    assert np.array_equal(r[0], g["sparse_csum"]) and np.array_equal(r[1], g["sparse_xp"].astype(np.int32))
    assert np.array_equal(r[2], g["sparse_yp"].astype(np.int32)) and np.array_equal(r[3], g["b_offset_out"])
    al2, sc = orc.sparse_traceback(*r, c["N"], c["M"])
    assert np.array_equal(rows(al2), g["sparse_align"]) and np.array_equal(sc, g["sparse_scores"])
    xs = np.random.RandomState(1).randint(0, c["N"], 5000).astype(np.int32)
    ys = np.random.RandomState(2).randint(0, c["M"], 5000).astype(np.int32)
    out = np.empty(5000, np.float32)
    orc.score_path(xs, ys, n0[0], n1[0], a[0], b[0], out)
    assert np.array_equal(out, g["score_path"])
    pens = np.array([orc.del_penalty_from_scores(out, 0, max(out), f_) for f_ in (0.05, 0.2, 0.25, 0.5, 0.9)])
    assert np.array_equal(pens, g["del_pen"])
    assert np.array_equal(orc.dense_dp(np.zeros((2, 3), np.float32), 0.0)[1], g["dense_bp_zero_2x3"].astype(np.int32))
    e = orc.sparse_dp(np.zeros((0,) + f.shape[1:], np.float32), bo, [], 0.5, c["N"], c["M"])
    assert np.array_equal(e[0], g["empty_types_csum"]) and np.array_equal(e[1], g["empty_types_xp"].astype(np.int32))
    with pytest.raises(Exception, match="2 x overlaps requrested"):
        orc.make_sparse_costs(a[:1], b[:1], n0[:1], n1[:1], path, [(2, 1)], 3)


@pytest.mark.parametrize("name", list(PIPELINE_CASES))
def test_pipeline_golden(orc, name):
    g = np.load(os.path.join(GD, "pipeline.npz"))
    c = PIPELINE_CASES[name]
    v0, v1, types, W, _ = pipeline_inputs(c)
    np.random.seed(c["rng_seed"])
    st = orc.vecalign(v0.copy(), v1.copy(), types, c.get("frac", 0.2), W, c.get("max_full", 300), c.get("sample", 20000),
                      c.get("nsamp", 100))
    assert np.array_equal(rows(st[0]['final_alignments']), g[name + "/align"])
    assert np.abs(st[0]['alignment_scores'] - g[name + "/scores"]).max() < 1e-6
    assert np.abs(np.array([st[d]['del_penalty'] for d in sorted(st)]) - g[name + "/del_pen"]).max() < 1e-6
    assert np.abs(st[0]['n0'] - g[name + "/n0_l0"]).max() < 1e-6
    assert np.array(st[0]['searchpath'], np.int64)[:, 1].sum() == g[name + "/searchpath_sum"][0]

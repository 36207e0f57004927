"""First-light GPU checks: every C-ABI op and the fused pipeline against the CPU oracle."""
import numpy as np
import pytest

from synth import make_pair, alignment_types

pytestmark = pytest.mark.gpu


def test_ops_vs_oracle(orc):
    from svx.vecalign import dp_core, dp_utils
    K, N, M, d = 3, 70, 61, 64
    v0, v1 = make_pair(N, M, K, d, 1)
    a, b = v0.copy(), v1.copy()
    orc.make_norm1(a); orc.make_norm1(b)
    g0 = v0.copy(); dp_utils.make_norm1(g0)
    assert np.abs(g0 - a).max() < 1e-6
    ha, hg = orc.downsample_vectors(a), dp_utils.downsample_vectors(a)
    assert np.abs(ha - hg).max() < 2e-6
    rs = np.random.RandomState(5)
    n0 = orc.compute_norms(a, b, 100, rs)
    np.random.seed(5)
    g = dp_utils.compute_norms(a, b, 100)
    assert np.abs(n0 - g).max() < 2e-6
    n1 = orc.compute_norms(b, a, 100, rs)
    c_o = orc.make_dense_costs(a, b, n0, n1, 1, 2)
    c_g = dp_core.make_dense_costs(a, b, n0, n1, 1, 2)
    assert np.abs(c_o - c_g).max() < 1e-5
    cs_o, bp_o = orc.dense_dp(c_o, 0.3)
    cs_g, bp_g = dp_core.dense_dp(c_o, 0.3)
    assert np.array_equal(bp_o, bp_g) and np.array_equal(cs_o, cs_g)
    al_o = orc.dense_traceback(bp_o)
    assert dp_utils.dense_traceback(bp_o) == al_o
    path = orc.search_path(al_o, False, N, M)
    assert dp_utils.alignment_to_search_path(al_o) == path
    types = alignment_types(4)
    f_o, bo_o = orc.make_sparse_costs(a, b, n0, n1, path, types, 6)
    f_g, bo_g = dp_core.make_sparse_costs(a, b, n0, n1, path, types, 6)
    assert np.array_equal(bo_o, bo_g)
    assert np.array_equal(np.isinf(f_o), np.isinf(f_g))
    fin = np.isfinite(f_o)
    assert np.abs(f_o[fin] - f_g[fin]).max() < 1e-5
    r_o = orc.sparse_dp(f_o, bo_o, types, 0.25, N, M)
    r_g = dp_core.sparse_dp(f_o, bo_o, types, 0.25, N, M)
    for x, y in zip(r_o, r_g):
        assert np.array_equal(x, y)
    al2_o, sc_o = orc.sparse_traceback(*r_o, N, M)
    al2_g, sc_g = dp_utils.sparse_traceback(*r_o, N, M)
    assert al2_o == al2_g and np.array_equal(sc_o, sc_g)
    xs = np.random.RandomState(1).randint(0, N, 5000).astype(np.int32)
    ys = np.random.RandomState(2).randint(0, M, 5000).astype(np.int32)
    so = np.empty(5000, np.float32); sg = np.empty(5000, np.float32)
    orc.score_path(xs, ys, n0[0], n1[0], a[0], b[0], so)
    dp_core.score_path(xs, ys, n0[0], n1[0], a[0], b[0], sg)
    assert np.abs(so - sg).max() < 1e-5
    pen_o = orc.del_penalty_from_scores(so, 0, max(so), 0.2)
    pen_g = dp_utils.DeletionKnob(so, 0, max(so)).percentile_frac_to_del_penalty(0.2)
    assert pen_o == pen_g


@pytest.mark.parametrize("N,M,K,d,a", [(40, 37, 3, 64, 4), (237, 217, 5, 64, 6), (700, 650, 4, 64, 5), (1100, 1000, 4, 32, 5)])
def test_pipeline_vs_oracle(orc, N, M, K, d, a):
    from svx.vecalign import dp_utils
    v0, v1 = make_pair(N, M, K, d, 1)
    types = alignment_types(a)
    W = int(np.ceil((a - 1) / 2.0)) + 5
    np.random.seed(3)
    so = orc.vecalign(v0.copy(), v1.copy(), types, 0.2, W, 300, 20000, 100)
    np.random.seed(3)
    sg = dp_utils.vecalign(v0, v1, types, 0.2, W, 300, 20000, 100)
    for dep in so:
        assert abs(so[dep]['del_penalty'] - sg[dep]['del_penalty']) < 5e-5, dep
    assert so[0]['final_alignments'] == sg[0]['final_alignments']
    assert np.abs(so[0]['alignment_scores'] - sg[0]['alignment_scores']).max() < 1e-4

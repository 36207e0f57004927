"""Pins the oracle: every intermediate of the oracle's vecalign() equals the REAL reference's,
bit for bit, on seeded inputs.  Runs only where /root/reference exists (the build container)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import ref_loader  # noqa: E402
from synth import alignment_types, make_pair  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference not present (GPU box)")

KEYS = ['v0', 'v1', 'n0', 'n1', 'del_penalty', 'costs_1to1', 'x_y_tb', 'alignments', 'searchpath', 'a_b_costs', 'b_offset',
        'a_b_csum', 'a_b_xp', 'a_b_yp', 'new_b_offset', 'final_alignments', 'alignment_scores']


@pytest.mark.parametrize("N,M,K,d,a,kw", [
    (40, 37, 3, 64, 4, {}), (237, 217, 5, 64, 6, {}), (700, 650, 4, 64, 5, {}), (1101, 1003, 4, 32, 5, {}),
    (300, 310, 1, 64, 2, {}), (500, 480, 4, 64, 5, dict(zero_rows=25)), (512, 512, 4, 1024, 5, {}),
    (420, 400, 3, 64, 4, dict(max_full=50)), (150, 140, 3, 64, 4, dict(max_full=10 ** 6, W=151)),
])
def test_stack_bit_exact(orc, N, M, K, d, a, kw):
    ref = ref_loader.load()
    v0, v1 = make_pair(N, M, K, d, 1, zero_rows=kw.get("zero_rows", 0))
    types = ref.vecalign.make_alignment_types(a)
    assert types == alignment_types(a)
    W = kw.get("W", int(np.ceil((a - 1) / 2.0)) + 5)
    mf = kw.get("max_full", 300)
    np.random.seed(3)
    sr = ref.dp_utils.vecalign(v0.copy(), v1.copy(), types, 0.2, W, mf, 20000, 100)
    np.random.seed(3)
    so = orc.vecalign(v0.copy(), v1.copy(), types, 0.2, W, mf, 20000, 100)
    assert sorted(sr) == sorted(so)
    for dep in sr:
        for key in KEYS:
            if key not in sr[dep]:
                continue
            x, y = sr[dep][key], so[dep][key]
            if key in ('alignments', 'final_alignments'):
                assert [(list(p), list(q)) for p, q in x] == [(list(p), list(q)) for p, q in y], (dep, key)
            elif key == 'searchpath':
                assert [tuple(p) for p in x] == [tuple(p) for p in y], (dep, key)
            else:
                assert np.array_equal(np.asarray(x), np.asarray(y)), (dep, key)


def test_rng_draw_order_matches_reference():
    """svx.vecalign.dp_utils.draw_indices consumes numpy's global stream exactly like the reference."""
    from svx.vecalign import dp_utils
    ref = ref_loader.load()
    calls = []
    real = np.random.choice

    def spy(a, size=None, replace=True, p=None):
        out = real(a, size=size, replace=replace, p=p)
        calls.append(np.asarray(out).astype(np.int32))
        return out
    v0, v1 = make_pair(1100, 1000, 4, 32, 2)
    np.random.seed(9)
    np.random.choice = spy
    try:
        ref.dp_utils.vecalign(v0.copy(), v1.copy(), alignment_types(5), 0.2, 7, 300, 20000, 100)
    finally:
        np.random.choice = real
    np.random.seed(9)
    ni, ki = dp_utils.draw_indices(1100, 1000, 4, 4, 300, 20000, 100)
    flat = np.concatenate(calls)
    assert np.array_equal(flat, np.concatenate([ni, ki]))

"""Seeded test cases shared by the golden-vector generator (tests/golden/make_golden.py, which
runs the REAL reference) and by the parity tests (which run the oracle and the HIP path)."""
import numpy as np

from synth import alignment_types, make_pair, round_bf16

OPS_CASE = dict(N=70, M=61, K=3, d=64, seed=1, norm_seed=5, a=4, W=6, dense_pen=0.3, sparse_pen=0.25)

# name -> parameters.  a = alignment_max_size (K = a - 1 overlap layers per side).
PIPELINE_CASES = {
    "tiny_L0":        dict(N=40, M=37, a=4, d=64, seed=1, rng_seed=3),              # n*m < 20000: full enumeration of knob samples
    "example_like":   dict(N=237, M=217, a=6, d=64, seed=2, rng_seed=3),            # L = 0, 15 types, band 16 (shape of the shipped example)
    "L1":             dict(N=700, M=650, a=5, d=64, seed=3, rng_seed=4),
    "odd_L2":         dict(N=1101, M=1003, a=5, d=32, seed=4, rng_seed=5),          # odd sizes: row drop + extend quirk at two levels
    "ragged":         dict(N=333, M=1200, a=5, d=64, seed=5, rng_seed=6),           # N != M
    "enum_branch":    dict(N=90, M=80, a=3, d=64, seed=6, rng_seed=7),
    "k1":             dict(N=300, M=310, a=2, d=64, seed=7, rng_seed=8),            # only 1-1 (+ deletions), K = 1
    "zero_rows":      dict(N=500, M=480, a=5, d=64, seed=8, rng_seed=9, zero_rows=25),   # PAD / ignored candidates: exact cost ties
    "deletions":      dict(N=600, M=560, a=5, d=64, seed=9, rng_seed=10, deletions=30),
    "fp16_d1024":     dict(N=512, M=512, a=5, d=1024, seed=10, rng_seed=11, dtype="f16"),
    "f32_d1024":      dict(N=512, M=512, a=6, d=1024, seed=15, rng_seed=16),                # BASELINE configs[0]: 512 x 512, d = 1024, float32 storage, -a 6
    "bf16_d1024":     dict(N=640, M=600, a=5, d=1024, seed=11, rng_seed=12, dtype="bf16"),
    "frac_quarter":   dict(N=420, M=400, a=4, d=64, seed=12, rng_seed=13, frac=0.25),   # knot 7/28 exactly
    "small_full_dp":  dict(N=420, M=400, a=4, d=64, seed=13, rng_seed=14, max_full=50),  # deeper pyramid (L = 4)
    "fullband":       dict(N=150, M=140, a=4, d=64, seed=14, rng_seed=15, max_full=10 ** 6, W=151),  # Mode B: band covers the matrix
}

EXAMPLE_TRIM = dict(n_src=48, n_tgt=44)


def pipeline_inputs(c):
    """-> (vecs0 float32, vecs1 float32, types, width_over2, extras).  For fp16/bf16 cases the float32
    arrays hold values that are exactly representable in that type (round once)."""
    K = c["a"] - 1
    v0, v1 = make_pair(c["N"], c["M"], K, c["d"], c["seed"], zero_rows=c.get("zero_rows", 0), deletions=c.get("deletions", 0))
    dt = c.get("dtype", "f32")
    if dt == "f16":
        v0, v1 = v0.astype(np.float16).astype(np.float32), v1.astype(np.float16).astype(np.float32)
    elif dt == "bf16":
        v0, v1 = round_bf16(v0), round_bf16(v1)
    types = alignment_types(c["a"])
    W = c.get("W", int(np.ceil(K / 2.0)) + 5)
    return v0, v1, types, W, dict(dtype=dt)

"""Randomised check of the margin-scoring kernels against the numpy oracle (not collected by pytest):
    python tests/fuzz_margin_gpu.py --cases 200 --seed 0
Random query counts, database sizes, dimensions (multiples of 32 up to 1024), k (1..64), storage (fp16 / bf16),
query types (fp32 / fp16) and margins.  Tolerance: the rounding-flip bound of tests/test_gpu_margin.py (one element
of a normalised query may land on the other side of a storage rounding boundary) scales with the size of an
element, i.e. with 1/d: 8e-6 * 1024/d for fp16 storage, ten times that for bf16.  Rows whose squared norm lies on a
power of four are reported as knife-edges (see the comment in run())."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "speech-vecalign_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def run(cases, seed):
    import oracle
    from svx.postprocess.flat_index import FlatIndex
    from svx.postprocess.score_align import compute_sim_with_nonflat_idx
    rs = np.random.RandomState(seed)
    bad = edges = 0
    t0 = time.time()
    for c in range(cases):
        d = 32 * int(rs.randint(1, 33))
        k = int(rs.choice([1, 2, 7, 16, 16, 16, 17, 32, 64]))
        N = int(rs.randint(k, 3000))
        n = int(rs.randint(1, 700)) if rs.rand() < 0.97 else int(rs.randint(16384, 20000))
        storage = str(rs.choice(["fp16", "bf16"]))
        margin = str(rs.choice(["ratio", "distance"]))
        qdt = np.float16 if rs.rand() < 0.3 else np.float32
        centres = rs.randn(8, d).astype(np.float32)
        mk = lambda r: (centres[rs.randint(0, 8, r)] + 0.6 * rs.randn(r, d).astype(np.float32)).astype(qdt)
        x, y = mk(n), mk(n)
        ex, ey = mk(max(N - n, 0)), mk(max(N - n, 0))
        ix, iy = FlatIndex(d, storage), FlatIndex(d, storage)
        ix.add(np.concatenate([x, ex])[:max(N, k)] if N >= n else np.concatenate([x, mk(k)]))
        iy.add(np.concatenate([ey, y])[-max(N, k):] if N >= n else np.concatenate([y, mk(k)]))
        got = compute_sim_with_nonflat_idx(ix, iy, x, y, k, margin)
        want = oracle.margin_scores(x.astype(np.float32), y.astype(np.float32), ix.rows.float().cpu().numpy(), iy.rows.float().cpu().numpy(),
                                    k, margin, storage)
        tol = (8e-6 if storage == "fp16" else 8e-5) * 1024.0 / d
        err = float(np.abs(got - want).max())
        if not np.isfinite(err) or err > tol:
            # Knife-edge: a 16-bit query row whose squared norm sits on a power of four has 1/norm on a power of
            # two, so that every scaled element is again a 16-bit number and many of them are exact rounding ties
            # of the storage type; the last bit of the sum of squares (summation order) then moves them all.
            off = np.nonzero(~(np.abs(got - want) <= tol))[0]
            def on_edge(v):
                ss = float((v.astype(np.float64) ** 2).sum())
                return ss > 0 and abs(np.log2(ss) / 2 - round(np.log2(ss) / 2)) < 1e-5
            if len(off) <= 3 and all(on_edge(x[j]) or on_edge(y[j]) for j in off):
                edges += 1
                print("norm knife-edge", dict(n=n, d=d, k=k, storage=storage, rows=[int(j) for j in off]), err, flush=True)
                continue
            bad += 1
            print("MISMATCH", dict(n=n, N=ix.ntotal, d=d, k=k, storage=storage, margin=margin, q=str(np.dtype(qdt))), err, flush=True)
    print(f"margin fuzz: {cases} cases, {bad} mismatches, {edges} norm knife-edges, {time.time() - t0:.0f} s")
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    sys.exit(1 if run(a.cases, a.seed) else 0)

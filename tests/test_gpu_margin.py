"""Margin scoring on the GPU through the C ABI (svx_unit_rows, svx_knn_mean_sim, svx_margin_scores and the
svx.postprocess mirror) against
(a) the reference's shipped example (tests/golden/margin_example.npz: rows of its Flat indexes + the
    scores of its margin file) -- tolerance 3e-4, see tests/test_margin_cpu.py for why;
(b) the CPU oracle on the same inputs.  Both sides multiply the same fp16 / bf16 values and differ in fp32
    summation order (about 3e-8 on a mean) and, rarely, in ONE rounding of a normalised query element
    (the row's sum of squares may differ by an fp32 ulp, which can move an element across a storage
    rounding boundary: one fp16 ulp = 2^-11, one bf16 ulp = 2^-8 of an element of size ~d^-1/2).  Hence
    4e-6 (fp16) / 4e-5 (bf16) on the k-NN means and 1e-5 on fp16 margin scores."""
TOL = {"fp16": 4e-6, "bf16": 4e-5}
import os

import numpy as np
import pytest

from test_margin_cpu import unit_rows

pytestmark = pytest.mark.gpu
GD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def t():
    import torch
    return torch


def stored(orc, x, storage):
    return orc.round_storage(orc.normalize_l2(x), storage)


def make_index(x, storage):
    from svx.postprocess.flat_index import FlatIndex
    idx = FlatIndex(d=x.shape[1], storage=storage)
    idx.add(x)
    return idx


def test_shipped_margin_example(orc):
    from svx.postprocess.flat_index import FlatIndex
    from svx.postprocess.score_align import compute_sim_with_nonflat_idx
    g = np.load(os.path.join(GD, "margin_example.npz"))
    ix, iy = FlatIndex(1024), FlatIndex(1024)
    ix.add_unit_rows(g["db_src"])
    iy.add_unit_rows(g["db_tgt"])
    assert ix.ntotal == iy.ntotal == 347
    x, y = g["db_src"].astype(np.float32), g["db_tgt"].astype(np.float32)
    s = compute_sim_with_nonflat_idx(ix, iy, x, y, 16, "ratio")
    assert s.dtype == np.float32 and s.shape == (347,)
    assert np.abs(s - g["expected"]).max() < 3e-4
    want = orc.margin_scores(x, y, x, y, 16, "ratio", "fp16")
    assert np.abs(s - want).max() < 1e-5
    # fp16 query files give the same answer as their fp32 widening
    s16 = compute_sim_with_nonflat_idx(ix, iy, g["db_src"], g["db_tgt"], 16, "ratio")
    assert np.abs(s16 - s).max() < 5e-6


@pytest.mark.parametrize("storage", ["fp16", "bf16"])
def test_unit_rows_matches_oracle(orc, t, storage):
    x = unit_rows(300, 256, 3) * np.float32(7.5)
    x[17] = 0  # a zero row stays zero (faiss.normalize_L2 leaves it alone)
    idx = make_index(x, storage)
    got = idx.rows.float().cpu().numpy()
    want = stored(orc, x, storage)
    ulp = 2.0 ** -10 if storage == "fp16" else 2.0 ** -7
    assert np.abs(got - want).max() <= ulp * np.abs(want).max()     # the sum of squares may differ by one fp32 ulp
    assert (got != want).mean() < 0.01
    assert not got[17].any()
    for dt in (np.float16,):
        h = make_index(x.astype(dt), storage).rows.float().cpu().numpy()
        assert np.abs(h - stored(orc, x.astype(dt).astype(np.float32), storage)).max() <= ulp * np.abs(want).max()


@pytest.mark.parametrize("n,N,d,k,storage", [
    (1000, 3333, 1024, 16, "fp16"),
    (77, 64, 1024, 64, "fp16"),      # k = all rows of a two-tile database
    (130, 517, 256, 1, "bf16"),
    (64, 33, 32, 7, "fp16"),         # one k-step, ragged last tile
    (5, 4096, 512, 16, "bf16"),
])
def test_knn_mean_sim_matches_oracle(orc, n, N, d, k, storage):
    q = unit_rows(n, d, 100 + n, 9) * np.float32(3.0)
    dbsrc = unit_rows(N, d, 200 + N, 9)
    idx = make_index(dbsrc, storage)
    db = idx.rows.float().cpu().numpy()
    got = idx.mean_sim(q, k).cpu().numpy()
    want = orc.knn_mean_sim(q, db, k, storage)
    assert got.shape == (n,) and np.abs(got - want).max() < TOL[storage]
    assert np.median(np.abs(got - want)) < 1e-7
    # database order is irrelevant
    perm = np.random.RandomState(0).permutation(N)
    idx2 = make_index(dbsrc[perm], storage)
    assert np.abs(idx2.mean_sim(q, k).cpu().numpy() - got).max() < 1e-6


@pytest.mark.parametrize("margin", ["ratio", "distance"])
def test_margin_scores_match_oracle(orc, margin):
    n, d = 700, 1024
    x, y = unit_rows(n, d, 1, 12), unit_rows(n, d, 2, 12)
    extra_x, extra_y = unit_rows(900, d, 3, 12), unit_rows(1100, d, 4, 12)
    ix, iy = make_index(np.concatenate([x, extra_x]), "fp16"), make_index(np.concatenate([extra_y, y]), "fp16")
    from svx.postprocess.score_align import compute_sim_with_nonflat_idx
    x0, y0 = x.copy(), y.copy()
    got = compute_sim_with_nonflat_idx(ix, iy, x, y, 16, margin)
    assert np.array_equal(x, x0) and np.array_equal(y, y0)
    want = orc.margin_scores(x, y, ix.rows.float().cpu().numpy(), iy.rows.float().cpu().numpy(), 16, margin, "fp16")
    assert np.abs(got - want).max() < 1e-5
    with pytest.raises(ValueError, match="Wrong margin type: cosine"):
        compute_sim_with_nonflat_idx(ix, iy, x, y, 16, "cosine")


def test_large_query_block_path(orc):
    """n >= 32768 switches the kernel to two 16-row blocks per wave; checked on a sample of rows against the
    oracle and on all rows against the small-block path (same values, different tiling)."""
    n, N, d, k = 32768 + 77, 1500, 1024, 16
    q = unit_rows(n, d, 5, 20)
    idx = make_index(unit_rows(N, d, 6, 20), "fp16")
    got = idx.mean_sim(q, k).cpu().numpy()
    db = idx.rows.float().cpu().numpy()
    pick = np.r_[0:300, 16000:16300, n - 300:n]
    assert np.abs(got[pick] - orc.knn_mean_sim(q[pick], db, k, "fp16")).max() < TOL["fp16"]
    small = np.concatenate([idx.mean_sim(q[i:i + 8192], k).cpu().numpy() for i in range(0, n, 8192)])
    assert np.abs(small - got).max() < 1e-6


def test_argument_errors():
    from svx import _lib
    from svx.postprocess.flat_index import FlatIndex
    idx = make_index(unit_rows(10, 64, 1), "fp16")
    with pytest.raises(_lib.SvxError, match="fewer than k"):
        idx.mean_sim(unit_rows(3, 64, 2), 16)
    with pytest.raises(_lib.SvxError, match="supported 1..64"):
        idx.mean_sim(unit_rows(3, 64, 2), 65)
    with pytest.raises(_lib.SvxError, match="multiple of 32"):
        make_index(unit_rows(10, 40, 1), "fp16")
    with pytest.raises(ValueError, match="fp16 or bf16"):
        FlatIndex(64, "fp32")
    assert idx.mean_sim(np.zeros((0, 64), np.float32), 4).shape == (0,)


def test_cli_end_to_end(orc, tmp_path):
    """prep_index (both sides) + score_align on a small corpus laid out like the reference's pipeline."""
    from svx.postprocess import prep_index, score_align
    rs = np.random.RandomState(0)
    d = 1024
    emb_dir, ali_dir = tmp_path / "embed" / "en-de", tmp_path / "align" / "en-de"
    emb_dir.mkdir(parents=True), ali_dir.mkdir(parents=True)
    meta, xs, ys = [], [], []
    for p, n in enumerate((40, 1, 75)):
        sid, tid = f"doc{p}_en", f"doc{p}_de"
        meta.append((f"/a/{sid}.ogg", f"/a/{tid}.ogg"))
        for side, store in (("src", xs), ("tgt", ys)):
            e = unit_rows(n + 5, d, 10 * p + (side == "tgt"), 8).astype(np.float16)
            e.tofile(emb_dir / f"{sid}-{tid}.{side}.embed")
            rows = rs.permutation(n + 5)[:n]
            with open(emb_dir / f"{sid}-{tid}.{side}.tsv", "w") as f:
                for r in rows:
                    f.write(f"{emb_dir / f'{sid}-{tid}.{side}.embed'}\t{r}\n")
            store.append(e[rows])
        with open(ali_dir / f"{sid}-{tid}.txt", "w") as f:
            for i in range(n):
                f.write(f"[{i}]:[{i}, {i + 1}]:0.{i}\n")
    meta.append(("/a/missing_en.ogg", "/a/missing_de.ogg"))
    with open(tmp_path / "metadata.tsv", "w") as f:
        for s, tg in meta:
            f.write(f"{s}\t{tg}\n")
    common = ["--src_lang", "en", "--tgt_lang", "de", "--embed_fp16"]
    prep_index.main([str(tmp_path / "metadata.tsv"), str(tmp_path / "index"), "--data_dir", str(tmp_path / "embed")] + common)
    prep_index.main([str(tmp_path / "metadata.tsv"), str(tmp_path / "index"), "--data_dir", str(tmp_path / "embed"), "--use_tgt"] + common)
    assert (tmp_path / "index" / "en-de" / "en" / "Flat.populate.idx").exists()
    score_align.main([str(tmp_path / "metadata.tsv"), str(tmp_path / "margin"), "--embed_dir", str(tmp_path / "embed"),
                      "--align_dir", str(tmp_path / "align"), "--index_dir", str(tmp_path / "index"), "--k", "16"] + common)
    X, Y = np.concatenate(xs).astype(np.float32), np.concatenate(ys).astype(np.float32)
    want = orc.margin_scores(X, Y, stored(orc, X, "fp16"), stored(orc, Y, "fp16"), 16, "ratio", "fp16")
    got, at = [], 0
    for p, n in enumerate((40, 1, 75)):
        lines = (tmp_path / "margin" / "en-de" / f"doc{p}_en-doc{p}_de.txt").read_text().splitlines()
        assert len(lines) == n
        for i, l in enumerate(lines):
            src, tgt, sc = l.split(":")
            assert (src, tgt) == (f"[{i}]", f"[{i}, {i + 1}]")
            got.append(float(sc))
    assert np.abs(np.array(got, np.float32) - want).max() < 1e-5   # (index rows went through an fp32 file)


@pytest.mark.parametrize("storage,k,n,shards", [("fp16", 16, 700, [300, 0, 5, 395]), ("bf16", 8, 260, [3, 257]),
                                                 ("fp16", 40, 500, [20, 30, 450]), ("fp16", 16, 20000, [7000, 13000])])
def test_knn_topk_merge_shard_by_shard(orc, t, storage, k, n, shards):
    """svx_knn_topk_merge: the database handed over shard by shard (shards smaller than k, an empty shard, k > 16: the
    LDS lists) leaves the same k nearest neighbours as one search of the whole database -- the kept similarities are
    the same set, so the means agree to the summation order of k fp32 values -- and agrees with the oracle."""
    from svx.postprocess.flat_index import FlatIndex
    d = 256
    db = unit_rows(n, d, 21)
    q = unit_rows(333, d, 22)
    whole = make_index(db, storage)
    want = whole.mean_sim(q, k).cpu().numpy()
    topk, mean, lo = None, None, 0
    for m in shards:
        part = FlatIndex(d=d, storage=storage)
        part.add_unit_rows(whole.rows[lo:lo + m])
        topk, mean = part.merge_topk(q, k, topk, want_mean=True)
        lo += m
    assert lo == n
    got = mean.cpu().numpy()
    assert np.abs(got - want).max() < 2e-7
    assert np.allclose(np.sort(topk.cpu().numpy(), axis=1).mean(axis=1), want, atol=2e-7)
    assert np.abs(got - orc.knn_mean_sim(q, stored(orc, db, storage), k, storage)).max() < TOL[storage]


def test_global_margin_ring_exchange_world_of_one(orc, t):
    """global_margin_scores(exchange="ring") on one process: the ring has a single shard, so the path through
    ring_shards -> FlatIndex.merge_topk (svx_knn_topk_merge) -> svx_margin_scores runs on the device without any
    communication, and must agree with the all-gather path and with the oracle."""
    from svx.postprocess.score_align import global_margin_scores
    n, d, k = 900, 256, 16
    x, y = unit_rows(n, d, 31), unit_rows(n, d, 32)
    xd, yd = t.from_numpy(x).cuda(), t.from_numpy(y).cuda()
    ring = global_margin_scores(xd, yd, k=k, exchange="ring").cpu().numpy()
    gathered = global_margin_scores(xd, yd, k=k, exchange="allgather").cpu().numpy()
    default = global_margin_scores(xd, yd, k=k).cpu().numpy()
    assert np.array_equal(default, gathered)
    assert np.abs(ring - gathered).max() < 2e-7
    want = orc.margin_scores(x, y, stored(orc, x, "fp16"), stored(orc, y, "fp16"), k, "ratio", "fp16")
    assert np.abs(ring - want).max() < 1e-5
    with pytest.raises(ValueError):
        global_margin_scores(xd[:5], yd[:5], k=k, exchange="ring")   # fewer rows than k


def test_global_margin_over_rccl():
    """BASELINE configs[4]'s exchange on real devices: one process per GPU, every rank scoring its own shard against the
    union of all ranks' unit rows -- once with the shards travelling round the ring (ring_shards + svx_knn_topk_merge,
    point-to-point over RCCL) and once all-gathered (all_gather_rows); both against one GPU holding everything.  Needs two
    visible GPUs (the driver's one-GPU box skips it; gloo covers the logic on CPU in tests/test_margin_cpu.py)."""
    import os
    import subprocess
    import sys
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs")
    world = min(n, 4)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                          "--master-port", "29671", os.path.join(os.path.dirname(__file__), "rccl_margin_worker.py")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "RCCL_MARGIN ring max|diff|" in out.stdout and "RCCL_MARGIN allgather max|diff|" in out.stdout

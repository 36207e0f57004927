"""Margin scoring (svecalign/postprocess/score_align.py) on the CPU side:
* the oracle against the reference's shipped example -- tests/golden/margin_example.npz holds the rows
  of the example's two populated Flat indexes and the scores of its margin file.  Tolerance 3e-4: the
  example was written by faiss' fp16 GPU search (L2^2 through precomputed fp16 norms), the oracle by an
  exact product; the observed difference is 1.7e-4 on scores of about 1.3;
* host logic of the mirror: faiss Flat file reader / writer, tsv loader, output writer, pair filter;
* the all-gather that assembles the global database, world_size 2 over gloo."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GD = os.path.join(ROOT, "tests", "golden")
REF_IDX = "/root/reference/example/voxpopuli/align_0.7_clean_cat3_min1s_embed_indexes/en-de/en/Flat.populate.idx"


def unit_rows(n, d, seed, clusters=0):
    rs = np.random.RandomState(seed)
    x = rs.randn(n, d).astype(np.float32)
    if clusters:  # neighbours that are actually near: a few centres plus noise
        c = rs.randn(clusters, d).astype(np.float32)
        x = c[rs.randint(0, clusters, n)] + 0.7 * x
    return x


def test_oracle_reproduces_shipped_margin_example(orc):
    g = np.load(os.path.join(GD, "margin_example.npz"))
    dbx, dby = g["db_src"].astype(np.float32), g["db_tgt"].astype(np.float32)
    for storage in ("fp16", "fp32"):
        s = orc.margin_scores(dbx, dby, dbx, dby, 16, "ratio", storage)
        assert s.dtype == np.float32 and s.shape == g["expected"].shape
        assert np.abs(s - g["expected"]).max() < 3e-4
    with pytest.raises(ValueError, match="Wrong margin type: nope"):
        orc.margin_scores(dbx[:20], dby[:20], dbx, dby, 16, "nope")


def test_oracle_margin_properties(orc):
    x, y = unit_rows(40, 64, 1, 5), unit_rows(40, 64, 2, 5)
    db = np.concatenate([orc.round_storage(orc.normalize_l2(y), "fp16"), orc.round_storage(orc.normalize_l2(unit_rows(100, 64, 3, 5)), "fp16")])
    m = orc.knn_mean_sim(x, db, 4)
    # brute force in python for a few rows
    q = orc.round_storage(orc.normalize_l2(x), "fp16").astype(np.float64)
    for i in (0, 7, 39):
        sims = sorted((float(q[i] @ db[j].astype(np.float64)) for j in range(db.shape[0])), reverse=True)
        assert abs(m[i] - np.mean(sims[:4])) < 1e-6
    # k = all rows: the mean similarity to the whole database; row order is irrelevant
    perm = np.random.RandomState(0).permutation(db.shape[0])
    assert np.allclose(orc.knn_mean_sim(x, db, 9), orc.knn_mean_sim(x, db[perm], 9), atol=1e-7)
    # scaling a query does not change its score (normalize_L2)
    assert np.allclose(orc.knn_mean_sim(3.0 * x, db, 4), m, atol=1e-6)
    b = orc.round_storage(np.array([[1.0, 1 / 3, 3.1415927, 65504.0, 1e-3]], np.float32), "bf16")
    assert np.array_equal(b.view(np.uint32) & 0xFFFF, np.zeros_like(b, dtype=np.uint32))


def test_faiss_flat_file_roundtrip(tmp_path):
    from svx.postprocess.flat_index import read_faiss_flat, write_faiss_flat
    rows = unit_rows(37, 64, 5)
    write_faiss_flat(tmp_path / "Flat.populate.idx", rows)
    back = read_faiss_flat(tmp_path / "Flat.populate.idx")
    assert back.shape == (37, 64) and np.array_equal(np.asarray(back), rows)
    write_faiss_flat(tmp_path / "empty.idx", np.zeros((0, 64), np.float32))
    assert read_faiss_flat(tmp_path / "empty.idx").shape == (0, 64)
    (tmp_path / "ivf.idx").write_bytes(b"IwFl" + bytes(64))
    with pytest.raises(NotImplementedError, match="only Flat indexes"):
        read_faiss_flat(tmp_path / "ivf.idx")
    (tmp_path / "short.idx").write_bytes(b"IxF2")
    with pytest.raises(ValueError):
        read_faiss_flat(tmp_path / "short.idx")


@pytest.mark.skipif(not os.path.exists(REF_IDX), reason="reference example not present")
def test_reader_on_the_reference_index_file(tmp_path):
    """Our writer produces byte-for-byte the file faiss wrote for the reference's example."""
    from svx.postprocess.flat_index import read_faiss_flat, write_faiss_flat
    rows = read_faiss_flat(REF_IDX)
    g = np.load(os.path.join(GD, "margin_example.npz"))
    assert np.array_equal(np.asarray(rows), g["db_src"].astype(np.float32))
    assert np.abs(np.linalg.norm(rows, axis=1) - 1).max() < 1e-3
    write_faiss_flat(tmp_path / "w.idx", np.asarray(rows))
    assert (tmp_path / "w.idx").read_bytes() == open(REF_IDX, "rb").read()


def test_tsv_loader_and_writers(tmp_path):
    from svx.postprocess.prep_index import find_embed_files, load_embed_from_tsv
    from svx.postprocess.score_align import find_valid_metas, write_to_output
    a, b = unit_rows(9, 1024, 1).astype(np.float16), unit_rows(5, 1024, 2).astype(np.float16)
    a.tofile(tmp_path / "a.embed")
    b.tofile(tmp_path / "b.embed")
    order = [("a", 3), ("b", 4), ("a", 0), ("b", 0), ("a", 8), ("a", 3)]
    with open(tmp_path / "p-q.src.tsv", "w") as f:
        for name, row in order:
            f.write(f"{tmp_path / (name + '.embed')}\t{row}\n")
    got = load_embed_from_tsv(tmp_path / "p-q.src.tsv", fp16_embed=True, use_stopes=False)
    want = np.stack([(a if n == "a" else b)[r] for n, r in order])
    assert got.dtype == np.float16 and np.array_equal(got, want)
    (tmp_path / "p-q.tgt.tsv").write_text((tmp_path / "p-q.src.tsv").read_text())
    (tmp_path / "r-s.src.tsv").write_text("")
    meta = [("x/p.wav", "y/q.wav"), ("x/m.wav", "y/n.wav")]
    assert find_valid_metas(meta, tmp_path) == ["p-q"]
    assert find_embed_files(meta, tmp_path, use_tgt=True) == [tmp_path / "p-q.tgt.tsv"]
    with pytest.raises(Exception, match="r-s.src.tsv"):
        find_valid_metas([("r.wav", "s.wav")], tmp_path)
    ad, od = tmp_path / "al", tmp_path / "out"
    ad.mkdir(), od.mkdir()
    (ad / "p-q.txt").write_text("[0, 1]:[0]:0.25\n[2]:[1, 2]:0.5\n")
    write_to_output(ad, ["p-q"], np.array([1.25, 0.75], np.float32), od)
    assert (od / "p-q.txt").read_text() == "[0, 1]:[0]:1.25\n[2]:[1, 2]:0.75\n"
    with pytest.raises(AssertionError):
        write_to_output(ad, ["p-q"], np.array([1.0, 2.0, 3.0], np.float32), od)


def _gather_worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "speech-vecalign_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from svx.postprocess.flat_index import all_gather_rows
    sizes = [23, 0, 41][:world] if world == 3 else [23, 41][:world]
    lo = sum(sizes[:rank])
    x_all, y_all = unit_rows(sum(sizes), 64, 11, 6), unit_rows(sum(sizes), 64, 12, 6)
    x, y = x_all[lo:lo + sizes[rank]], y_all[lo:lo + sizes[rank]]
    out = {}
    for storage, tdt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        sx = torch.from_numpy(oracle.round_storage(oracle.normalize_l2(x), storage)).to(tdt)
        sy = torch.from_numpy(oracle.round_storage(oracle.normalize_l2(y), storage)).to(tdt)
        gx, gy = all_gather_rows(sx, None), all_gather_rows(sy, None)
        assert gx.dtype == tdt and gx.shape == (sum(sizes), 64)
        out[storage] = oracle.margin_scores(x, y, gx.float().numpy(), gy.float().numpy(), 8, "ratio", storage)
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_global_database_all_gather_gloo(orc, world):
    """Ranks hold disjoint alignments (one of them possibly none); after the all-gather every rank scores
    its rows against the union, and the concatenation equals the single-process result."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, 29640 + world, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 64
    x_all, y_all = unit_rows(n, 64, 11, 6), unit_rows(n, 64, 12, 6)
    for storage in ("fp16", "bf16"):
        dbx = orc.round_storage(orc.normalize_l2(x_all), storage)
        dby = orc.round_storage(orc.normalize_l2(y_all), storage)
        want = orc.margin_scores(x_all, y_all, dbx, dby, 8, "ratio", storage)
        got = np.concatenate([g[storage] for g in gathered])
        assert np.array_equal(got, want)


def _ring_worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "speech-vecalign_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from svx.postprocess.flat_index import ring_shards
    sizes = [23, 0, 41][:world] if world == 3 else [23, 41][:world]
    lo = sum(sizes[:rank])
    x_all, y_all = unit_rows(sum(sizes), 64, 11, 6), unit_rows(sum(sizes), 64, 12, 6)
    x, y = x_all[lo:lo + sizes[rank]], y_all[lo:lo + sizes[rank]]
    out = {}
    for storage, tdt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        sx = torch.from_numpy(oracle.round_storage(oracle.normalize_l2(x), storage)).to(tdt)
        sy = torch.from_numpy(oracle.round_storage(oracle.normalize_l2(y), storage)).to(tdt)
        owners, px, py = [], [], []
        for owner, rows in ring_shards(sx, None):
            owners.append(owner)
            assert rows.dtype == tdt and rows.shape == (sizes[owner], 64)
            px.append((owner, rows.clone()))   # (the wire buffer is reused two steps later)
        for owner, rows in ring_shards(sy, None):
            py.append((owner, rows.clone()))
        assert owners == [(rank - s) % world for s in range(world)]   # own shard first, then round the ring
        gx = torch.cat([r for _, r in sorted(px, key=lambda t: t[0])])
        gy = torch.cat([r for _, r in sorted(py, key=lambda t: t[0])])
        out[storage] = oracle.margin_scores(x, y, gx.float().numpy(), gy.float().numpy(), 8, "ratio", storage)
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_global_database_ring_gloo(orc, world):
    """The ring exchange (svx.postprocess.flat_index.ring_shards): every rank sees every rank's shard exactly once,
    its own first and then those of ranks r-1, r-2, ... (one of them possibly empty), bit-identical rows; scoring
    against their union equals the single-process result."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, 29650 + world, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 64
    x_all, y_all = unit_rows(n, 64, 11, 6), unit_rows(n, 64, 12, 6)
    for storage in ("fp16", "bf16"):
        dbx = orc.round_storage(orc.normalize_l2(x_all), storage)
        dby = orc.round_storage(orc.normalize_l2(y_all), storage)
        want = orc.margin_scores(x_all, y_all, dbx, dby, 8, "ratio", storage)
        got = np.concatenate([g[storage] for g in gathered])
        assert np.array_equal(got, want)

"""Host-side logic of the path (no GPU): alignment types, search parameters, file formats, candidate
index tables, scoring, sharding, CLI surface.  Where the reference is present (build container) the
results are compared with the reference's own functions; otherwise with fixtures / known answers."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_loader  # noqa: E402

TRIM = os.path.join(ROOT, "tests", "golden", "example_trim")
needs_ref = pytest.mark.skipif(not ref_loader.available(), reason="reference not present (GPU box)")


def test_alignment_types_and_search_params():
    from svx.vecalign import vecalign as V
    assert V.make_alignment_types(3) == [(1, 1), (1, 2), (2, 1)]
    assert len(V.make_alignment_types(6)) == 15 and len(V.make_alignment_types(5)) == 10
    assert V.make_alignment_types(4) == [(1, 1), (1, 2), (1, 3), (2, 1), (2, 2), (3, 1)]
    assert V.make_many_to_one_alignment_types(3) == [(1, 1), (2, 1), (3, 1)]
    types, sk, tk, W = V.resolve_search_params(6, None, 5)
    assert (len(types), sk, tk, W) == (15, 5, 5, 8)
    types, sk, tk, W = V.resolve_search_params(5, None, 5)
    assert (len(types), sk, tk, W) == (10, 4, 4, 7)
    types, sk, tk, W = V.resolve_search_params(1, None, 5)  # clamped to 2 (vecalign.py:230-232)
    assert (types, sk, tk, W) == ([(1, 1)], 1, 1, 6)
    types, sk, tk, W = V.resolve_search_params(10, 4, 5)
    assert (types, sk, tk, W) == ([(1, 1), (2, 1), (3, 1), (4, 1)], 4, 1, 7)


@needs_ref
def test_types_match_reference():
    from svx.vecalign import vecalign as V
    ref = ref_loader.load()
    for a in range(2, 12):
        assert V.make_alignment_types(a) == ref.vecalign.make_alignment_types(a)
        assert V.make_many_to_one_alignment_types(a) == ref.vecalign.make_many_to_one_alignment_types(a)


def test_alignment_files_round_trip(tmp_path):
    from svx.utils import file_utils as F
    from svx.vecalign.vecalign import print_alignments
    al = [([0, 1], [0]), ([], [1]), ([2], [2, 3, 4]), ([3], [])]
    sc = [0.349737, 0.0, 1.25, 0.0]
    p = tmp_path / "a.txt"
    with open(p, "w") as fp:
        print_alignments(al, scores=sc, ofile=fp)
    assert p.read_text().splitlines() == ["[0, 1]:[0]:0.349737", "[]:[1]:0.000000", "[2]:[2, 3, 4]:1.250000", "[3]:[]:0.000000"]
    assert F.read_alignments(p) == al
    assert F.read_alignments_with_score(p) == [(a[0], a[1], s) for a, s in zip(al, sc)]
    F.write_alignment(al, tmp_path / "b.txt")
    assert F.read_alignments(tmp_path / "b.txt") == al
    segs = [(0, 10), (10, 25), (30, 40), (41, 50)]
    s, t, n = F.alignments_to_timestamps(al, segs, [(0, 5), (6, 9), (9, 12), (12, 20), (21, 30)])
    assert n == 2 and s == [(0, 25), (30, 40)] and t == [(0, 5), (9, 30)]
    exp = F.read_alignments_with_score(os.path.join(TRIM, "expected_seed0.txt"))
    assert len(exp) > 20 and all(len(e) == 3 for e in exp)
    assert F.read_segments(os.path.join(TRIM, "segments_en.txt"))[0][0] >= 0
    with pytest.raises(Exception, match="does not have at least two"):
        (tmp_path / "c.txt").write_text("garbage\n")
        F.read_alignments(tmp_path / "c.txt")


def _load_trim(lang):
    from svx.utils import embedding_utils as E
    s2i, emb = E.read_in_embeddings(os.path.join(TRIM, f"cat_segs_{lang}.txt"), os.path.join(TRIM, f"embeds_{lang}.f16"), False, True)
    lines = open(os.path.join(TRIM, f"segments_{lang}.txt")).readlines()
    return s2i, emb, lines


def test_candidate_index_table_fixture():
    from svx.utils import embedding_utils as E
    from svx.vecalign.vecalign import load_ignore_index_file
    s2i, emb, lines = _load_trim("en")
    assert emb.dtype == np.float16 and emb.shape[1] == 1024 and emb.shape[0] == len(s2i)
    ign = load_ignore_index_file(os.path.join(TRIM, "ignore_src.txt"))
    tab = E.candidate_index_table(s2i, lines, 5, ign, overlap_segments=True)
    assert tab.shape == (5, len(lines)) and tab.dtype == np.int32
    for j in range(5):
        assert (tab[j, :j] == -1).all()          # PAD triangle: no j+1 segments end before index j
    assert (tab[0] >= 0).sum() >= len(lines) - len(ign) - 1
    for (s, j) in ign:                            # an ignore entry zeroes every overlap from (s, j) on
        for k in range(j - s, min(len(lines), s + 5) - s):
            assert tab[k, s + k] == -1
    # stopes files are plain .npy: the loader needs no stopes
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "x.npy"), np.asarray(emb[:7]))
        os.rename(os.path.join(td, "x.npy"), os.path.join(td, "x.embed"))
        got = E.load_sent_embeddings(os.path.join(td, "x.embed"), use_stopes=True)
        assert got.dtype == np.float16 and np.array_equal(np.asarray(got), np.asarray(emb[:7]))


@needs_ref
def test_candidate_tensor_matches_reference():
    """table + gather == the reference's make_doc_embedding, bit for bit, on the shipped example."""
    from svx.utils import embedding_utils as E
    ref = ref_loader.load()
    ex = os.path.join(ref.root, "example", "voxpopuli")
    stem = "20180313-0900-PLENARY-15"
    for lang, side in (("en", "src"), ("de", "tgt")):
        cat = os.path.join(ex, "cat_segs", lang, f"{stem}_{lang}.txt")
        s2i, emb = E.read_in_embeddings(cat, os.path.join(ex, "embeds", lang, f"{stem}_{lang}.embed"), use_stopes=True)
        lines = open(os.path.join(ex, "segments", lang, f"{stem}_{lang}.txt")).readlines()
        ign = ref.vecalign.load_ignore_index_file(os.path.join(ex, "untrans_cat_seg_ids", "en-de", f"{stem}_en-{stem}_de.{side}.txt"))
        tab = E.candidate_index_table(s2i, lines, 5, ign, overlap_segments=True)
        mine = np.where(tab[..., None] >= 0, np.asarray(emb, dtype=np.float32)[np.clip(tab, 0, None)], 0.0).astype(np.float32)
        r_s2i = {}
        for i, l in enumerate(open(cat)):
            r_s2i.setdefault(l.strip(), i)
        theirs = ref.embedding_utils.make_doc_embedding(r_s2i, np.asarray(emb, dtype=np.float32), lines, 5, ignore_indices=ign,
                                                        overlap_segments=True)
        assert np.array_equal(mine, theirs)


@needs_ref
def test_score_reproduces_readme_tables():
    """README.md:289-295 / 318-325 quality tables from the shipped alignment files."""
    from svx.utils.file_utils import read_alignments
    from svx.vecalign.score import score_multiple
    ref = ref_loader.load()
    ex = os.path.join(ref.root, "example", "voxpopuli")
    stem = "20180313-0900-PLENARY-15"
    gold = read_alignments(os.path.join(ex, f"{stem}.gold"))
    name = f"{stem}_en-{stem}_de.txt"
    for sub, want in (("alignments", (0.558, 0.632, 0.593, 0.942, 0.993, 0.967)), ("align_0.7", (0.593, 0.632, 0.612, 0.972, 0.978, 0.975))):
        test = read_alignments(os.path.join(ex, sub, "en-de", name))
        mine = score_multiple([gold], [test])
        theirs = ref.score.score_multiple(gold_list=[gold], test_list=[test])
        assert mine == theirs
        got = (mine["precision_strict"], mine["recall_strict"], mine["f1_strict"], mine["precision_lax"], mine["recall_lax"], mine["f1_lax"])
        assert all(abs(g - w) < 6e-4 for g, w in zip(got, want))


def test_score_small_known_answers():
    from svx.vecalign.score import score_multiple
    gold = [([0], [0]), ([1, 2], [1]), ([3], []), ([4], [2, 3])]
    assert score_multiple([gold], [gold])["f1_strict"] == 1.0
    test = [([0], [0]), ([1], [1]), ([2], []), ([3], []), ([4], [2]), ([], [3])]
    r = score_multiple([gold], [test])
    assert r["precision_strict"] == pytest.approx(2 / 6) and r["precision_lax"] == pytest.approx(4 / 6)
    assert r["recall_strict"] == pytest.approx(1 / 3) and r["recall_lax"] == 1.0
    assert score_multiple([[]], [[]])["f1_lax"] == 0.0


def test_shards():
    from svx.utils.mp_utils import balanced_shards, get_shard_range
    assert [get_shard_range(10, 3, r) for r in range(3)] == [(0, 3), (3, 7), (7, 10)]
    with pytest.raises(AssertionError):
        get_shard_range(10, 3, 3)
    costs = list(np.random.RandomState(0).randint(512, 8192, size=101))
    for n in (1, 2, 4, 8):
        sh = balanced_shards(costs, n)
        assert sorted(i for s in sh for i in s) == list(range(101))
        loads = [sum(costs[i] for i in s) for s in sh]
        assert max(loads) - min(loads) <= max(costs)
    from svx.seg_align.align import pair_rng
    assert pair_rng(None, 3) is None
    a, b = pair_rng(7, 3).randint(0, 1 << 30, 5), pair_rng(7, 3).randint(0, 1 << 30, 5)
    assert np.array_equal(a, b) and not np.array_equal(a, pair_rng(7, 4).randint(0, 1 << 30, 5))


def test_cli_surface(tmp_path):
    """Same flags and defaults as seg_align/align.py:13-96 and vecalign.py:36-151."""
    from svx.seg_align import align as A
    from svx.vecalign import vecalign as V
    a = A.parse_args(["meta.tsv", "out", "--src_lang", "en", "--tgt_lang", "de", "--seg_dir", "s", "--concat_dir", "c", "--embed_dir", "e"])
    assert (a.alignment_max_size, a.search_buffer_size, a.del_percentile_frac, a.max_size_full_dp, a.costs_sample_size,
            a.num_samps_for_norm, a.is_stopes_embed, a.fp16_embed, a.ign_indices_dir) == (6, 5, 0.2, 300, 20000, 100, False, False, None)
    v = V.parse_args(["-s", "a", "-t", "b", "--src_embed", "x", "y", "--tgt_embed", "z", "w"])
    assert (v.alignment_max_size, v.many_to_one, v.search_buffer_size, v.overlap_segments, v.print_results) == (10, None, 5, False, False)
    assert V.parse_args(["-s", "a", "-t", "b", "--src_embed", "x", "y", "--tgt_embed", "z", "w", "--many_to_one"]).many_to_one == 50
    # validate_inputs: directory conventions {dir}/{lang}/{stem}.{txt,embed}, output {out}/{s}-{t}.txt
    for sub in ("seg/en", "seg/de", "cat/en", "cat/de", "emb/en", "emb/de", "ign"):
        (tmp_path / sub).mkdir(parents=True)
    for d, suf in (("seg", ".txt"), ("cat", ".txt"), ("emb", ".embed")):
        (tmp_path / d / "en" / ("a_en" + suf)).write_text("x")
        (tmp_path / d / "de" / ("a_de" + suf)).write_text("x")
    (tmp_path / "ign" / "a_en-a_de.src.txt").write_text("1 2\n")
    from pathlib import Path
    got = A.validate_inputs([("/audio/a_en.ogg", "/audio/a_de.ogg"), ("/audio/missing_en.ogg", "/audio/missing_de.ogg")],
                            tmp_path / "seg/en", tmp_path / "seg/de", tmp_path / "cat/en", tmp_path / "cat/de",
                            tmp_path / "emb/en", tmp_path / "emb/de", Path("/out/en-de"), tmp_path / "ign")
    assert len(got) == 1 and got[0].output_path == "/out/en-de/a_en-a_de.txt"
    assert got[0].src_ignore_indices is not None and got[0].tgt_ignore_indices is None
    assert got[0].src_embed_path.endswith("emb/en/a_en.embed")


def test_draw_indices_layout():
    """Layout of the sampled-index arrays documented in include/svx.h:svx_pair."""
    from svx.vecalign import dp_utils
    sizes = dp_utils.level_sizes(1101, 1003, 300)
    assert sizes == [(1101, 1003), (550, 501), (275, 250)]
    ni, ki = dp_utils.draw_indices(1101, 1003, 4, 3, 300, 20000, 100, rng=np.random.RandomState(1))
    assert len(ni) == sum(3 * 34 + 4 * 25 for _ in sizes) and len(ki) == 2 * 20000 * 3
    off = 0
    for s0, s1 in sizes:
        assert ni[off:off + 3 * 34].max() < s1 and ni[off + 3 * 34:off + 3 * 34 + 4 * 25].max() < s0
        off += 3 * 34 + 4 * 25
    ni2, _ = dp_utils.draw_indices(1101, 1003, 4, 3, 300, 20000, 100, rng=np.random.RandomState(1), have_norms0=True)
    assert len(ni2) == len(ni) - 3 * 34
    _, ks = dp_utils.draw_indices(90, 80, 2, 2, 300, 20000, 100, rng=np.random.RandomState(1))
    assert len(ks) == 2 * 7200 and np.array_equal(ks[:80], np.zeros(80)) and np.array_equal(ks[7200:7280], np.arange(80))


def test_concat_segs_and_untranslated_fixture(tmp_path):
    """Candidate enumeration on the trimmed example: every line of the fixture's cat_segs file is
    produced (the fixture is a prefix cut of the reference's own output), in string-sorted order."""
    from svx.seg_align import concat_segs as C
    from svx.seg_align import detect_untranslate_concats as D
    seg = os.path.join(TRIM, "segments_en.txt")
    lines = sorted(C.get_overlaps(seg, 5, int(20.0 * C.SAMPLE_RATE)))
    want = open(os.path.join(TRIM, "cat_segs_en.txt")).read().splitlines()
    assert want == sorted(want) and set(want) <= set(lines)
    C.overlap(seg, tmp_path / "x" / "o.txt", 5, max_dur=20.0)
    assert (tmp_path / "x" / "o.txt").read_text().splitlines() == lines
    segs = [(0, 10), (10, 20), (20, 400000), (400000, 400010), (400010, 400020)]
    assert list(C.candidate_windows(segs, 3, 320000)) == [(0, 0), (0, 1), (1, 1), (3, 3), (3, 4), (4, 4)]
    (tmp_path / "segs.txt").write_text("".join(f"{a} {b}\n" for a, b in segs))
    (tmp_path / "ident.txt").write_text("1\n4\n")
    assert D.get_identical_overlap_ids(tmp_path / "segs.txt", 3, 320000, tmp_path / "ident.txt") == [(0, 1), (1, 1), (3, 4), (4, 4)]


@needs_ref
def test_concat_segs_and_untranslated_match_shipped_example(tmp_path):
    """Byte-exact regeneration of example/voxpopuli/cat_segs and untrans_cat_seg_ids."""
    from svx.seg_align import concat_segs as C
    from svx.seg_align import detect_untranslate_concats as D
    ref = ref_loader.load()
    ex = os.path.join(ref.root, "example", "voxpopuli")
    stem = "20180313-0900-PLENARY-15"
    for lang, side in (("en", "src"), ("de", "tgt")):
        seg = os.path.join(ex, "segments", lang, f"{stem}_{lang}.txt")
        C.overlap(seg, tmp_path / f"{lang}.txt", 5, max_dur=20.0)
        assert (tmp_path / f"{lang}.txt").read_text() == open(os.path.join(ex, "cat_segs", lang, f"{stem}_{lang}.txt")).read()
        got = D.get_identical_overlap_ids(seg, 5, int(20.0 * C.SAMPLE_RATE),
                                          os.path.join(ex, "untrans_segs", "en-de", f"{stem}_en-{stem}_de.{side}.txt"))
        want = open(os.path.join(ex, "untrans_cat_seg_ids", "en-de", f"{stem}_en-{stem}_de.{side}.txt")).read()
        assert "".join(f"{i} {j}\n" for i, j in got) == want


def test_postfilters_known_answers(tmp_path):
    from svx.postprocess import filters as F
    (tmp_path / "a.txt").write_text("[0]:[0]:0.2\n[]:[1]:0.0\n[1, 2]:[2]:0.9\n[3]:[3, 4]:0.7\n")
    assert F.keep_by_cost(str(tmp_path / "a.txt"), str(tmp_path / "o.txt"), max_cost=0.7) == 0.5
    assert (tmp_path / "o.txt").read_text() == "[0]:[0]:0.2\n[3]:[3, 4]:0.7\n"
    segs = [(0, 16000), (16000, 32000), (33000, 48000), (80000, 96000), (96000, 500000)]
    al = [([0], [0]), ([1], [1]), ([2], [2]), ([3], [3]), ([4], [4])]
    got = F.concat_consecutive(al, segs, segs, 3, 1.0, 20.0)
    assert got == [([0], [0]), ([0, 1], [0, 1]), ([0, 1, 2], [0, 1, 2]), ([1], [1]), ([1, 2], [1, 2]), ([2], [2]), ([3], [3]), ([4], [4])]
    (tmp_path / "s.txt").write_text("".join(f"{a} {b}\n" for a, b in segs))
    (tmp_path / "al.txt").write_text("[0]:[0]\n[1]:[]\n[2]:[2]\n[3]:[4]\n")
    assert F.keep_by_duration(tmp_path / "al.txt", tmp_path / "s.txt", tmp_path / "s.txt", 15500, tmp_path / "d.txt") == 2
    # reference quirk kept on purpose (filter_by_dur.py:58-64): timestamps skip deletions but the lines do not,
    # so with a deletion in the input the i-th surviving span is paired with the i-th LINE
    assert (tmp_path / "d.txt").read_text() == "[0]:[0]\n[2]:[2]\n"


@needs_ref
def test_postfilters_regenerate_shipped_example(tmp_path):
    """align_0.7, align_0.7_clean_cat3 and ..._min1s regenerate byte-identically from their predecessors."""
    from svx.postprocess import concat_aligns, filter_by_cost, filter_by_dur
    ref = ref_loader.load()
    ex = os.path.join(ref.root, "example", "voxpopuli")
    name = "en-de/20180313-0900-PLENARY-15_en-20180313-0900-PLENARY-15_de.txt"
    common = ["--src_lang", "en", "--tgt_lang", "de"]
    meta = os.path.join(ex, "metadata.tsv")
    filter_by_cost.main([meta, str(tmp_path / "c"), "--align_dir", os.path.join(ex, "alignments"), "--max_cost", "0.7"] + common)
    assert open(tmp_path / "c" / name).read() == open(os.path.join(ex, "align_0.7", name)).read()
    concat_aligns.main([meta, str(tmp_path / "k"), "--max_num_align", "3", "--align_dir", os.path.join(ex, "align_0.7_clean"),
                        "--seg_dir", os.path.join(ex, "segments"), "--apply_dur_cond_to_both_sides", "--max_dur", "20.0"] + common)
    assert open(tmp_path / "k" / name).read() == open(os.path.join(ex, "align_0.7_clean_cat3", name)).read()
    filter_by_dur.main([meta, str(tmp_path / "d"), "--align_dir", os.path.join(ex, "align_0.7_clean_cat3"),
                        "--seg_dir", os.path.join(ex, "segments")] + common)
    assert open(tmp_path / "d" / name).read() == open(os.path.join(ex, "align_0.7_clean_cat3_min1s", name)).read()


# ---- manifest steps after margin scoring (prep_tsv.py, sort_tsv.py) on the reference's shipped example -------------
POST = os.path.join(os.path.dirname(__file__), "golden", "example_post")


def _gz_lines(path):
    import gzip
    with gzip.open(path, "rt") as f:
        return f.read().splitlines()


def test_prep_tsv_reproduces_shipped_manifest(tmp_path):
    """margin-scored alignments + segment files -> align.tsv.gz, line for line as shipped."""
    import shutil
    from svx.postprocess import prep_tsv
    stem = "20180313-0900-PLENARY-15"
    (tmp_path / "margin" / "en-de").mkdir(parents=True)
    (tmp_path / "seg" / "en").mkdir(parents=True)
    (tmp_path / "seg" / "de").mkdir(parents=True)
    shutil.copy(os.path.join(POST, "margin.txt"), tmp_path / "margin" / "en-de" / f"{stem}_en-{stem}_de.txt")
    shutil.copy(os.path.join(POST, "segments_en.txt"), tmp_path / "seg" / "en" / f"{stem}_en.txt")
    shutil.copy(os.path.join(POST, "segments_de.txt"), tmp_path / "seg" / "de" / f"{stem}_de.txt")
    meta = tmp_path / "metadata.tsv"
    meta.write_text(open(os.path.join(POST, "metadata.tsv")).read() + "a/none_en.ogg\ta/none_de.ogg\n")
    args = [str(meta), str(tmp_path / "tsv"), "--src_lang", "en", "--tgt_lang", "de", "--align_dir", str(tmp_path / "margin"),
            "--seg_dir", str(tmp_path / "seg")]
    prep_tsv.main(args)
    got = _gz_lines(tmp_path / "tsv" / "en-de" / "align.tsv.gz")
    assert got == _gz_lines(os.path.join(POST, "align.tsv.gz")) and len(got) == 347
    with pytest.raises(AssertionError, match="Will not overwrite"):
        prep_tsv.main(args)


def test_sort_tsv_reproduces_shipped_manifest(tmp_path):
    from svx.postprocess import sort_tsv
    out = tmp_path / "o" / "sorted.tsv.gz"
    sort_tsv.main(["--in_tsv", os.path.join(POST, "align.rm_overlap.tsv.gz"), "--out_tsv", str(out)])
    got = _gz_lines(out)
    assert got == _gz_lines(os.path.join(POST, "align.rm_overlap.sort.tsv.gz")) and len(got) == 300
    scores = [float(l.split("\t")[0]) for l in got]
    assert scores == sorted(scores, reverse=True)
    with pytest.raises(AssertionError, match="exists"):
        sort_tsv.main(["--in_tsv", os.path.join(POST, "align.rm_overlap.tsv.gz"), "--out_tsv", str(out)])


def test_native_index_drawing_matches_numpy_stream():
    """svx_draw_indices / svx_mt19937_choice restate RandomState.choice (MT19937 + masked rejection) bit for bit and
    leave the stream where numpy would: per-pair generators and the global legacy stream."""
    from svx.vecalign import dp_utils as D
    cases = [(4096, 4096, 4, 4, 300, 20000, 100, False, False), (237, 217, 5, 5, 300, 20000, 100, False, False),
             (40, 37, 3, 3, 300, 20000, 100, False, False), (1101, 1003, 4, 3, 50, 5000, 30, True, False),
             (700, 1, 2, 2, 300, 20000, 100, False, True), (1, 1, 1, 1, 300, 20000, 100, False, False),
             (333, 1200, 4, 4, 300, 20000, 7, True, True), (65536, 3, 2, 1, 10, 999, 1, False, False)]
    for (n, m, k0, k1, mf, cs, ns, h0, h1) in cases:
        r1, r2 = np.random.RandomState(7), np.random.RandomState(7)
        a, b = D.draw_indices(n, m, k0, k1, mf, cs, ns, r1, h0, h1)
        nn, kn = D.index_counts(n, m, k0, k1, mf, cs, ns, h0, h1)
        no, ko = np.zeros(nn, np.int32), np.zeros(kn, np.int32)
        D.draw_indices_into(no, ko, n, m, k0, k1, mf, cs, ns, r2, h0, h1)
        assert np.array_equal(a[:nn], no) and np.array_equal(b, ko), (n, m)
        assert r1.randint(0, 10 ** 9) == r2.randint(0, 10 ** 9)   # both streams stand at the same place
    np.random.seed(3)
    a, b = D.draw_indices(900, 800, 4, 4, 300, 20000, 100)
    after = np.random.randint(0, 10 ** 9)
    np.random.seed(3)
    no, ko = np.zeros(len(a), np.int32), np.zeros(len(b), np.int32)
    D.draw_indices_into(no, ko, 900, 800, 4, 4, 300, 20000, 100)
    assert np.array_equal(a, no) and np.array_equal(b, ko) and np.random.randint(0, 10 ** 9) == after


@pytest.mark.parametrize("fixture", ["example_trim", "example_full"])
def test_native_candidate_table_and_formatter(fixture, tmp_path):
    """svx_candidate_table == the Python mirror of make_overlap + dict lookup (string keys, ignore entries, PAD
    triangle) on the shipped example; svx_format_alignments == print_alignments byte for byte."""
    import io
    from svx.seg_align.align import format_alignment_rows
    from svx.utils import embedding_utils as E
    from svx.utils.file_utils import read_alignments_with_score
    from svx.vecalign.dp_utils import alignments_to_rows
    from svx.vecalign.vecalign import load_ignore_index_file, print_alignments
    D = os.path.join(ROOT, "tests", "golden", fixture)
    for lang, side in (("en", "src"), ("de", "tgt")):
        s2i, emb = E.read_in_embeddings(f"{D}/cat_segs_{lang}.txt", f"{D}/embeds_{lang}.f16", False, True)
        lines = open(f"{D}/segments_{lang}.txt").readlines()
        ign = load_ignore_index_file(f"{D}/ignore_{side}.txt")
        for K, ig, igf in ((5, ign, f"{D}/ignore_{side}.txt"), (3, None, None), (7, ign, f"{D}/ignore_{side}.txt")):
            want = E.candidate_index_table(s2i, lines, K, ig, overlap_segments=True)
            got, ncand = E.candidate_table_from_files(f"{D}/segments_{lang}.txt", f"{D}/cat_segs_{lang}.txt", K, igf)
            assert np.array_equal(want, got) and ncand == emb.shape[0]
        host = E.read_embeddings_pinned(f"{D}/embeds_{lang}.f16", False, True)
        assert np.array_equal(host.numpy(), np.asarray(emb))
    # duplicate candidate lines keep the first row; CRLF and blank lines; a missing candidate is -1
    (tmp_path / "seg.txt").write_text("0 10\r\n10 25\n\n")
    (tmp_path / "cat.txt").write_text("10 25\n0 10\n0 25\n0 10\n")
    with pytest.raises(Exception, match="start and an end"):
        E.candidate_table_from_files(str(tmp_path / "seg.txt"), str(tmp_path / "cat.txt"), 2)
    (tmp_path / "seg.txt").write_text("0 10\r\n10 25\n25 31\n")
    got, ncand = E.candidate_table_from_files(str(tmp_path / "seg.txt"), str(tmp_path / "cat.txt"), 2)
    assert ncand == 4 and got.tolist() == [[1, 0, -1], [-1, 2, -1]]
    exp = read_alignments_with_score(os.path.join(D, "expected_seed0.txt"))
    al, sc = [(a, b) for a, b, _ in exp], np.array([s for _, _, s in exp]) + 1e-7
    buf = io.StringIO()
    print_alignments(al, scores=sc, ofile=buf)
    assert format_alignment_rows(alignments_to_rows(al), sc) == buf.getvalue().encode()
    buf = io.StringIO()
    print_alignments(al, ofile=buf)
    assert format_alignment_rows(alignments_to_rows(al), None) == buf.getvalue().encode()

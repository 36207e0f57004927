"""The reference's one shipped known answer for the alignment path -- example/voxpopuli (237 x 217 segments,
1148 / 1035 candidates, real SONAR fp16 embeddings, -a 6: 15 types, band 16, no pyramid levels, sampled
deletion knob) -- as data under tests/golden/example_full (generator: tests/golden/make_golden.py example_full).

  * spans must equal the shipped 156-line alignment file for np.random.seed(0 / 1 / 42) (the reference is
    unseeded: its scores move with the stream, its spans do not -- SURVEY.md 8c);
  * scores must equal the REAL reference's output for the same seed (expected_seed*.txt; 6 printed decimals);
  * P / R / F against the shipped gold file must equal README.md:289-295.

CPU: the oracle.  GPU (-m gpu): the HIP path through the reference's command line (svx.seg_align.align)."""
import os
import shutil

import numpy as np
import pytest

FULL = os.path.join(os.path.dirname(__file__), "golden", "example_full")
SEEDS = (0, 1, 42)
README_TABLE = (0.558, 0.632, 0.593, 0.942, 0.993, 0.967)  # strict P R F, lax P R F (README.md:289-295)


def parse(path):
    from svx.utils.file_utils import read_alignments_with_score
    return read_alignments_with_score(path)


def check_against_files(got, seed):
    shipped = parse(os.path.join(FULL, "shipped_alignment.txt"))
    want = parse(os.path.join(FULL, "expected_seed%d.txt" % seed))
    assert len(got) == len(shipped) == 156
    assert [(a, b) for a, b, _ in got] == [(a, b) for a, b, _ in shipped]
    assert [(a, b) for a, b, _ in want] == [(a, b) for a, b, _ in shipped]
    assert max(abs(g[2] - w[2]) for g, w in zip(got, want)) < 1e-4 + 5e-7  # north-star tolerance; the files hold 6 decimals


def check_quality(alignments):
    from svx.utils.file_utils import read_alignments
    from svx.vecalign.score import score_multiple
    r = score_multiple([read_alignments(os.path.join(FULL, "gold.txt"))], [alignments])
    got = (r["precision_strict"], r["recall_strict"], r["f1_strict"], r["precision_lax"], r["recall_lax"], r["f1_lax"])
    assert all(abs(g - w) < 6e-4 for g, w in zip(got, README_TABLE)), got


def load_side(lang, side):
    """Host-side candidate tensor of one document, float32 [5][n][1024] (what the reference feeds vecalign())."""
    from svx.utils import embedding_utils as E
    from svx.vecalign.vecalign import load_ignore_index_file
    s2i, emb = E.read_in_embeddings(os.path.join(FULL, f"cat_segs_{lang}.txt"), os.path.join(FULL, f"embeds_{lang}.f16"), False, True)
    lines = open(os.path.join(FULL, f"segments_{lang}.txt")).readlines()
    ign = load_ignore_index_file(os.path.join(FULL, f"ignore_{side}.txt"))
    tab = E.candidate_index_table(s2i, lines, 5, ign, overlap_segments=True)
    emb32 = np.asarray(emb, dtype=np.float32)
    return np.where(tab[..., None] >= 0, emb32[np.clip(tab, 0, None)], 0.0).astype(np.float32)


@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_reproduces_shipped_alignment(orc, seed, tmp_path):
    from svx.vecalign.vecalign import make_alignment_types, print_alignments
    v0, v1 = load_side("en", "src"), load_side("de", "tgt")
    assert v0.shape == (5, 237, 1024) and v1.shape == (5, 217, 1024)
    np.random.seed(seed)
    st = orc.vecalign(v0, v1, make_alignment_types(6), 0.2, 8, 300, 20000, 100)
    assert len(st) == 1  # 237 * 217 <= 300^2: no pyramid; 237 * 217 >= 20000: sampled knob branch
    out = tmp_path / "o.txt"
    with open(out, "w") as fp:
        print_alignments(st[0]['final_alignments'], scores=st[0]['alignment_scores'], ofile=fp)
    check_against_files(parse(out), seed)
    check_quality(st[0]['final_alignments'])


def build_tree(root, stopes):
    for lang in ("en", "de"):
        for sub in ("seg", "cat", "emb"):
            os.makedirs(os.path.join(root, sub, lang), exist_ok=True)
        shutil.copy(os.path.join(FULL, f"segments_{lang}.txt"), os.path.join(root, "seg", lang, f"doc_{lang}.txt"))
        shutil.copy(os.path.join(FULL, f"cat_segs_{lang}.txt"), os.path.join(root, "cat", lang, f"doc_{lang}.txt"))
        dst = os.path.join(root, "emb", lang, f"doc_{lang}.embed")
        if stopes:  # the shipped files are stopes Embedding files = .npy v1.0
            arr = np.fromfile(os.path.join(FULL, f"embeds_{lang}.f16"), dtype=np.float16).reshape(-1, 1024)
            np.save(dst + ".npy", arr)
            os.rename(dst + ".npy", dst)
        else:
            shutil.copy(os.path.join(FULL, f"embeds_{lang}.f16"), dst)
    os.makedirs(os.path.join(root, "ign", "en-de"), exist_ok=True)
    for side in ("src", "tgt"):
        shutil.copy(os.path.join(FULL, f"ignore_{side}.txt"), os.path.join(root, "ign", "en-de", f"doc_en-doc_de.{side}.txt"))
    with open(os.path.join(root, "metadata.tsv"), "w") as f:
        f.write("/audio/doc_en.ogg\t/audio/doc_de.ogg\n")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,stopes", [(0, True), (1, False), (42, False)])
def test_gpu_cli_reproduces_shipped_alignment(tmp_path, seed, stopes):
    """README.md:262-275 command line on the shipped files, through the HIP path."""
    from svx.seg_align import align as A
    from svx.utils.file_utils import read_alignments
    root, out = str(tmp_path / "data"), str(tmp_path / "out")
    build_tree(root, stopes)
    np.random.seed(seed)
    A.main([os.path.join(root, "metadata.tsv"), out, "--src_lang", "en", "--tgt_lang", "de", "--seg_dir", os.path.join(root, "seg"),
            "--concat_dir", os.path.join(root, "cat"), "--embed_dir", os.path.join(root, "emb"),
            "--ign_indices_dir", os.path.join(root, "ign"), "--is_stopes_embed" if stopes else "--fp16_embed"])
    res = os.path.join(out, "en-de", "doc_en-doc_de.txt")
    check_against_files(parse(res), seed)
    check_quality(read_alignments(res))

#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (runs only in the build container, where
/root/reference exists; see oracle/ref_loader.py).  Inputs are NOT stored: they are regenerated
from seeds by tests/synth.py.  Stored: outputs of the reference's own functions.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_loader  # noqa: E402
from synth import alignment_types, make_pair, round_bf16  # noqa: E402
from cases import OPS_CASE, PIPELINE_CASES, EXAMPLE_TRIM, pipeline_inputs  # noqa: E402


def rows_of(alignments):
    rows = np.zeros((len(alignments), 4), dtype=np.int32)
    for i, (x, y) in enumerate(alignments):
        rows[i] = (x[0] if len(x) else 0, len(x), y[0] if len(y) else 0, len(y))
    return rows


def gen_ops(ref):
    c = OPS_CASE
    v0, v1 = make_pair(c["N"], c["M"], c["K"], c["d"], c["seed"])
    R, C = ref.dp_utils, ref.dp_core
    a, b = v0.copy(), v1.copy()
    R.make_norm1(a)
    R.make_norm1(b)
    out = {"norm_a_row0": a[:, 0, :].copy(), "norm_b_last": b[:, -1, :].copy()}
    half = R.downsample_vectors(a)
    out["half"] = half
    np.random.seed(c["norm_seed"])
    n0 = R.compute_norms(a, b, 100)
    n1 = R.compute_norms(b, a, 100)
    out["n0"], out["n1"] = n0, n1
    out["dense_costs_1_2"] = C.make_dense_costs(a, b, n0, n1, 1, 2)
    costs = C.make_dense_costs(a, b, n0, n1)
    out["dense_costs"] = costs
    csum, bp = C.dense_dp(costs, c["dense_pen"])
    out["dense_csum"], out["dense_bp"] = csum, bp.astype(np.int8)
    al = R.dense_traceback(bp)
    out["dense_align"] = rows_of(al)
    path = R.alignment_to_search_path(al)
    out["path"] = np.array(path, dtype=np.int32)
    types = alignment_types(c["a"])
    feats, boff = C.make_sparse_costs(a, b, n0, n1, path, types, c["W"])
    out["sparse_costs"], out["b_offset"] = feats, boff
    scsum, xp, yp, bout = C.sparse_dp(feats, boff, types, c["sparse_pen"], c["N"], c["M"])
    out["sparse_csum"], out["sparse_xp"], out["sparse_yp"], out["b_offset_out"] = scsum, xp.astype(np.int8), yp.astype(np.int8), bout
    al2, sc2 = R.sparse_traceback(scsum, xp, yp, bout, c["N"], c["M"])
    out["sparse_align"], out["sparse_scores"] = rows_of(al2), np.asarray(sc2, dtype=np.float64)
    # coarse path of a half-size problem, up-sampled (extend quirk: index == size is appended)
    up = R.upsample_alignment(al)
    R.extend_alignments(up, 2 * c["N"] + 1, 2 * c["M"])
    out["path_up"] = np.array(R.alignment_to_search_path(up), dtype=np.int32)
    xs = np.random.RandomState(1).randint(0, c["N"], 5000).astype(np.int32)
    ys = np.random.RandomState(2).randint(0, c["M"], 5000).astype(np.int32)
    sc = np.empty(5000, np.float32)
    C.score_path(xs, ys, n0[0], n1[0], a[0], b[0], sc)
    out["score_path"] = sc
    knob = R.DeletionKnob(sc, 0, max(sc))
    out["del_pen"] = np.array([knob.percentile_frac_to_del_penalty(f) for f in (0.05, 0.2, 0.25, 0.5, 0.9)], dtype=np.float64)
    # edge cases
    out["dense_bp_zero_2x3"] = C.dense_dp(np.zeros((2, 3), np.float32), 0.0)[1].astype(np.int8)
    e_cs, e_xp, e_yp, e_bo = C.sparse_dp(np.zeros((0,) + feats.shape[1:], np.float32), boff, [], 0.5, c["N"], c["M"])
    out["empty_types_csum"], out["empty_types_xp"] = e_cs, e_xp.astype(np.int8)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)
    print("ops.npz:", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


def gen_pipeline(ref):
    out = {}
    for name, c in PIPELINE_CASES.items():
        v0, v1, types, W, kw = pipeline_inputs(c)
        np.random.seed(c["rng_seed"])
        stack = ref.dp_utils.vecalign(v0.copy(), v1.copy(), types, c.get("frac", 0.2), W, c.get("max_full", 300),
                                      c.get("sample", 20000), c.get("nsamp", 100))
        out[name + "/align"] = rows_of(stack[0]['final_alignments'])
        out[name + "/scores"] = np.asarray(stack[0]['alignment_scores'], dtype=np.float64)
        out[name + "/del_pen"] = np.array([stack[d]['del_penalty'] for d in sorted(stack)], dtype=np.float64)
        out[name + "/n0_l0"] = stack[0]['n0'].astype(np.float32)
        out[name + "/searchpath_sum"] = np.array([np.array(stack[0]['searchpath'], dtype=np.int64)[:, 1].sum()], dtype=np.int64)
        print(name, "levels", len(stack), "alignments", len(stack[0]['final_alignments']))
    np.savez_compressed(os.path.join(HERE, "pipeline.npz"), **out)


def gen_example(ref):
    """A trimmed prefix of the reference's shipped example (data files, not code): the first
    segments of example/voxpopuli with their candidate lines and fp16 embeddings, plus the
    reference's alignment of that prefix under np.random.seed(0)."""
    ex = os.path.join(ref.root, "example", "voxpopuli")
    stem = "20180313-0900-PLENARY-15"
    od = os.path.join(HERE, "example_trim")
    os.makedirs(od, exist_ok=True)
    nseg = {"en": EXAMPLE_TRIM["n_src"], "de": EXAMPLE_TRIM["n_tgt"]}
    for lang in ("en", "de"):
        segs = open(os.path.join(ex, "segments", lang, f"{stem}_{lang}.txt")).read().splitlines()[:nseg[lang]]
        last_end = int(segs[-1].split()[1])
        first_start = int(segs[0].split()[0])
        cats = open(os.path.join(ex, "cat_segs", lang, f"{stem}_{lang}.txt")).read().splitlines()
        emb = np.load(os.path.join(ex, "embeds", lang, f"{stem}_{lang}.embed"), allow_pickle=False)
        starts = {s.split()[0] for s in segs}
        ends = {s.split()[1] for s in segs}
        keep = [i for i, l in enumerate(cats) if l.split()[0] in starts and l.split()[1] in ends
                and first_start <= int(l.split()[0]) and int(l.split()[1]) <= last_end]
        with open(os.path.join(od, f"segments_{lang}.txt"), "w") as f:
            f.write("\n".join(segs) + "\n")
        with open(os.path.join(od, f"cat_segs_{lang}.txt"), "w") as f:
            f.write("\n".join(cats[i] for i in keep) + "\n")
        np.ascontiguousarray(emb[keep]).astype(np.float16).tofile(os.path.join(od, f"embeds_{lang}.f16"))
    for side, lang in (("src", "en"), ("tgt", "de")):
        ign = open(os.path.join(ex, "untrans_cat_seg_ids", "en-de", f"{stem}_en-{stem}_de.{side}.txt")).read().splitlines()
        with open(os.path.join(od, f"ignore_{side}.txt"), "w") as f:
            for l in ign:
                i, j = l.split()
                if int(i) < nseg[lang] and int(j) < nseg[lang]:
                    f.write(l + "\n")
    np.random.seed(0)
    out_txt = os.path.join(od, "expected_seed0.txt")
    ref.vecalign.align(src=os.path.join(od, "segments_en.txt"), tgt=os.path.join(od, "segments_de.txt"),
                       src_embed=[os.path.join(od, "cat_segs_en.txt"), os.path.join(od, "embeds_en.f16")], src_stopes=False,
                       src_fp16=True, tgt_embed=[os.path.join(od, "cat_segs_de.txt"), os.path.join(od, "embeds_de.f16")],
                       tgt_stopes=False, tgt_fp16=True, alignment_max_size=6, many_to_one=None, search_buffer_size=5,
                       del_percentile_frac=0.2, max_size_full_dp=300, costs_sample_size=20000, num_samps_for_norm=100,
                       overlap_segments=True, print_aligned_text=False, print_results=True, save_aligned_text_to_file=out_txt,
                       src_ignore_indices=os.path.join(od, "ignore_src.txt"), tgt_ignore_indices=os.path.join(od, "ignore_tgt.txt"))
    print("example_trim:", sum(os.path.getsize(os.path.join(od, f)) for f in os.listdir(od)) // 1024, "KiB")


def gen_example_full(ref):
    """The reference's whole shipped example as data (tests/golden/example_full): segments, candidate lines,
    ignore files, the two .embed payloads as raw fp16 (the shipped files are .npy v1.0 '<f2'), the shipped
    156-line alignment file and the gold alignment -- plus the REAL reference's own output on these files
    for np.random.seed(0 / 1 / 42) (its scores depend on the unseeded global stream; its spans do not)."""
    import shutil
    ex = os.path.join(ref.root, "example", "voxpopuli")
    stem = "20180313-0900-PLENARY-15"
    od = os.path.join(HERE, "example_full")
    os.makedirs(od, exist_ok=True)
    for lang in ("en", "de"):
        shutil.copyfile(os.path.join(ex, "segments", lang, f"{stem}_{lang}.txt"), os.path.join(od, f"segments_{lang}.txt"))
        shutil.copyfile(os.path.join(ex, "cat_segs", lang, f"{stem}_{lang}.txt"), os.path.join(od, f"cat_segs_{lang}.txt"))
        emb = np.load(os.path.join(ex, "embeds", lang, f"{stem}_{lang}.embed"), allow_pickle=False)
        assert emb.dtype == np.float16 and emb.shape[1] == 1024
        np.ascontiguousarray(emb).tofile(os.path.join(od, f"embeds_{lang}.f16"))
    for side in ("src", "tgt"):
        shutil.copyfile(os.path.join(ex, "untrans_cat_seg_ids", "en-de", f"{stem}_en-{stem}_de.{side}.txt"),
                        os.path.join(od, f"ignore_{side}.txt"))
    shutil.copyfile(os.path.join(ex, "alignments", "en-de", f"{stem}_en-{stem}_de.txt"), os.path.join(od, "shipped_alignment.txt"))
    shutil.copyfile(os.path.join(ex, f"{stem}.gold"), os.path.join(od, "gold.txt"))
    for f in os.listdir(od):
        os.chmod(os.path.join(od, f), 0o644)
    for seed in (0, 1, 42):
        np.random.seed(seed)
        ref.vecalign.align(src=os.path.join(od, "segments_en.txt"), tgt=os.path.join(od, "segments_de.txt"),
                           src_embed=[os.path.join(od, "cat_segs_en.txt"), os.path.join(od, "embeds_en.f16")], src_stopes=False,
                           src_fp16=True, tgt_embed=[os.path.join(od, "cat_segs_de.txt"), os.path.join(od, "embeds_de.f16")],
                           tgt_stopes=False, tgt_fp16=True, alignment_max_size=6, many_to_one=None, search_buffer_size=5,
                           del_percentile_frac=0.2, max_size_full_dp=300, costs_sample_size=20000, num_samps_for_norm=100,
                           overlap_segments=True, print_aligned_text=False, print_results=True,
                           save_aligned_text_to_file=os.path.join(od, "expected_seed%d.txt" % seed),
                           src_ignore_indices=os.path.join(od, "ignore_src.txt"), tgt_ignore_indices=os.path.join(od, "ignore_tgt.txt"))
    print("example_full:", sum(os.path.getsize(os.path.join(od, f)) for f in os.listdir(od)) // 1024, "KiB")


def gen_margin(ref_root):
    """The reference's shipped margin-scoring example as data: the rows of its two populated Flat indexes
    (exactly fp16-representable, stored as fp16) and the third field of its margin file.  The aligned
    embeddings of the example ARE the index rows, in file order (one document pair in the corpus)."""
    import struct
    ex = os.path.join(ref_root, "example", "voxpopuli")
    stem = "20180313-0900-PLENARY-15"

    def flat_rows(path):
        b = open(path, "rb").read()
        assert b[:4] == b"IxF2"
        d, ntotal = struct.unpack("<iq", b[4:16])
        rows = np.frombuffer(b, dtype="<f4", count=ntotal * d, offset=45).reshape(ntotal, d)
        assert np.array_equal(rows.astype(np.float16).astype(np.float32), rows)
        return rows.astype(np.float16)
    idx = os.path.join(ex, "align_0.7_clean_cat3_min1s_embed_indexes", "en-de")
    lines = open(os.path.join(ex, "align_0.7_clean_cat3_min1s_margin", "en-de", f"{stem}_en-{stem}_de.txt")).read().splitlines()
    out = {"db_src": flat_rows(os.path.join(idx, "en", "Flat.populate.idx")),
           "db_tgt": flat_rows(os.path.join(idx, "de", "Flat.populate.idx")),
           "expected": np.array([float(l.split(":")[2]) for l in lines], dtype=np.float32)}
    assert out["db_src"].shape[0] == out["db_tgt"].shape[0] == out["expected"].shape[0]
    np.savez_compressed(os.path.join(HERE, "margin_example.npz"), **out)
    print("margin_example.npz:", os.path.getsize(os.path.join(HERE, "margin_example.npz")) // 1024, "KiB")


def gen_post(ref_root):
    """Data files of the reference's shipped example for the manifest steps after margin scoring (copied as
    they are): segment files, the margin-scored alignment file, and the three manifests made from it."""
    import shutil
    ex = os.path.join(ref_root, "example", "voxpopuli")
    stem = "20180313-0900-PLENARY-15"
    od = os.path.join(HERE, "example_post")
    os.makedirs(od, exist_ok=True)
    for src, dst in ((f"segments/en/{stem}_en.txt", "segments_en.txt"), (f"segments/de/{stem}_de.txt", "segments_de.txt"),
                     ("metadata.tsv", "metadata.tsv"),
                     (f"align_0.7_clean_cat3_min1s_margin/en-de/{stem}_en-{stem}_de.txt", "margin.txt"),
                     ("align_0.7_clean_cat3_min1s_tsvs/en-de/align.tsv.gz", "align.tsv.gz"),
                     ("align_0.7_clean_cat3_min1s_tsvs/en-de/align.rm_overlap.tsv.gz", "align.rm_overlap.tsv.gz"),
                     ("align_0.7_clean_cat3_min1s_tsvs/en-de/align.rm_overlap.sort.tsv.gz", "align.rm_overlap.sort.tsv.gz")):
        shutil.copyfile(os.path.join(ex, src), os.path.join(od, dst))
        os.chmod(os.path.join(od, dst), 0o644)
    print("example_post:", sum(os.path.getsize(os.path.join(od, f)) for f in os.listdir(od)) // 1024, "KiB")


if __name__ == "__main__":
    what = sys.argv[1:] or ["ops", "pipeline", "example", "example_full", "margin", "post"]
    if "post" in what:
        gen_post(ref_loader.REF_ROOT)
    if "margin" in what:
        gen_margin(ref_loader.REF_ROOT)
    if set(what) - {"margin", "post"}:
        ref = ref_loader.load()
        if "ops" in what:
            gen_ops(ref)
        if "pipeline" in what:
            gen_pipeline(ref)
        if "example" in what:
            gen_example(ref)
        if "example_full" in what:
            gen_example_full(ref)

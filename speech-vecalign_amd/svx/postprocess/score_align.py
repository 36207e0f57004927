"""python -m svx.postprocess.score_align METADATA OUT_DIR --embed_dir E --align_dir A --index_dir I --src_lang en --tgt_lang de
(reference: svecalign/postprocess/score_align.py)

Margin scores (Artetxe & Schwenk, https://aclanthology.org/P19-1309) of the mined alignments against the
whole corpus: score = cos(x, y) / mean of the two k-NN mean similarities ("ratio").  Same CLI, file
layout and output lines as the reference; the k-NN searches run as svx_knn_mean_sim over exact fp16
databases in HBM (flat_index.py), all embedding files of a rank in one launch per direction.

More than one process (torch.distributed, one per GPU): the document pairs are split into contiguous
shards, every rank holds the whole database (read from the index files, or all-gathered from the
ranks' own rows with `global_margin_scores`), and each rank writes the files of its shard -- no
collective on the scoring path itself."""
import argparse
import ctypes
import logging
from pathlib import Path
from typing import List, Tuple

import numpy as np

from .. import _lib
from ..utils.file_utils import read_alignments, read_metadata
from .flat_index import FlatIndex, all_gather_rows, ring_shards, to_device_rows, _torch_dtype_code
from .prep_index import load_embed_from_tsv

logger = logging.getLogger(__name__)
MARGINS = {"ratio": _lib.SVX_MARGIN_RATIO, "distance": _lib.SVX_MARGIN_DISTANCE}


def find_valid_metas(meta: List[Tuple[str, str]], embed_dir: Path) -> List[str]:
    """ids of the pairs whose two embedding tsvs exist (score_align.py:73-94)."""
    res = []
    for src_aud, tgt_aud in meta:
        stem = f"{Path(src_aud).stem}-{Path(tgt_aud).stem}"
        src_tsv, tgt_tsv = embed_dir / f"{stem}.src.tsv", embed_dir / f"{stem}.tgt.tsv"
        if src_tsv.exists() and tgt_tsv.exists():
            res.append(stem)
        elif not src_tsv.exists() and not tgt_tsv.exists():
            logger.warning(f"{src_tsv} and {tgt_tsv} not exist")
        else:
            raise Exception(f"{src_tsv}: {src_tsv.exists()} | {tgt_tsv}: {tgt_tsv.exists()}")
    logger.info(f"Kept {len(res)}/{len(meta)}")
    return res


def write_to_output(align_dir: Path, align_ids: List[str], margin_scores: np.ndarray, out_dir: Path):
    """`src:tgt:margin` lines, scores consumed in file order (score_align.py:97-116)."""
    at = 0
    for ali_id in align_ids:
        alignments = read_alignments(align_dir / f"{ali_id}.txt")
        with open(out_dir / f"{ali_id}.txt", mode="w") as fp:
            for src, tgt in alignments:
                fp.write(f"{src}:{tgt}:{margin_scores[at]}\n")
                at += 1
    assert at == margin_scores.shape[0], f"{at}, {margin_scores.shape}"


def margin_scores_device(idx_x: FlatIndex, idx_y: FlatIndex, x, y, k: int, margin: str):
    """-> device float32 [n].  x, y: [n, d] numpy / torch rows in any of fp32 / fp16 / bf16 (not modified)."""
    if margin not in MARGINS:
        raise ValueError(f"Wrong margin type: {margin}")
    ctx = idx_y.ctx
    xd, yd = to_device_rows(ctx, x), to_device_rows(ctx, y)
    assert xd.shape == yd.shape, f"{tuple(xd.shape)} {tuple(yd.shape)}"
    if yd.dtype != xd.dtype:
        yd = yd.to(xd.dtype)
    mean_xy = idx_y.mean_sim(xd, k)   # x against its neighbours among the targets
    mean_yx = idx_x.mean_sim(yd, k)   # y against its neighbours among the sources
    out = ctx.torch.empty((xd.shape[0],), dtype=ctx.torch.float32, device=ctx.tdev)
    ctx.check(ctx.lib.svx_margin_scores(ctx.h, ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(yd.data_ptr()),
                                        _torch_dtype_code(ctx.torch, xd.dtype), int(xd.shape[0]), int(xd.shape[1]),
                                        ctypes.c_void_p(mean_xy.data_ptr()), ctypes.c_void_p(mean_yx.data_ptr()),
                                        MARGINS[margin], ctypes.c_void_p(out.data_ptr())))
    return out


def compute_sim_with_nonflat_idx(idx_x: FlatIndex, idx_y: FlatIndex, x: np.ndarray, y: np.ndarray, k: int,
                                 margin: str) -> np.ndarray:
    """The reference's entry point (score_align.py:124-161) -> float32 [n] on the host."""
    num_x, dim_x = x.shape
    num_y, dim_y = y.shape
    assert num_x == num_y and dim_x == dim_y, f"{x.shape} {y.shape}"
    return margin_scores_device(idx_x, idx_y, x, y, k, margin).cpu().numpy()


def global_margin_scores(x_local, y_local, k: int = 16, margin: str = "ratio", storage: str = "fp16", group=None,
                         device=None, exchange: str = "allgather"):
    """Each rank passes the embeddings of ITS alignments; the databases are the union over ranks, the scores come
    back for the local rows only.  exchange="allgather" (default): one RCCL all-gather, then a local search of the
    whole corpus.  exchange="ring": the normalised fp16 shards travel round the ring of ranks and every rank merges
    its queries' top-k shard by shard (`ring_shards`, svx_knn_topk_merge) -- two shards of HBM per GPU, transfer
    overlapped with search; it stays opt-in until a multi-GPU run of tests/test_gpu_margin.py::
    test_global_margin_over_rccl is on record (the one-GPU box can only run it with a world of one).  Both give the
    k nearest neighbours over the whole corpus."""
    if exchange not in ("ring", "allgather"):
        raise ValueError(f"exchange {exchange!r}: 'ring' or 'allgather'")
    if margin not in MARGINS:
        raise ValueError(f"Wrong margin type: {margin}")
    ctx = _lib.context(device)
    t = ctx.torch
    d = int(x_local.shape[1])
    idx_x, idx_y = FlatIndex(d, storage, device), FlatIndex(d, storage, device)
    idx_x.add(x_local)
    idx_y.add(y_local)
    if exchange == "allgather":
        gx, gy = FlatIndex(d, storage, device), FlatIndex(d, storage, device)
        gx.add_unit_rows(all_gather_rows(idx_x.rows, group))
        gy.add_unit_rows(all_gather_rows(idx_y.rows, group))
        ctx.sync()
        return margin_scores_device(gx, gy, x_local, y_local, k, margin)
    xd, yd = to_device_rows(ctx, x_local), to_device_rows(ctx, y_local)
    assert xd.shape == yd.shape, f"{tuple(xd.shape)} {tuple(yd.shape)}"
    if yd.dtype != xd.dtype:
        yd = yd.to(xd.dtype)
    means, totals = [], []
    for queries, index in ((xd, idx_y), (yd, idx_x)):  # x among the targets, y among the sources
        topk, mean, total = None, None, 0
        ctx.sync()  # the shard rows are complete before the communication stream reads them
        for _, rows in ring_shards(index.rows, group):
            shard = FlatIndex(d, storage, device)
            shard.add_unit_rows(rows)
            topk, mean = shard.merge_topk(queries, k, topk, want_mean=True)
            total += int(rows.shape[0])
            ctx.sync()  # the search has read the shard before its buffer receives the next one
        means.append(mean)
        totals.append(total)
    if min(totals) < k:
        raise ValueError(f"the corpus has {min(totals)} rows on one side, fewer than k = {k}")
    out = t.empty((xd.shape[0],), dtype=t.float32, device=ctx.tdev)
    ctx.check(ctx.lib.svx_margin_scores(ctx.h, ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(yd.data_ptr()),
                                        _torch_dtype_code(t, xd.dtype), int(xd.shape[0]), int(xd.shape[1]),
                                        ctypes.c_void_p(means[0].data_ptr()), ctypes.c_void_p(means[1].data_ptr()),
                                        MARGINS[margin], ctypes.c_void_p(out.data_ptr())))
    return out


def _dist_rank_world():
    import os
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
        return dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", "0"))
    return 0, 1, 0


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("metadata", type=str, help="the meta file that each line contains paired audio paths")
    p.add_argument("out_dir", type=str, help="dir to store the margin-scored alignments.")
    p.add_argument("--embed_dir", type=str, required=True, help="the dir for embedding tsvs.")
    p.add_argument("--align_dir", type=str, required=True, help="the dir for concatenated alignments.")
    p.add_argument("--src_lang", type=str, required=True)
    p.add_argument("--tgt_lang", type=str, required=True)
    p.add_argument("--index_dir", type=str, required=True, help="where the indexes are saved.")
    p.add_argument("--num_probe", type=int, default=128, help="accepted for compatibility: the search is exact.")
    p.add_argument("--gpu_type", type=str, default="fp16-shard", help="fp16* keeps the database in fp16, bf16* in bf16.")
    p.add_argument("--embed_fp16", action="store_true", default=False, help="whether the embeddings are saved in fp16.")
    p.add_argument("--embed_stopes", action="store_true", default=False, help="whether the input embeddings are saved with stopes.")
    p.add_argument("--margin", type=str, default="ratio", help="Margin for xSIM calculation. See: https://aclanthology.org/P19-1309")
    p.add_argument("--k", type=int, default=16, help="number of nearest number.")
    a = p.parse_args(argv)
    logger.info(a)

    rank, world, local_rank = _dist_rank_world()
    src_lang, tgt_lang = a.src_lang, a.tgt_lang
    embed_dir = Path(a.embed_dir) / f"{src_lang}-{tgt_lang}"
    align_dir = Path(a.align_dir) / f"{src_lang}-{tgt_lang}"
    out_dir = Path(a.out_dir) / f"{src_lang}-{tgt_lang}"
    out_dir.mkdir(parents=True, exist_ok=True)

    metas = find_valid_metas(read_metadata(a.metadata), embed_dir)
    metas = metas[len(metas) * rank // world:len(metas) * (rank + 1) // world]  # (a rank may get none)

    index_dir = Path(a.index_dir) / f"{src_lang}-{tgt_lang}"
    storage = "bf16" if a.gpu_type.startswith("bf16") else "fp16"
    src_index_path = list((index_dir / src_lang).glob("*.populate.idx"))[0]
    tgt_index_path = list((index_dir / tgt_lang).glob("*.populate.idx"))[0]
    logger.info(f"Loading {src_index_path} and {tgt_index_path} ({storage})")
    src_index = FlatIndex.read(src_index_path, storage=storage, device=local_rank)
    tgt_index = FlatIndex.read(tgt_index_path, storage=storage, device=local_rank)

    # every file of this shard goes through ONE search per direction
    xs = [load_embed_from_tsv(embed_dir / f"{m}.src.tsv", fp16_embed=a.embed_fp16, use_stopes=a.embed_stopes) for m in metas]
    ys = [load_embed_from_tsv(embed_dir / f"{m}.tgt.tsv", fp16_embed=a.embed_fp16, use_stopes=a.embed_stopes) for m in metas]
    if metas:
        scores = compute_sim_with_nonflat_idx(src_index, tgt_index, np.concatenate(xs, axis=0), np.concatenate(ys, axis=0),
                                              a.k, a.margin)
    else:
        scores = np.zeros((0,), dtype=np.float32)
    logger.info(f"Writing to {out_dir}...")
    write_to_output(align_dir, metas, scores, out_dir)
    logger.info("Done!")


if __name__ == '__main__':
    main()

"""python -m svx.postprocess.prep_index METADATA OUT_DIR --data_dir D --src_lang en --tgt_lang de [--use_tgt]
(reference: svecalign/postprocess/prep_index.py)

Builds the database that score_align searches.  The reference samples half of the embedding files, trains
a faiss index whose type depends on the corpus size and adds every embedding to it (prep_index.py:197-249);
here the index is always exact ("Flat"), so there is nothing to train: the rows are normalised on the
device, kept in fp16 and written as `Flat.populate.idx` in faiss' own IndexFlat format (the name and layout
score_align looks for, score_align.py:185-189)."""
import argparse
import logging
from collections import defaultdict
from pathlib import Path
from typing import List, Tuple

import numpy as np

from ..utils.embedding_utils import load_sent_embeddings
from ..utils.file_utils import read_metadata
from .flat_index import FlatIndex

logger = logging.getLogger(__name__)
INDEX_TYPE = "Flat"


def find_embed_files(meta: List[Tuple[str, str]], data_dir: Path, use_tgt: bool) -> List[Path]:
    """{src_id}-{tgt_id}.{src|tgt}.tsv for every metadata pair that has both (prep_index.py:65-88)."""
    res = []
    for src_aud, tgt_aud in meta:
        stem = f"{Path(src_aud).stem}-{Path(tgt_aud).stem}"
        src_tsv, tgt_tsv = data_dir / f"{stem}.src.tsv", data_dir / f"{stem}.tgt.tsv"
        if src_tsv.exists() and tgt_tsv.exists():
            res.append(tgt_tsv if use_tgt else src_tsv)
        elif not src_tsv.exists() and not tgt_tsv.exists():
            logger.warning(f"{src_tsv} and {tgt_tsv} do not exist")
        else:
            raise Exception(f"{src_tsv}: {src_tsv.exists()} | {tgt_tsv}: {tgt_tsv.exists()}")
    logger.info(f"Kept {len(res)}/{len(meta)} files")
    return res


def load_embed_from_tsv(tsv_path, fp16_embed: bool, use_stopes: bool) -> np.ndarray:
    """Rows named by a tsv of `embedding_file<TAB>row` lines, in line order (prep_index.py:91-127).
    Each embedding file is opened once; the result keeps the files' storage type."""
    by_file = defaultdict(list)
    n = 0
    with open(tsv_path) as fp:
        for n, line in enumerate(fp, 1):
            path, row = line.strip().split("\t")
            by_file[path].append((n - 1, int(row)))
    out = None
    for path, wanted in by_file.items():
        emb = load_sent_embeddings(path, fp16_embed=fp16_embed, use_stopes=use_stopes, stopes_mode="memory")
        if out is None:
            out = np.empty((n, emb.shape[1]), dtype=emb.dtype)
        at, rows = zip(*wanted)
        out[list(at)] = emb[list(rows)]
    if out is None:
        raise ValueError(f"{tsv_path}: empty")  # np.stack([]) of the reference raises ValueError too
    return out


def populate_index(embed_paths: List[Path], out_path: Path, fp16_embed: bool, use_stopes: bool,
                   storage: str = "fp16", device=None) -> FlatIndex:
    """normalize_L2 + add for every embedding file, then write the index (prep_index.py:153-185)."""
    index = None
    for path in embed_paths:
        embed = load_embed_from_tsv(path, fp16_embed=fp16_embed, use_stopes=use_stopes)
        if index is None:
            index = FlatIndex(d=embed.shape[1], storage=storage, device=device)
        index.add(embed)
    if index is None:
        raise Exception("no embedding files to index")
    index.write(out_path)
    return index


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("metadata", type=str, help="the meta file that each line contains paired audio paths")
    p.add_argument("out_dir", type=str, help="dir to store the indices.")
    p.add_argument("--data_dir", type=str, required=True, help="the dir for embedding tsvs.")
    p.add_argument("--use_tgt", action="store_true", default=False)
    p.add_argument("--sample_ratio", type=float, default=0.5, help="accepted for compatibility: a Flat index is not trained.")
    p.add_argument("--embed_fp16", action="store_true", default=False, help="whether the embeddings are saved in fp16.")
    p.add_argument("--embed_stopes", action="store_true", default=False, help="whether the input embeddings are saved with stopes.")
    p.add_argument("--src_lang", type=str, required=True)
    p.add_argument("--tgt_lang", type=str, required=True)
    a = p.parse_args(argv)
    logger.info(a)
    data_dir = Path(a.data_dir) / f"{a.src_lang}-{a.tgt_lang}"
    out_dir = Path(a.out_dir) / f"{a.src_lang}-{a.tgt_lang}" / (a.tgt_lang if a.use_tgt else a.src_lang)
    out_dir.mkdir(parents=True, exist_ok=True)
    embed_paths = find_embed_files(read_metadata(a.metadata), data_dir, a.use_tgt)
    index = populate_index(embed_paths, out_dir / f"{INDEX_TYPE}.populate.idx", a.embed_fp16, a.embed_stopes)
    logger.info(f"#embeddings: {index.ntotal}")
    logger.info("Finished!")


if __name__ == '__main__':
    main()

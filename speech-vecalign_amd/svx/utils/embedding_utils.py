"""Embedding files and the candidate ("overlap") tensor of a document.

Reference: svecalign/utils/embedding_utils.py -- readers :38-103, make_overlap :106-132,
make_doc_embedding :135-203.  Same file formats and the same tensor layout
    V[j, i, :] = embedding of the concatenation of segments i-j .. i   (j+1 segments ENDING at i)
with all-zero rows for PAD / ignored / missing / NaN candidates.  What differs is how it is built:
the host only resolves candidate strings to row numbers (an int32 [K, N] table, -1 = zero row);
the gather itself runs on the device and keeps the file's storage type (fp16 stays fp16 -- the
kernels widen exactly, the reference up-casts on the host, embedding_utils.py:43,50).
"""
import logging
from typing import List, Optional, Set, Tuple

import numpy as np

EMBED_DIM = 1024
PAD_LABEL = "PAD"

logger = logging.getLogger(__name__)


def preprocess_line(line):
    line = line.strip()
    if len(line) == 0:
        logger.warning("Encountered empty line.")
        line = '[BLANK_LINE]'
    return line


def load_stopes_embeddings(path: str, mode: str = "mmap") -> np.ndarray:
    """stopes' Embedding file is a NumPy .npy v1.0 file (128-byte header, '<f2', [n, 1024]);
    read it with numpy directly -- no stopes needed."""
    return np.load(path, mmap_mode="r" if mode == "mmap" else None, allow_pickle=False)


def load_np_embeddings(embed_file: str, fp16_embed: bool) -> np.ndarray:
    """Raw headerless fp16 / fp32 matrix (embedding_utils.py:47-54), memory-mapped."""
    return np.memmap(embed_file, dtype=np.float16 if fp16_embed else np.float32, mode="r")


def load_sent_embeddings(embed_file: str, use_stopes: bool = False, fp16_embed: bool = False,
                         stopes_mode: str = "mmap") -> np.ndarray:
    """-> [n_candidates, 1024] in the file's storage type."""
    if use_stopes:
        emb = load_stopes_embeddings(embed_file, mode=stopes_mode)
        if emb.ndim != 2:
            raise Exception('stopes embedding file %s is not 2-D' % embed_file)
        return emb
    emb = load_np_embeddings(embed_file, fp16_embed)
    if emb.size == 0:
        raise Exception('Got empty embedding file')
    return emb[:(emb.shape[0] // EMBED_DIM) * EMBED_DIM].reshape(-1, EMBED_DIM)


def read_in_embeddings(text_file: str, embed_file: str, use_stopes: bool = False,
                       fp16_embed: bool = False) -> Tuple[dict, np.ndarray]:
    """candidate line -> first row index (duplicates keep the first), and the embedding matrix."""
    sent2line = dict()
    with open(text_file, 'rt', encoding="utf-8") as fin:
        for i, line in enumerate(fin):
            sent2line.setdefault(line.strip(), i)
    return sent2line, load_sent_embeddings(embed_file, use_stopes, fp16_embed)


def make_overlap(lines: List[str], num_overlaps: int, start_id: int,
                 ignore_indices: Optional[Set[Tuple[int, int]]] = None, comb: str = ' ',
                 overlap_segments: bool = False) -> List[str]:
    """Candidate strings starting at segment `start_id` (embedding_utils.py:106-132)."""
    res = []
    stop = min(len(lines), start_id + num_overlaps)
    for j in range(start_id, stop):
        if ignore_indices and (start_id, j) in ignore_indices:
            res.extend([PAD_LABEL] * (stop - j))  # everything from (start_id, j) on is padded
            break
        if overlap_segments:
            res.append(f"{lines[start_id].split()[0]} {lines[j].split()[1]}")
        else:
            res.append(comb.join(lines[start_id:j + 1]))
    return res


def candidate_index_table(sent2id: dict, lines: List[str], max_overlaps: int,
                          ignore_indices: Optional[Set[Tuple[int, int]]] = None,
                          overlap_segments: bool = False, bad_rows: Optional[np.ndarray] = None) -> np.ndarray:
    """int32 [max_overlaps, len(lines)]: row of the embedding file for slot (j, end index i), -1 for a
    zero vector (slot i < j, PAD, ignored, candidate not embedded, or NaN row)."""
    lines = [preprocess_line(line) for line in lines]
    n = len(lines)
    table = np.full((max_overlaps, n), -1, dtype=np.int32)
    n_match = n_miss = 0
    for i in range(n):
        for j, cand in enumerate(make_overlap(lines, max_overlaps, i, ignore_indices, overlap_segments=overlap_segments)):
            row = -1 if cand == PAD_LABEL else sent2id.get(cand, -1)
            if row >= 0 and bad_rows is not None and bad_rows[row]:
                logger.error(f"loaded a vector with nan value at {row} with overlap {cand}. Will reset to zero.")
                row = -1
            table[j, i + j] = row
            n_match += row >= 0
            n_miss += row < 0
    logger.debug(f"Match: {n_match} || Miss: {n_miss}")
    return table


def gather_candidates(line_embeddings, table: np.ndarray, device=None):
    """Device gather (svx_gather_rows): out[j, i] = line_embeddings[table[j, i]] or 0 -> torch tensor [K, N, d]."""
    import ctypes
    from .. import _lib
    ctx = _lib.context(device)
    t = ctx.torch
    # (np.array copies: the file readers hand out read-only memory maps)
    emb = line_embeddings if hasattr(line_embeddings, "data_ptr") else t.from_numpy(np.array(line_embeddings))
    emb = emb.to(ctx.tdev).contiguous()
    dt = {t.float32: _lib.SVX_F32, t.float16: _lib.SVX_F16, t.bfloat16: _lib.SVX_BF16}[emb.dtype]
    idx = t.from_numpy(np.ascontiguousarray(table, dtype=np.int32)).to(ctx.tdev)
    out = t.empty(tuple(table.shape) + (emb.shape[1],), dtype=emb.dtype, device=ctx.tdev)
    ctx.check(ctx.lib.svx_gather_rows(ctx.h, ctypes.c_void_p(emb.data_ptr()), int(emb.shape[0]), int(emb.shape[1]), dt,
                                      ctypes.c_void_p(idx.data_ptr()), int(idx.numel()), ctypes.c_void_p(out.data_ptr())))
    return out


def make_doc_embedding(sent2id: dict, line_embeddings: np.ndarray, lines: List[str], max_overlaps: int,
                       ignore_indices: Optional[Set[Tuple[int, int]]] = None, overlap_segments: bool = False,
                       device=None):
    """-> device tensor [max_overlaps, len(lines), d] (embedding_utils.py:135-203)."""
    arr = np.asarray(line_embeddings)
    bad = np.isnan(arr).any(axis=1) if arr.size else None
    table = candidate_index_table(sent2id, lines, max_overlaps, ignore_indices, overlap_segments, bad)
    return gather_candidates(line_embeddings, table, device)

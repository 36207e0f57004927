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


def candidate_table_from_files(seg_file: str, cat_file: str, max_overlaps: int, ignore_file: Optional[str] = None):
    """candidate_index_table for speech segments straight from the files (svx_candidate_table, native and
    GIL-free: a thread pool can build the tables of many documents at once): -> (int32 [max_overlaps, n_lines],
    number of candidate lines).  String keys exactly as the reference builds them (embedding_utils.py:93-99,128);
    NaN rows are handled by the device gather."""
    import ctypes
    import os
    from .. import _lib
    lib = _lib.load()
    cap = os.path.getsize(seg_file) // 2 + 2  # a line has at least two bytes
    table = np.empty((max_overlaps, cap), dtype=np.int32)
    n_lines, n_cand = ctypes.c_int32(0), ctypes.c_int64(0)
    err = ctypes.create_string_buffer(512)
    ign = None if ignore_file is None else os.fspath(ignore_file).encode()
    rc = lib.svx_candidate_table(os.fspath(seg_file).encode(), os.fspath(cat_file).encode(), ign, int(max_overlaps),
                                 ctypes.c_void_p(table.ctypes.data), int(cap), ctypes.byref(n_lines), ctypes.byref(n_cand), err, 512)
    if rc != 0:
        raise Exception(err.value.decode(errors="replace") or "svx_candidate_table failed")
    return np.ascontiguousarray(table[:, :n_lines.value]), int(n_cand.value)


def embedding_file_layout(embed_file: str, use_stopes: bool = False, fp16_embed: bool = False):
    """-> (byte offset of the first row, number of rows, numpy dtype) of an .embed file: stopes files are NumPy
    .npy v1/v2 ('<f2' or '<f4', C order, [n, 1024]); the others are headerless fp16 / fp32 (embedding_utils.py:38-76)."""
    import os
    size = os.path.getsize(embed_file)
    if use_stopes:
        with open(embed_file, "rb") as f:
            major, _ = np.lib.format.read_magic(f)
            shape, fortran, dt = (np.lib.format.read_array_header_1_0(f) if major == 1 else np.lib.format.read_array_header_2_0(f))
            off = f.tell()
        if fortran or len(shape) != 2 or dt not in (np.dtype('<f2'), np.dtype('<f4')):
            raise Exception('stopes embedding file %s: unsupported layout %s %s' % (embed_file, shape, dt))
        return off, int(shape[0]), dt, int(shape[1])
    dt = np.dtype(np.float16 if fp16_embed else np.float32)
    if size == 0:
        raise Exception('Got empty embedding file')
    return 0, size // (EMBED_DIM * dt.itemsize), dt, EMBED_DIM


def read_embeddings_pinned(embed_file: str, use_stopes: bool = False, fp16_embed: bool = False):
    """The embedding matrix of a file read into page-locked host memory (one readinto, GIL released), as a
    torch tensor [n_rows, d] in the file's storage type, ready for an asynchronous upload."""
    import torch
    off, rows, dt, d = embedding_file_layout(embed_file, use_stopes, fp16_embed)
    tdt = torch.float16 if dt.itemsize == 2 else torch.float32
    host = torch.empty((rows, d), dtype=tdt, pin_memory=torch.cuda.is_available())
    view = host.numpy().reshape(-1).view(np.uint8)
    with open(embed_file, "rb", buffering=0) as f:
        f.seek(off)
        got = 0
        mv = memoryview(view)
        while got < view.size:
            n = f.readinto(mv[got:])
            if not n:
                raise Exception('%s is shorter than its header says' % embed_file)
            got += n
    return host


def gather_candidates(line_embeddings, table: np.ndarray, device=None):
    """Device gather (svx_gather_rows): out[j, i] = line_embeddings[table[j, i]] or 0 -> torch tensor [K, N, d]."""
    import ctypes
    from .. import _lib
    ctx = _lib.context(device)
    t = ctx.torch
    # (np.array copies: the file readers hand out read-only memory maps)
    emb = line_embeddings if hasattr(line_embeddings, "data_ptr") else t.from_numpy(np.array(line_embeddings))
    emb = emb.to(ctx.tdev, non_blocking=True).contiguous()
    dt = {t.float32: _lib.SVX_F32, t.float16: _lib.SVX_F16, t.bfloat16: _lib.SVX_BF16}[emb.dtype]
    if hasattr(table, "data_ptr"):
        idx = table.to(ctx.tdev, dtype=t.int32).contiguous()
    else:
        idx = t.from_numpy(np.ascontiguousarray(table, dtype=np.int32)).to(ctx.tdev)
    out = t.empty(tuple(table.shape) + (emb.shape[1],), dtype=emb.dtype, device=ctx.tdev)
    ctx.check(ctx.lib.svx_gather_rows(ctx.h, ctypes.c_void_p(emb.data_ptr()), int(emb.shape[0]), int(emb.shape[1]), dt,
                                      ctypes.c_void_p(idx.data_ptr()), int(idx.numel()), ctypes.c_void_p(out.data_ptr())))
    return out


def make_doc_embedding(sent2id: dict, line_embeddings: np.ndarray, lines: List[str], max_overlaps: int,
                       ignore_indices: Optional[Set[Tuple[int, int]]] = None, overlap_segments: bool = False,
                       device=None):
    """-> device tensor [max_overlaps, len(lines), d] (embedding_utils.py:135-203).  Rows of the embedding file that
    hold a NaN come out as zero rows like in the reference (:183-190); the device gather tests for them."""
    table = candidate_index_table(sent2id, lines, max_overlaps, ignore_indices, overlap_segments)
    return gather_candidates(line_embeddings, table, device)

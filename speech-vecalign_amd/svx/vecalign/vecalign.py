"""Single-pair API and standalone CLI, same surface as the reference's
svecalign/vecalign/vecalign.py (align :198-293, parse_args :36-151, make_alignment_types :154-162,
make_many_to_one_alignment_types :165-171, print_alignments :174-184, load_ignore_index_file
:187-195).  The alignment itself runs on the GPU (dp_utils.vecalign -> svx_align_batch).

    python -m svx.vecalign.vecalign -s SRC -t TGT --src_embed TXT BIN --tgt_embed TXT BIN [...]
"""
import argparse
import logging
import math
import pickle
import sys
from pathlib import Path
from typing import List, Optional, Set, Tuple, Union

from ..utils.embedding_utils import make_doc_embedding, read_in_embeddings
from ..utils.file_utils import read_alignments
from .dp_utils import vecalign

logger = logging.getLogger("vecalign")
logger.propagate = False


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-s", "--src", type=str, dest="src", required=True, help='Source file.')
    p.add_argument('-t', '--tgt', type=str, dest="tgt", required=True, help='Target file.')
    p.add_argument('--src_embed', type=str, nargs=2, required=True,
                   help='Source embeddings: a text file of candidates, then the binary embeddings file.')
    p.add_argument("--src_stopes", action="store_true", default=False, help="source embeddings were dumped by stopes (.npy)")
    p.add_argument("--src_fp16", action="store_true", default=False, help="source embeddings are raw fp16 (SONAR)")
    p.add_argument('--tgt_embed', type=str, nargs=2, required=True,
                   help='Target embeddings: a text file of candidates, then the binary embeddings file.')
    p.add_argument("--tgt_stopes", action="store_true", default=False, help="target embeddings were dumped by stopes (.npy)")
    p.add_argument("--tgt_fp16", action="store_true", default=False, help="target embeddings are raw fp16 (SONAR)")
    p.add_argument('-a', '--alignment_max_size', dest="alignment_max_size", type=int, default=10,
                   help='Searches for alignments up to size N-M, where N+M <= this value.')
    p.add_argument('--many_to_one', type=int, nargs='?', default=None, const=50,
                   help='Many-to-one (1:1 ... M:1) alignment; M defaults to 50. Overrides -a.')
    p.add_argument('-d', '--del_percentile_frac', type=float, default=0.2,
                   help='Deletion penalty = this percentile (fraction) of the sampled cost distribution.')
    p.add_argument('--search_buffer_size', type=int, default=5, help='Width (one side) of the search buffer.')
    p.add_argument('--max_size_full_dp', type=int, default=300, help='Largest N for full N^2 dynamic programming.')
    p.add_argument('--costs_sample_size', type=int, default=20000, help='Samples for the cost distribution.')
    p.add_argument('--num_samps_for_norm', type=int, default=100, help='Samples for normalising embeddings.')
    p.add_argument("--overlap_segments", default=False, action="store_true",
                   help="speech mode: each line holds a start and an end frame.")
    p.add_argument("--src_ignore_indices", default=None, type=str, help="(start, j) pairs whose overlaps are zeroed (source)")
    p.add_argument("--tgt_ignore_indices", default=None, type=str, help="(start, j) pairs whose overlaps are zeroed (target)")
    p.add_argument('-g', '--gold_alignment', dest="gold_alignment", type=str, required=False, default=None,
                   help='gold alignment file to score against')
    p.add_argument('--print_aligned_text', action='store_true', help='Print aligned text next to the alignments.')
    p.add_argument("--save_to_file", type=str, default=None, dest="save_aligned_text_to_file",
                   help="If given, write the alignments to this file.")
    p.add_argument("-v", "--verbose", dest="verbose", action="store_true", help='console logging at DEBUG')
    p.add_argument('--debug_save_stack', type=str, default=None, help='Write the result stack to a pickle file')
    p.add_argument("--print_results", default=False, action="store_true", help="whether to print results at all.")
    # additive: search region (the reference always runs its coarse-to-fine recursion)
    p.add_argument("--mode", choices=["ref", "band", "dense"], default="ref",
                   help="ref = coarse-to-fine (default); band = Sakoe-Chiba band of --band cells around the straight diagonal; "
                        "dense = every cell of the lattice")
    p.add_argument("--band", type=int, default=2048, help="--mode band: cells per diagonal (2 * width_over2)")
    return p.parse_args(argv)


def make_alignment_types(max_alignment_size: int):
    """All (n, m), n, m >= 1, n + m <= max size, n-major order: the order defines DP tie-breaking."""
    return [(x, y) for x in range(1, max_alignment_size) for y in range(1, max_alignment_size)
            if x + y <= max_alignment_size]


def make_many_to_one_alignment_types(max_alignment_size: int):
    return [(m, 1) for m in range(1, max_alignment_size + 1)]


def print_alignments(alignments, scores=None, src_lines=None, tgt_lines=None, ofile=sys.stdout):
    """`[src ids]:[tgt ids]:%.6f` per alignment (vecalign.py:174-184)."""
    if scores is None:
        scores = [None] * len(alignments)
    for (x, y), s in zip(alignments, scores):
        print('%s:%s' % (x, y) if s is None else '%s:%s:%.6f' % (x, y, s), file=ofile)
        if src_lines is not None and tgt_lines is not None:
            print(' ' * 40, 'SRC: ', ' '.join([src_lines[i].replace('\n', ' ').strip() for i in x]), file=ofile)
            print(' ' * 40, 'TGT: ', ' '.join([tgt_lines[i].replace('\n', ' ').strip() for i in y]), file=ofile)


def load_ignore_index_file(path: Union[str, Path]) -> Set[Tuple[int, int]]:
    res = set()
    with open(path) as fp:
        for line in fp:
            i, j = line.strip().split(" ")
            item = (int(i), int(j))
            assert item not in res, f"{path}, {item}"
            res.add(item)
    return res


def resolve_search_params(alignment_max_size: int, many_to_one: Optional[int], search_buffer_size: int):
    """alignment types, per-side overlap counts and band half-width (vecalign.py:230-243)."""
    if alignment_max_size < 2:
        logger.warning('Alignment_max_size < 2. Increasing to 2 so that 1-1 alignments will be considered')
        alignment_max_size = 2
    src_k = many_to_one if many_to_one is not None else alignment_max_size - 1
    tgt_k = 1 if many_to_one is not None else alignment_max_size - 1
    types = make_many_to_one_alignment_types(many_to_one) if many_to_one is not None else make_alignment_types(alignment_max_size)
    width_over2 = math.ceil(max(src_k, tgt_k) / 2.0) + search_buffer_size
    return types, src_k, tgt_k, width_over2


def load_document(seg_file, embed: List[str], use_stopes: bool, fp16: bool, max_overlaps: int,
                  ignore_indices: Optional[Union[str, Path]], overlap_segments: bool):
    """-> (lines, candidate tensor [max_overlaps, n_lines, d] on the device)."""
    lines = open(seg_file, 'rt', encoding="utf-8").readlines()
    if overlap_segments:
        # speech segments: the table comes from the native parser, the rows go file -> pinned memory -> device
        from ..utils.embedding_utils import candidate_table_from_files, gather_candidates, read_embeddings_pinned
        table, _ = candidate_table_from_files(seg_file, embed[0], max_overlaps, ignore_indices)
        return lines, gather_candidates(read_embeddings_pinned(embed[1], use_stopes, fp16), table)
    sent2id, embeddings = read_in_embeddings(embed[0], embed[1], use_stopes, fp16)
    vectors = make_doc_embedding(sent2id, embeddings, lines, max_overlaps,
                                 ignore_indices=load_ignore_index_file(ignore_indices) if ignore_indices else None,
                                 overlap_segments=overlap_segments)
    return lines, vectors


def align(src: str, tgt: str, src_embed: List[str], src_stopes: bool, tgt_stopes: bool, tgt_embed: List[str],
          alignment_max_size: int, many_to_one: Optional[int], search_buffer_size: int, del_percentile_frac: float,
          max_size_full_dp: int, costs_sample_size: int, num_samps_for_norm: int, overlap_segments: bool,
          print_aligned_text: bool, src_fp16: bool = False, tgt_fp16: bool = False,
          src_ignore_indices: Optional[Union[str, Path]] = None, tgt_ignore_indices: Optional[Union[str, Path]] = None,
          verbose: bool = False, debug_save_stack: Optional[str] = None, gold_alignment: Optional[str] = None,
          print_results: bool = False, save_aligned_text_to_file: Optional[str] = None, mode: str = "ref", band: int = 2048):
    """Align one pair of documents (vecalign.py:198-293); returns the result stack."""
    if verbose:
        logger.setLevel(logging.DEBUG)
    types, src_k, tgt_k, width_over2 = resolve_search_params(alignment_max_size, many_to_one, search_buffer_size)
    logger.debug('Considering alignment types %s', types)
    src_lines, src_vectors = load_document(src, src_embed, src_stopes, src_fp16, src_k, src_ignore_indices, overlap_segments)
    tgt_lines, tgt_vectors = load_document(tgt, tgt_embed, tgt_stopes, tgt_fp16, tgt_k, tgt_ignore_indices, overlap_segments)
    if src_vectors.dtype != tgt_vectors.dtype:
        src_vectors, tgt_vectors = src_vectors.float(), tgt_vectors.float()
    logger.info(f'Aligning src={src} to tgt={tgt}')
    if mode == "ref":
        stack = vecalign(vecs0=src_vectors, vecs1=tgt_vectors, final_alignment_types=types,
                         del_percentile_frac=del_percentile_frac, width_over2=width_over2,
                         max_size_full_dp=max_size_full_dp, costs_sample_size=costs_sample_size,
                         num_samps_for_norm=num_samps_for_norm, full_stack=bool(debug_save_stack))
    else:
        from .dp_utils import align_band
        w2 = max(3, (band + 1) // 2) if mode == "band" else max(int(src_vectors.shape[1]), int(tgt_vectors.shape[1])) + 1
        al, sc = align_band(src_vectors, tgt_vectors, types, del_percentile_frac, w2, costs_sample_size, num_samps_for_norm)
        stack = {0: {'final_alignments': al, 'alignment_scores': sc, 'size0': int(src_vectors.shape[1]), 'size1': int(tgt_vectors.shape[1]),
                     'alignment_types': list(types)}}
    if print_results:
        fp = open(save_aligned_text_to_file, mode="w") if save_aligned_text_to_file else sys.stdout
        print_alignments(stack[0]['final_alignments'], scores=stack[0]['alignment_scores'],
                         src_lines=src_lines if print_aligned_text else None,
                         tgt_lines=tgt_lines if print_aligned_text else None, ofile=fp)
        if save_aligned_text_to_file:
            fp.close()
    if debug_save_stack:
        pickle.dump(stack, open(debug_save_stack, mode="wb"))
    if gold_alignment is not None:
        from .score import log_final_scores, score_multiple
        res = score_multiple(gold_list=[read_alignments(gold_alignment)], test_list=[stack[0]['final_alignments']])
        log_final_scores(res)
    return stack


if __name__ == '__main__':
    align(**vars(parse_args()))

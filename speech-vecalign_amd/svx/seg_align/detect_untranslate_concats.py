"""From untranslated SEGMENT ids to the (start, j) candidate pairs the aligner must zero out.

Same behaviour and CLI as the reference's svecalign/seg_align/detect_untranslate_concats.py
(get_identical_overlap_ids :65-102, main :105-148): walking the same candidate windows as concat_segs,
a start segment i yields (i, i) if i itself is untranslated, otherwise the first (i, j) whose newly
added segment j is untranslated; everything from that (i, j) on is padded by make_overlap
(svecalign/utils/embedding_utils.py:123-126).

    python -m svx.seg_align.detect_untranslate_concats METADATA OUT_DIR --seg_dir D --identical_seg_dir D \
        --src_lang en --tgt_lang de [--num_overlaps 5] [--max_dur 20]
"""
import argparse
import logging
from pathlib import Path
from typing import List, Set, Tuple, Union

from ..utils.file_utils import read_metadata, read_segments
from ..utils.log_utils import my_tqdm
from .concat_segs import SAMPLE_RATE, candidate_windows

logger = logging.getLogger(__name__)


def load_indices(path: Union[str, Path]) -> Set[int]:
    with open(path) as fp:
        return {int(line.strip()) for line in fp if line.strip()}


def get_identical_overlap_ids(in_path: Union[str, Path], num_overlaps: int, max_frames: int,
                              identical_segs_path: Union[str, Path]) -> List[Tuple[int, int]]:
    segs = read_segments(in_path)
    untranslated = load_indices(identical_segs_path)
    out, closed = [], -1  # `closed` = last start index that already produced its pair
    for i, j in candidate_windows(segs, num_overlaps, max_frames):
        if i != closed and j in untranslated:
            out.append((i, j))
            closed = i
    return out


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("metadata", type=str, help="the meta file that each line contains paired audio paths")
    p.add_argument("out_dir", type=str, help="the output dir.")
    p.add_argument("--seg_dir", type=str, required=True, help="dir saving all segments.")
    p.add_argument("--identical_seg_dir", type=str, required=True, help="dir saving all untranslated identical segment ids.")
    p.add_argument("--src_lang", required=True, type=str)
    p.add_argument("--tgt_lang", required=True, type=str)
    p.add_argument("--num_overlaps", type=int, default=5, help="Maximum number of allowed overlaps.")
    p.add_argument("--max_dur", type=float, default=20.0, help="Maximum time each concatenated segment can have. (in seconds)")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    logger.info(args)
    src_lang, tgt_lang = args.src_lang, args.tgt_lang
    seg_dir = Path(args.seg_dir)
    ident = Path(args.identical_seg_dir) / f"{src_lang}-{tgt_lang}"
    out_dir = Path(args.out_dir) / f"{src_lang}-{tgt_lang}"
    out_dir.mkdir(parents=True, exist_ok=True)
    max_frames = int(args.max_dur * SAMPLE_RATE)
    for src_audio, tgt_audio in my_tqdm(read_metadata(args.metadata)):
        s, t = Path(src_audio).stem, Path(tgt_audio).stem
        for side, lang, stem in (("src", src_lang, s), ("tgt", tgt_lang, t)):
            pairs = get_identical_overlap_ids(seg_dir / lang / f"{stem}.txt", args.num_overlaps, max_frames,
                                              ident / f"{s}-{t}.{side}.txt")
            with open(out_dir / f"{s}-{t}.{side}.txt", mode="w") as fp:
                fp.writelines(f"{i} {j}\n" for i, j in pairs)
    logger.info("Finished!")


if __name__ == '__main__':
    main()

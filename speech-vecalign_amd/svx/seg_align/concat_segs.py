"""Candidate enumeration: which concatenations of consecutive segments get an embedding.

Same behaviour and CLI as the reference's svecalign/seg_align/concat_segs.py (get_overlaps :64-92,
overlap :95-126, main :129-160): for every segment i that is itself at most `max_dur` long, the
candidates are segments i..j for j < i + num_overlaps while the span end_j - start_i stays within
`max_dur`; the output lines `start end` are sorted as STRINGS (concat_segs.py:118) -- that order is the
row order of the `.embed` files.

    python -m svx.seg_align.concat_segs METADATA OUT_DIR --seg_dir D --lang en [--use_tgt] [--num_overlaps 5] [--max_dur 20]
"""
import argparse
import logging
from pathlib import Path
from typing import Iterator, List, Sequence, Tuple, Union

from ..utils.file_utils import read_metadata, read_segments
from ..utils.log_utils import my_tqdm

SAMPLE_RATE = 16000  # svecalign/utils/audio_utils.py:12
logger = logging.getLogger(__name__)


def candidate_windows(segs: Sequence[Tuple[int, int]], num_overlaps: int, max_frames: int) -> Iterator[Tuple[int, int]]:
    """(i, j) for every embedded candidate = segments i..j (shared by concat_segs and detect_untranslate_concats)."""
    assert num_overlaps > 0, num_overlaps
    for i, (start, end) in enumerate(segs):
        if end - start > max_frames:  # a single segment can already be too long
            continue
        for j in range(i, min(len(segs), i + num_overlaps)):
            if j > i and segs[j][1] - start > max_frames:
                break
            yield i, j


def get_overlaps(in_path: Union[str, Path], num_overlaps: int, max_frames: int) -> List[str]:
    segs = read_segments(in_path)
    return [f"{segs[i][0]} {segs[j][1]}" for i, j in candidate_windows(segs, num_overlaps, max_frames)]


def overlap(in_path: Union[str, Path], out_path: Path, num_overlaps: int, min_dur: float = 0.0, max_dur: float = 30.0,
            sample_rate: int = SAMPLE_RATE):
    lines = sorted(get_overlaps(in_path, num_overlaps, max_frames=int(max_dur * sample_rate)))  # string sort
    if not lines:
        logger.warning(f"encountered 0 line from {in_path}")
    out_path = Path(out_path)
    out_path.parent.mkdir(parents=True, exist_ok=True)
    with open(out_path, mode="w") as fp:
        fp.writelines(f"{l}\n" for l in lines)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("metadata", type=str, help="the meta file that each line contains paired audio paths")
    p.add_argument("out_dir", type=str, help="the output dir.")
    p.add_argument("--seg_dir", type=str, required=True, help="dir saving all segments.")
    p.add_argument("--lang", type=str, required=True, help="the language code.")
    p.add_argument("--use_tgt", action="store_true", default=False, help="whether to use the target side.")
    p.add_argument("--num_overlaps", type=int, default=5, help="Maximum number of allowed overlaps.")
    p.add_argument("--max_dur", type=float, default=20.0, help="Maximum time each concatenated segment can have. (in seconds)")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    logger.info(args)
    side = 1 if args.use_tgt else 0
    audios = sorted({pair[side] for pair in read_metadata(args.metadata)})
    seg_dir, out_dir = Path(args.seg_dir) / args.lang, Path(args.out_dir) / args.lang
    out_dir.mkdir(parents=True, exist_ok=True)
    for audio in my_tqdm(audios):
        stem = Path(audio).stem
        overlap(seg_dir / f"{stem}.txt", out_dir / f"{stem}.txt", num_overlaps=args.num_overlaps, max_dur=args.max_dur)
    logger.info("Finished!")


if __name__ == '__main__':
    main()

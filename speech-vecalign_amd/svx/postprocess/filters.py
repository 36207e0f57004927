"""The three text-only post-filters between `seg_align.align` and the mining stage.

Same behaviour, file formats and defaults as the reference scripts:
  keep_by_cost        svecalign/postprocess/filter_by_cost.py:39-87   (drop deletions, drop cost > max / < min)
  concat_consecutive  svecalign/postprocess/concat_aligns.py:56-110   (runs of up to `max_num_align` connected
                                                                        alignments, silence and duration limits)
  keep_by_duration    svecalign/postprocess/filter_by_dur.py:43-71    (both sides at least `min_frames` long)
Each has the reference's command line in its own module (filter_by_cost.py, concat_aligns.py, filter_by_dur.py).
"""
import logging
import os
from pathlib import Path
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

from ..utils.file_utils import (alignments_to_timestamps, read_alignments, read_alignments_with_score, read_lines,
                                read_metadata, read_segments, write_alignment)
from ..utils.log_utils import my_tqdm

SAMPLE_RATE = 16000  # svecalign/utils/audio_utils.py:12
logger = logging.getLogger(__name__)
Alignment = Tuple[List[int], List[int]]


def keep_by_cost(in_path: str, out_path: str, max_cost: Optional[float] = None, min_cost: Optional[float] = None) -> float:
    """Write the alignments of `in_path` that are not deletions and pass the threshold; returns kept / total."""
    assert (max_cost is None) != (min_cost is None), f"{min_cost} {max_cost}"
    rows = read_alignments_with_score(in_path)
    kept = [(s, t, c) for s, t, c in rows
            if len(s) and len(t) and not (max_cost is not None and c > max_cost) and not (min_cost is not None and c < min_cost)]
    if kept:
        with open(out_path, mode="w") as fp:
            fp.writelines(f"{s}:{t}:{c}\n" for s, t, c in kept)
    else:
        logger.warning("Empty output. Will not write!")
    n_del = sum(1 for s, t, _ in rows if not (len(s) and len(t)))
    logger.debug(f"{os.path.basename(in_path)} || Threshold: {max_cost} || #Kept: {len(kept)}/{len(rows)} "
                 f"|| #Low quality: {len(rows) - len(kept) - n_del} || #Deletions: {n_del}")
    return len(kept) / len(rows)


def concat_consecutive(alignments: Sequence[Alignment], src_frames: Sequence[Tuple[int, int]],
                       tgt_frames: Sequence[Tuple[int, int]], max_num_align: int, max_sil: float, max_dur: float,
                       sample_rate: int = SAMPLE_RATE, apply_dur_cond_to_both_sides: bool = False) -> List[Alignment]:
    """Every alignment, followed by its extensions with the next 1 .. max_num_align-1 alignments as long as the
    next one is directly connected on both sides, the gap is at most `max_sil` seconds on both sides and the
    source span (optionally also the target span) stays within `max_dur` seconds."""
    assert max_num_align >= 1, max_num_align
    out: List[Alignment] = []
    for i, (s0, t0) in enumerate(alignments):
        src, tgt = list(s0), list(t0)
        out.append((src, tgt))
        for nxt_s, nxt_t in alignments[i + 1:i + max_num_align]:
            if (src_frames[nxt_s[-1]][1] - src_frames[src[0]][0]) / sample_rate > max_dur:
                break
            if apply_dur_cond_to_both_sides and (tgt_frames[nxt_t[-1]][1] - tgt_frames[tgt[0]][0]) / sample_rate > max_dur:
                break
            if nxt_s[0] != src[-1] + 1 or nxt_t[0] != tgt[-1] + 1:
                break
            if (src_frames[nxt_s[0]][0] - src_frames[src[-1]][1]) / sample_rate > max_sil or \
                    (tgt_frames[nxt_t[0]][0] - tgt_frames[tgt[-1]][1]) / sample_rate > max_sil:
                break
            src, tgt = src + list(nxt_s), tgt + list(nxt_t)
            out.append((src, tgt))
    return out


def keep_by_duration(align_path: Path, src_seg_path: Path, tgt_seg_path: Path, min_frames: int, out_path: Path) -> int:
    """Copy the lines of `align_path` whose source AND target spans are at least `min_frames` long."""
    src_t, tgt_t, cnt = alignments_to_timestamps(align_path, read_segments(src_seg_path), read_segments(tgt_seg_path),
                                                 ignore_empty=True)
    lines = read_lines(align_path)
    kept = [lines[i] for i in range(cnt)
            if min_frames <= src_t[i][1] - src_t[i][0] and min_frames <= tgt_t[i][1] - tgt_t[i][0]]
    if kept:
        with open(out_path, mode="w") as fp:
            fp.writelines(l + "\n" for l in kept)
    else:
        logger.info(f"Skip {Path(out_path).as_posix()}. You can double check inputs {Path(align_path).as_posix()}")
    return len(kept)


def for_each_pair(metadata: str, src_lang: str, tgt_lang: str, align_dir: str, out_dir: str,
                  fn: Callable[[str, str, Path, Path], None]) -> None:
    """Shared driver: {align_dir}/{src}-{tgt}/{s}-{t}.txt -> {out_dir}/{src}-{tgt}/{s}-{t}.txt per metadata line."""
    ind = Path(align_dir) / f"{src_lang}-{tgt_lang}"
    outd = Path(out_dir) / f"{src_lang}-{tgt_lang}"
    outd.mkdir(parents=True, exist_ok=True)
    for src, tgt in my_tqdm(read_metadata(metadata)):
        s, t = Path(src).stem, Path(tgt).stem
        fn(s, t, ind / f"{s}-{t}.txt", outd / f"{s}-{t}.txt")

"""python -m svx.postprocess.concat_aligns METADATA OUT_DIR --max_num_align 3 --align_dir D --seg_dir D --src_lang en
--tgt_lang de [--max_sil 1.0] [--max_dur 20.0] [--apply_dur_cond_to_both_sides]
(reference: svecalign/postprocess/concat_aligns.py)"""
import argparse
import logging
from pathlib import Path

from ..utils.file_utils import read_alignments, read_segments, write_alignment
from .filters import SAMPLE_RATE, concat_consecutive, for_each_pair

logger = logging.getLogger(__name__)
group_aligns_by_num = concat_consecutive  # the reference's name


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("metadata", type=str)
    p.add_argument("out_dir", type=str)
    p.add_argument("--max_num_align", type=int, required=True)
    p.add_argument("--align_dir", type=str, required=True)
    p.add_argument("--seg_dir", type=str, required=True)
    p.add_argument("--src_lang", type=str, required=True)
    p.add_argument("--tgt_lang", type=str, required=True)
    p.add_argument("--max_sil", type=float, default=1.0)
    p.add_argument("--max_dur", type=float, default=20.0)
    p.add_argument("--apply_dur_cond_to_both_sides", action="store_true", default=False)
    a = p.parse_args(argv)

    def one(s, t, src, dst):
        if not src.exists():
            logger.warning(f"{src.as_posix()} not exist. Skip.")
            return
        raw = read_alignments(src)
        if not raw:
            logger.warning(f"{src.as_posix()} is empty. Skip.")
            return
        segs_s = read_segments(Path(a.seg_dir) / a.src_lang / f"{s}.txt")
        segs_t = read_segments(Path(a.seg_dir) / a.tgt_lang / f"{t}.txt")
        write_alignment(concat_consecutive(raw, segs_s, segs_t, a.max_num_align, a.max_sil, a.max_dur, SAMPLE_RATE,
                                           a.apply_dur_cond_to_both_sides), dst)
    for_each_pair(a.metadata, a.src_lang, a.tgt_lang, a.align_dir, a.out_dir, one)


if __name__ == '__main__':
    main()

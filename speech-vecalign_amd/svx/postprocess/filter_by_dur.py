"""python -m svx.postprocess.filter_by_dur METADATA OUT_DIR --align_dir D --seg_dir D [--min_dur 1.0] --src_lang en
--tgt_lang de   (reference: svecalign/postprocess/filter_by_dur.py)"""
import argparse
import logging
from pathlib import Path

from .filters import SAMPLE_RATE, for_each_pair, keep_by_duration

logger = logging.getLogger(__name__)
do_filter = keep_by_duration  # the reference's name


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("metadata", type=str)
    p.add_argument("out_dir", type=str)
    p.add_argument("--align_dir", type=str, required=True)
    p.add_argument("--seg_dir", type=str, required=True)
    p.add_argument("--min_dur", type=float, default=1.0)
    p.add_argument("--src_lang", type=str, required=True)
    p.add_argument("--tgt_lang", type=str, required=True)
    a = p.parse_args(argv)
    min_frames = int(SAMPLE_RATE * a.min_dur)

    def one(s, t, src, dst):
        if not src.exists():
            logger.warning(f"{src.as_posix()} not exist. Skip.")
            return
        keep_by_duration(src, Path(a.seg_dir) / a.src_lang / f"{s}.txt", Path(a.seg_dir) / a.tgt_lang / f"{t}.txt", min_frames, dst)
    for_each_pair(a.metadata, a.src_lang, a.tgt_lang, a.align_dir, a.out_dir, one)


if __name__ == '__main__':
    main()

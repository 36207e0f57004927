"""python -m svx.postprocess.filter_by_cost METADATA OUT_DIR --align_dir D --max_cost 0.7 --src_lang en --tgt_lang de
(reference: svecalign/postprocess/filter_by_cost.py)"""
import argparse
import logging

from .filters import for_each_pair, keep_by_cost

logger = logging.getLogger(__name__)
do_filter = keep_by_cost  # the reference's name


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("metadata", type=str)
    p.add_argument("out_dir", type=str)
    p.add_argument("--align_dir", type=str, required=True, help="where the alignments are saved.")
    p.add_argument("--max_cost", type=float, required=True, help="the threshold.")
    p.add_argument("--src_lang", type=str, required=True)
    p.add_argument("--tgt_lang", type=str, required=True)
    a = p.parse_args(argv)
    poor = []

    def one(s, t, src, dst):
        if keep_by_cost(src.as_posix(), dst.as_posix(), max_cost=a.max_cost) < 0.5:
            poor.append(f"{s}-{t}")
    for_each_pair(a.metadata, a.src_lang, a.tgt_lang, a.align_dir, a.out_dir, one)
    logger.info(f"{len(poor)} pairs kept less than half alignments.")


if __name__ == '__main__':
    main()

"""python -m svx.postprocess.sort_tsv --in_tsv A.tsv.gz --out_tsv B.tsv.gz
(reference: svecalign/postprocess/sort_tsv.py): manifest lines by descending score, stable."""
import argparse
import logging
from pathlib import Path

from ..utils import file_utils

logger = logging.getLogger(__name__)


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--in_tsv", type=str, required=True, help="input tsv")
    p.add_argument("--out_tsv", type=str, required=True, help="output tsv")
    a = p.parse_args(argv)
    out = Path(a.out_tsv)
    assert not out.exists(), f"Output file {out} exists!"
    out.parent.mkdir(parents=True, exist_ok=True)
    data = []
    with file_utils.open(a.in_tsv) as fp:
        for line in fp:
            score, _, _ = line.strip().split("\t")
            data.append((float(score), line.strip()))
    data.sort(key=lambda x: -x[0])
    with file_utils.open(out, mode="w") as fp:
        for _, line in data:
            fp.write(line + "\n")
    logger.info("Finished!")


if __name__ == '__main__':
    main()

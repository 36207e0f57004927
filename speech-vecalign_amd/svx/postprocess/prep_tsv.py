"""python -m svx.postprocess.prep_tsv METADATA OUT_DIR --src_lang en --tgt_lang de --align_dir A --seg_dir S
(reference: svecalign/postprocess/prep_tsv.py)

Margin-scored alignment files -> one training manifest, best score first:
    [7, 8, 9]:[7, 8, 9]:0.95855   ->   0.95855 \\t src_audio start end 16 \\t tgt_audio start end 16"""
import argparse
import logging
from pathlib import Path
from typing import List, Tuple, Union

from ..utils import file_utils
from ..utils.file_utils import alignments_to_timestamps, read_alignments_with_score, read_segments

logger = logging.getLogger(__name__)


def make_meta(align_path: Union[Path, str], src_seg_path: Union[Path, str], tgt_seg_path: Union[Path, str],
              src_audio_path: str, tgt_audio_path: str) -> List[Tuple[float, str]]:
    """(score, tsv line) for every alignment of one file (prep_tsv.py:50-81); empty alignments are an error."""
    src_frames, tgt_frames, tot = alignments_to_timestamps(align_path, read_segments(src_seg_path), read_segments(tgt_seg_path),
                                                           ignore_empty=False)
    alignments = read_alignments_with_score(align_path)
    assert len(src_frames) == len(tgt_frames) == len(alignments)
    res = []
    for i in range(tot):
        score = float(alignments[i][2])
        res.append((score, f"{score}\t{src_audio_path} {src_frames[i][0]} {src_frames[i][1]} 16"
                           f"\t{tgt_audio_path} {tgt_frames[i][0]} {tgt_frames[i][1]} 16"))
    return res


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("metadata", type=str, help="the meta file that each line contains paired audio paths")
    p.add_argument("out_dir", type=str, help="output dir of the tsv file.")
    p.add_argument("--src_lang", type=str, required=True)
    p.add_argument("--tgt_lang", type=str, required=True)
    p.add_argument("--align_dir", type=str, required=True, help="dir to all alignments.")
    p.add_argument("--seg_dir", type=str, required=True, help="dir for original segments.")
    a = p.parse_args(argv)
    logger.info(a)
    align_dir = Path(a.align_dir) / f"{a.src_lang}-{a.tgt_lang}"
    out_dir = Path(a.out_dir) / f"{a.src_lang}-{a.tgt_lang}"
    out_dir.mkdir(parents=True, exist_ok=True)
    out_path = out_dir / "align.tsv.gz"
    assert not out_path.exists(), f"{out_path} exists. Will not overwrite."
    rows = []
    for src_aud, tgt_aud in file_utils.read_metadata(a.metadata):
        s, t = Path(src_aud).stem, Path(tgt_aud).stem
        align_path = align_dir / f"{s}-{t}.txt"
        if not align_path.exists():
            logger.warning(f"{align_path} not exist. Skip.")
            continue
        rows.extend(make_meta(align_path, Path(a.seg_dir) / a.src_lang / f"{s}.txt", Path(a.seg_dir) / a.tgt_lang / f"{t}.txt",
                              src_aud, tgt_aud))
    rows.sort(key=lambda x: -x[0])  # highest margin score first (stable: ties keep file order)
    with file_utils.open(out_path, mode="w") as fp:
        for _, line in rows:
            fp.write(line + "\n")
    logger.info("Finished!")


if __name__ == '__main__':
    main()

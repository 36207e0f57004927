"""Text formats of the alignment path (reference: svecalign/utils/file_utils.py):
metadata tsv (:21-22), segment files (:66-77), alignment files (:80-117,120-126)."""
import builtins
import gzip
import logging
import lzma
from ast import literal_eval
from pathlib import Path
from typing import IO, List, Optional, Tuple, Union

logger = logging.getLogger(__name__)


def open(filename: Union[Path, str], mode: str = "rt", encoding: Optional[str] = "utf-8") -> IO:  # noqa: A001
    """Transparent .gz / .xz open (file_utils.py:48-63)."""
    if len(mode) == 1:
        mode += "t"
    if "b" in mode:
        encoding = None
    filename = Path(filename)
    opener = {".gz": gzip.open, ".xz": lzma.open}.get(filename.suffix, builtins.open)
    return opener(filename, encoding=encoding, mode=mode)


def read_lines(path: Union[str, Path]) -> List[str]:
    with open(path) as fp:
        return [line.strip() for line in fp]


def read_metadata(path: Union[str, Path]) -> List[Tuple[str, ...]]:
    """One `src_audio<TAB>tgt_audio` pair per line."""
    return [tuple(line.split("\t")) for line in read_lines(path)]


def check_exist(path: Union[Path, str]) -> bool:
    path = Path(path)
    if not path.exists():
        logger.warning(f"{path} does not exist.")
        return False
    return True


def read_segments(path: Union[str, Path]) -> List[Tuple[int, int]]:
    """`start end` (frames) per line."""
    res = []
    with open(path) as fp:
        for line in fp:
            parts = line.strip().split(" ")
            assert len(parts) == 2, parts
            res.append((int(parts[0]), int(parts[1])))
    return res


def _fields(line: str) -> List[str]:
    return [x.strip() for x in line.split(':') if len(x.strip())]


def read_alignments(fin) -> List[Tuple[List[int], List[int]]]:
    """`[src ids]:[tgt ids][:score]` per line -> [(src, tgt)]."""
    out = []
    with open(fin, 'rt', encoding="utf-8") as infile:
        for line in infile:
            f = _fields(line)
            if len(f) < 2:
                raise Exception('Got line "%s", which does not have at least two ":" separated fields' % line.strip())
            try:
                out.append((literal_eval(f[0]), literal_eval(f[1])))
            except Exception:
                raise Exception('Failed to parse line "%s"' % line.strip())
    return out


def read_alignments_with_score(fin) -> List[Tuple[List[int], List[int], float]]:
    out = []
    with open(fin, 'rt', encoding="utf-8") as infile:
        for line in infile:
            f = _fields(line)
            assert len(f) == 3, 'Got line "%s", which does not have at least two ":" separated fields' % line.strip()
            try:
                out.append((literal_eval(f[0]), literal_eval(f[1]), float(f[2])))
            except Exception:
                raise Exception('Failed to parse line "%s"' % line.strip())
    return out


def write_alignment(alignments, path: Union[Path, str]) -> None:
    with open(path, mode="w") as fp:
        for src_segs, tgt_segs in alignments:
            fp.write(f"{src_segs}:{tgt_segs}\n")


def alignments_to_timestamps(align, src_segs, tgt_segs, ignore_empty: bool = True):
    """Index alignments -> (src (start,end) list, tgt (start,end) list, count)  (file_utils.py:129-165)."""
    if isinstance(align, (str, Path)):
        alignments = read_alignments(align)
    elif isinstance(align, list):
        alignments = align
    else:
        raise TypeError(f"{align} type is unexpected. {type(align)}")
    src_out, tgt_out = [], []
    for src, tgt in alignments:
        if not src or not tgt:
            if ignore_empty:
                continue
            raise Exception("Got empty alignments!")
        src_out.append((src_segs[src[0]][0], src_segs[src[-1]][1]))
        tgt_out.append((tgt_segs[tgt[0]][0], tgt_segs[tgt[-1]][1]))
    return src_out, tgt_out, len(src_out)

"""Segment-alignment driver: the reference's `python -m svecalign.seg_align.align` on MI355X.

Same positional arguments, flags, directory conventions and output format as
svecalign/seg_align/align.py (parse_args :13-96, validate_inputs :117-179, main :182-230):

    python -m svx.seg_align.align METADATA OUT_DIR --src_lang en --tgt_lang de \
        --seg_dir D --concat_dir D --embed_dir D [--is_stopes_embed] [--fp16_embed] [-a 6] ...

Additive flags only: --batch_size (document pairs aligned per device pass; the reference loops one
pair at a time), --seed (per-pair sampling streams derived from (seed, pair index): results do not
depend on batch size or shard count; without it the global numpy stream is consumed pair after
pair exactly like the reference), --skip_existing, --rank/--n_shard (default from torchrun's
RANK/WORLD_SIZE: one process per GPU, pairs split by cost, no collective needed).
"""
import argparse
import dataclasses
import logging
import os
from pathlib import Path
from typing import List, Optional, Tuple, Union

import numpy as np

from ..utils.file_utils import check_exist, read_metadata
from ..utils.log_utils import my_tqdm
from ..utils.mp_utils import balanced_shards
from ..vecalign.vecalign import resolve_search_params

logger = logging.getLogger(__name__)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("metadata", type=str, help="the meta file that each line contains paired audio paths")
    p.add_argument("out_dir", type=str, help="dir to save alignments.")
    p.add_argument("--src_lang", type=str, required=True)
    p.add_argument("--tgt_lang", type=str, required=True)
    p.add_argument("--seg_dir", type=str, required=True, help="the dir for raw segments.")
    p.add_argument("--concat_dir", type=str, required=True, help="the dir for concatenated segments.")
    p.add_argument("--embed_dir", type=str, required=True, help="Dir to embedding files.")
    p.add_argument("--is_stopes_embed", action="store_true", default=False, help="embeddings were dumped by stopes (SpeechLASER).")
    p.add_argument("--fp16_embed", action="store_true", default=False, help="embeddings are raw fp16 (SONAR numpy dumps)")
    p.add_argument('-a', '--alignment_max_size', dest="alignment_max_size", type=int, default=6,
                   help='Searches for alignments up to size N-M, where N+M <= this value.')
    p.add_argument('--search_buffer_size', type=int, default=5, help='Width (one side) of search buffer.')
    p.add_argument('-d', '--del_percentile_frac', dest="del_percentile_frac", type=float, default=0.2,
                   help='Deletion penalty percentile (fraction) of the cost distribution.')
    p.add_argument('--max_size_full_dp', type=int, default=300, help='Largest N for full N^2 dynamic programming.')
    p.add_argument('--costs_sample_size', type=int, default=20000, help='Samples for the cost distribution.')
    p.add_argument('--num_samps_for_norm', type=int, default=100, help='Samples for normalizing embeddings')
    p.add_argument("--ign_indices_dir", type=str, default=None,
                   help="if provided, then some segments will be ignored when loading embeddings.")
    # additive
    p.add_argument("--mode", choices=["ref", "band", "dense"], default="ref",
                   help="search region: ref = the reference's coarse-to-fine recursion (default); band = Sakoe-Chiba band of "
                        "--band cells around the straight diagonal; dense = every cell of the lattice (include/svx.h: SVX_SEARCH_STRAIGHT)")
    p.add_argument("--band", type=int, default=2048, help="--mode band: cells per diagonal (2 * width_over2)")
    p.add_argument("--batch_size", type=int, default=32, help="document pairs per device pass")
    p.add_argument("--io_threads", type=int, default=None, help="host threads that parse / read ahead of the GPU (default: the CPU count, at most 32)")
    p.add_argument("--seed", type=int, default=None, help="derive one sampling stream per pair from (seed, pair index)")
    p.add_argument("--skip_existing", action="store_true", default=False, help="do not recompute existing outputs")
    p.add_argument("--rank", type=int, default=int(os.environ.get("RANK", 0)))
    p.add_argument("--n_shard", type=int, default=int(os.environ.get("WORLD_SIZE", 1)))
    return p.parse_args(argv)


@dataclasses.dataclass
class VecalignData:
    src_seg_path: str
    tgt_seg_path: str
    src_concat_path: str
    tgt_concat_path: str
    src_embed_path: str
    tgt_embed_path: str
    output_path: str
    src_ignore_indices: Optional[Union[str, Path]] = None
    tgt_ignore_indices: Optional[Union[str, Path]] = None
    index: int = 0  # position in the validated list (seeds derive from it)


def validate_inputs(audio_pairs: List[Tuple[str, str]], src_seg_dir: Path, tgt_seg_dir: Path, src_concat_dir: Path,
                    tgt_concat_dir: Path, src_embed_dir: Path, tgt_embed_dir: Path, out_dir: Path,
                    ign_indices_dir: Optional[Path] = None) -> List[VecalignData]:
    """Resolve the per-pair file set; pairs with a missing file are dropped (align.py:117-179)."""
    res = []
    for src_audio, tgt_audio in audio_pairs:
        s, t = Path(src_audio), Path(tgt_audio)
        found = {}
        for kind, sdir, tdir, suffix in (("seg", src_seg_dir, tgt_seg_dir, ".txt"),
                                         ("concat", src_concat_dir, tgt_concat_dir, ".txt"),
                                         ("embed", src_embed_dir, tgt_embed_dir, ".embed")):
            sp, tp = (sdir / s.name).with_suffix(suffix), (tdir / t.name).with_suffix(suffix)
            if not check_exist(sp) or not check_exist(tp):
                found = None
                break
            found[kind] = (sp.as_posix(), tp.as_posix())
        if found is None:
            continue
        src_ign = tgt_ign = None
        if ign_indices_dir is not None:
            src_ign = ign_indices_dir / f"{s.stem}-{t.stem}.src.txt"
            tgt_ign = ign_indices_dir / f"{s.stem}-{t.stem}.tgt.txt"
            src_ign = src_ign if check_exist(src_ign) else None
            tgt_ign = tgt_ign if check_exist(tgt_ign) else None
        res.append(VecalignData(found["seg"][0], found["seg"][1], found["concat"][0], found["concat"][1],
                                found["embed"][0], found["embed"][1], (out_dir / f"{s.stem}-{t.stem}.txt").as_posix(),
                                src_ign, tgt_ign, index=len(res)))
    return res


def pair_rng(seed: Optional[int], index: int):
    """Independent legacy RandomState per pair, a pure function of (seed, pair index)."""
    if seed is None:
        return None
    return np.random.RandomState(np.random.SeedSequence([seed, index]).generate_state(4))


def format_alignment_rows(rows: np.ndarray, scores: Optional[np.ndarray]) -> bytes:
    """print_alignments' text (vecalign.py:174-184) for alignment rows (x_start, x_len, y_start, y_len), by the
    native formatter (svx_format_alignments): byte-identical to the Python '%s:%s:%.6f' of lists."""
    import ctypes
    from .. import _lib
    lib = _lib.load()
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    n = int(rows.shape[0])
    sc = None if scores is None else np.ascontiguousarray(scores, dtype=np.float64)
    cap = 64 * n + 16 * int(rows[:, 1].sum() + rows[:, 3].sum()) + 16
    buf = ctypes.create_string_buffer(cap)
    need = lib.svx_format_alignments(ctypes.c_void_p(rows.ctypes.data), ctypes.c_void_p(sc.ctypes.data if sc is not None else 0), n, buf, cap)
    if need < 0 or need > cap:
        raise Exception("svx_format_alignments: buffer of %d bytes, %d needed" % (cap, need))
    return buf.raw[:need]


def _write_result(path: str, rows: np.ndarray, scores: np.ndarray):
    tmp = path + ".tmp"
    with open(tmp, "wb") as fp:  # write-then-rename, the repo's crash-safety idiom
        fp.write(format_alignment_rows(rows, scores))
    Path(tmp).replace(path)


def _prepare_pair(p: VecalignData, args, src_k: int, tgt_k: int):
    """Host half of one document pair (runs on a pool thread; the heavy parts are native / I/O and release the
    GIL): candidate index tables from the segment and candidate files, embedding files into pinned memory."""
    from ..utils.embedding_utils import candidate_table_from_files, read_embeddings_pinned
    st, _ = candidate_table_from_files(p.src_seg_path, p.src_concat_path, src_k, p.src_ignore_indices)
    tt, _ = candidate_table_from_files(p.tgt_seg_path, p.tgt_concat_path, tgt_k, p.tgt_ignore_indices)
    se = read_embeddings_pinned(p.src_embed_path, args.is_stopes_embed, args.fp16_embed)
    te = read_embeddings_pinned(p.tgt_embed_path, args.is_stopes_embed, args.fp16_embed)
    return st, tt, se, te


def align_pairs(pairs: List[VecalignData], args, batch_size: int, io_threads: Optional[int] = None, stats: Optional[dict] = None):
    """The reference's serial loop (align.py:206-230) as a three-stage pipeline around svx_align_batch:
      host threads   parse segment / candidate files into index tables, read .embed files into pinned memory
                     (`io_threads`, running ahead by two batches);
      copy stream    uploads batch i+1 while the compute stream aligns batch i (device gather of the candidate
                     tensor, then the whole of vecalign());
      writer threads format and write the alignment files of batch i-1.
    Results do not depend on the batch size, the thread count or the shard count when --seed is given."""
    import torch
    from collections import deque
    from multiprocessing.pool import ThreadPool
    from .. import _lib
    from ..utils.embedding_utils import gather_candidates
    from ..vecalign.dp_utils import PreparedBatch
    types, src_k, tgt_k, width_over2 = resolve_search_params(args.alignment_max_size, None, args.search_buffer_size)
    todo = [p for p in pairs if not (args.skip_existing and Path(p.output_path).exists())]
    if not todo:
        return
    ctx = _lib.context()
    dev = ctx.tdev
    nthr = io_threads or max(2, min(32, (os.cpu_count() or 4)))
    # (the current device is per thread: pool threads that allocate pinned memory must name this rank's GPU, or every
    #  rank would open a context on GPU 0)
    dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
    io_pool = ThreadPool(nthr, initializer=torch.cuda.set_device, initargs=(dev_index,))
    out_pool = ThreadPool(max(2, nthr // 2))
    copy_stream = torch.cuda.Stream(device=dev)
    compute = torch.cuda.current_stream(dev)
    batches = [todo[b0:b0 + batch_size] for b0 in range(0, len(todo), batch_size)]
    ahead = deque()   # (chunk, [async results of _prepare_pair])
    nxt = 0

    def submit_more():
        nonlocal nxt
        while nxt < len(batches) and len(ahead) < 3:
            ahead.append((batches[nxt], [io_pool.apply_async(_prepare_pair, (p, args, src_k, tgt_k)) for p in batches[nxt]]))
            nxt += 1

    def finish(job):
        chunk, pb, ev, held = job
        ev.synchronize()
        info, align, scores, _, offs = pb.raw_results()
        writes = []
        for i, p in enumerate(chunk):
            o, cnt = int(offs[i]), int(info[i, 0])
            writes.append(out_pool.apply_async(_write_result, (p.output_path, align[o:o + cnt], scores[o:o + cnt])))
        return writes, (pb, held)  # (the pinned result buffers stay alive until the writers are done)

    pending_job, pending_writes = None, deque()
    bar = my_tqdm(total=len(todo))
    try:
        submit_more()
        while ahead:
            chunk, futs = ahead.popleft()
            submit_more()
            prepared = [f.get() for f in futs]
            # ---- uploads on the copy stream (pinned -> device), then gather + align on the compute stream
            with torch.cuda.stream(copy_stream):
                dev_in = [(torch.from_numpy(st).to(dev, non_blocking=True), torch.from_numpy(tt).to(dev, non_blocking=True),
                           se.to(dev, non_blocking=True), te.to(dev, non_blocking=True)) for st, tt, se, te in prepared]
                up = torch.cuda.Event()
                up.record(copy_stream)
            compute.wait_event(up)
            docs = []
            for (st, tt, se, te), (dst, dtt, dse, dte) in zip(prepared, dev_in):
                for x in (dst, dtt, dse, dte):
                    x.record_stream(compute)
                sv, tv = gather_candidates(dse, dst), gather_candidates(dte, dtt)
                if sv.dtype != tv.dtype:
                    sv, tv = sv.float(), tv.float()
                docs.append((sv, tv))
            rngs = None if args.seed is None else [pair_rng(args.seed, p.index) for p in chunk]
            mode = getattr(args, "mode", "ref")
            if mode == "ref":
                w2, search = width_over2, "coarse_to_fine"
            elif mode == "band":
                w2, search = max(3, (args.band + 1) // 2), "straight"
            else:  # dense: the band covers the lattice (width_over2 > max(N, M))
                w2, search = max(max(int(sv.shape[1]), int(tv.shape[1])) for sv, tv in docs) + 1, "straight"
            pb = PreparedBatch(docs, types, args.del_percentile_frac, w2, args.max_size_full_dp,
                               args.costs_sample_size, args.num_samps_for_norm, rngs=rngs, search=search)
            pb.run()
            ev = pb.fetch_async()
            job = (chunk, pb, ev, (prepared, dev_in))
            # ---- while this batch computes: hand the previous batch to the writers
            if pending_job is not None:
                pending_writes.append(finish(pending_job))
                bar.update(len(pending_job[0]))
            pending_job = job
            while len(pending_writes) > 1:
                for w in pending_writes.popleft()[0]:
                    w.get()
        if pending_job is not None:
            pending_writes.append(finish(pending_job))
            bar.update(len(pending_job[0]))
        for writes, _ in pending_writes:
            for w in writes:
                w.get()
    except BaseException:
        # a failed pair must not leave threads behind: drop the reads still queued, let the writes already handed
        # over finish (their files are complete or absent: write-then-rename), then re-raise
        io_pool.terminate()
        for writes, _ in pending_writes:
            for w in writes:
                try:
                    w.get(timeout=120)
                except Exception:
                    pass
        out_pool.terminate()
        raise
    finally:
        bar.close()
        io_pool.close()
        out_pool.close()
        io_pool.join()
        out_pool.join()
    if stats is not None:
        stats["pairs"] = len(todo)
        stats["io_threads"] = nthr


def main(argv=None):
    args = parse_args(argv)
    logger.info(args)
    src_lang, tgt_lang = args.src_lang, args.tgt_lang
    out_dir = Path(args.out_dir) / f"{src_lang}-{tgt_lang}"
    out_dir.mkdir(parents=True, exist_ok=True)
    ign = None
    if args.ign_indices_dir is not None:
        ign = Path(args.ign_indices_dir) / f"{src_lang}-{tgt_lang}"
        logger.info(f"Will ignore segments indicated by {ign}")
    valid = validate_inputs(read_metadata(args.metadata),
                            Path(args.seg_dir) / src_lang, Path(args.seg_dir) / tgt_lang,
                            Path(args.concat_dir) / src_lang, Path(args.concat_dir) / tgt_lang,
                            Path(args.embed_dir) / src_lang, Path(args.embed_dir) / tgt_lang, out_dir, ign)
    if args.n_shard > 1:
        # one process per GPU; document pairs are independent, so shards never communicate
        import torch
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", args.rank)) % max(1, torch.cuda.device_count()))
        costs = [os.path.getsize(p.src_embed_path) + os.path.getsize(p.tgt_embed_path) for p in valid]
        valid = [valid[i] for i in balanced_shards(costs, args.n_shard)[args.rank]]
        logger.info(f"rank {args.rank} of {args.n_shard}: {len(valid)} pairs")
    align_pairs(valid, args, max(1, args.batch_size), io_threads=args.io_threads)


if __name__ == '__main__':
    main()

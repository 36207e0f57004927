"""The N > 1 path on CPU (gloo, world_size 2): document pairs are independent, so ranks only share
the work list.  Checks that the shard assignment covers every pair once, that per-pair sampling
streams make results independent of the shard count, and that the bench's timing reduction
(barrier + MAX over ranks) works.  The CPU oracle stands in for the device pipeline here."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "speech-vecalign_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from synth import alignment_types, make_pair
    from svx.seg_align.align import pair_rng
    from svx.utils.mp_utils import balanced_shards
    shapes = [(120, 110), (300, 280), (64, 90), (200, 190), (150, 160), (90, 70), (260, 255)]
    mine = balanced_shards([n + m for n, m in shapes], world)[rank]
    types = alignment_types(4)
    out = {}
    for i in mine:
        v0, v1 = make_pair(shapes[i][0], shapes[i][1], 3, 32, 40 + i)
        st = oracle.vecalign(v0, v1, types, 0.2, 7, 300, 20000, 100, rng=pair_rng(5, i))
        out[i] = (st[0]['final_alignments'], st[0]['alignment_scores'].tolist())
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((gathered, float(t.item())))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_rank():
    ctx = mp.get_context("spawn")
    results = {}
    for world, port in ((1, 29611), (2, 29612)):
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        gathered, tmax = q.get(timeout=180)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        merged = {}
        for part in gathered:
            assert not (set(part) & set(merged))  # no pair aligned twice
            merged.update(part)
        assert sorted(merged) == list(range(7))
        assert tmax == pytest.approx(0.1 * world)
        results[world] = merged
    assert results[1] == results[2]  # shard-count invariant: per-pair streams, no shared state

"""Worker of tests/test_gpu_margin.py::test_global_margin_over_rccl (one process per GPU, started by
torch.distributed.run): every rank scores ITS rows against the union of all ranks' rows -- shards round the ring
(point-to-point) and all-gathered, both over RCCL -- rank 0 also scores everything on one GPU and compares."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-vecalign_amd"))


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from svx.postprocess.score_align import global_margin_scores
    rs = np.random.RandomState(5)
    n, d = 700, 1024
    x = rs.standard_normal((n, d)).astype(np.float32)
    y = (x + 0.3 * rs.standard_normal((n, d))).astype(np.float32)
    cuts = [round(n * r / world) + (17 if 0 < r < world else 0) for r in range(world + 1)]  # uneven shards
    lo, hi = cuts[rank], cuts[rank + 1]
    sizes = [cuts[r + 1] - cuts[r] for r in range(world)]
    cap = max(sizes)
    solo = [dist.new_group([r]) for r in range(world)][rank]   # (every rank creates every group; a group of one exchanges nothing)
    want = None
    if rank == 0:
        want = global_margin_scores(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), k=16, group=solo).cpu().numpy()
    for exchange in ("ring", "allgather"):
        mine = global_margin_scores(torch.from_numpy(x[lo:hi]).cuda(), torch.from_numpy(y[lo:hi]).cuda(), k=16, exchange=exchange)
        # (scores come back per shard; gather them for the comparison)
        pad = torch.zeros(cap, dtype=torch.float32, device="cuda")
        pad[:hi - lo] = mine
        got = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(got, pad)
        if rank == 0:
            allscores = torch.cat([g[:s] for g, s in zip(got, sizes)]).cpu().numpy()
            err = float(np.abs(allscores - want).max())
            print("RCCL_MARGIN %s max|diff| = %.3e over %d rows on %d ranks" % (exchange, err, n, world))
            assert err < 1e-5, err
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

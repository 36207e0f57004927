"""Shard arithmetic for splitting a pair list over GPUs/ranks: same rounding as the reference's
get_shard_range (svecalign/utils/mp_utils.py:7-16), plus a cost-balanced variant."""
from typing import List, Sequence, Tuple


def get_shard_range(tot: int, nshard: int, rank: int) -> Tuple[int, int]:
    assert rank < nshard and rank >= 0, f"invaid rank/nshard {rank}/{nshard}"
    start = round(tot / nshard * rank)
    end = round(tot / nshard * (rank + 1))
    assert start < end, f"start={start}, end={end}"
    return start, end


def balanced_shards(costs: Sequence[float], nshard: int) -> List[List[int]]:
    """Longest-processing-time assignment of items (cost ~ N+M per document pair) to shards;
    deterministic, and every item lands in exactly one shard."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * nshard
    out: List[List[int]] = [[] for _ in range(nshard)]
    for i in order:
        r = min(range(nshard), key=lambda s: (loads[s], s))
        out[r].append(i)
        loads[r] += costs[i]
    for o in out:
        o.sort()
    return out

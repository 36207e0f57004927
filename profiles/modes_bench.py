"""Timing + cross-check of the search modes of SURVEY 8(a) on one synthetic pair (not the headline bench):
Mode A (coarse-to-fine, svx_align_batch) against Mode C (Sakoe-Chiba band around the straight diagonal,
dp_utils.align_band through the per-op C ABI).  python profiles/modes_bench.py --n 32768 --band 2048"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-vecalign_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--m", type=int, default=0)
    ap.add_argument("--band", type=int, default=2048)
    ap.add_argument("--overlaps", type=int, default=4)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    import torch
    from synth import alignment_types, make_pair
    from svx.vecalign import dp_utils
    n, m = a.n, a.m or a.n
    t0 = time.time()
    v0, v1 = make_pair(n, m, a.overlaps, a.d, a.seed)
    t_gen = time.time() - t0
    types = alignment_types(a.overlaps + 1)
    np.random.seed(1)
    torch.cuda.synchronize()
    t0 = time.time()
    st = dp_utils.vecalign(v0, v1, types, 0.2, 7, 300, 20000, 100)
    torch.cuda.synchronize()
    t_a = time.time() - t0
    np.random.seed(1)
    t0 = time.time()
    al_c, sc_c = dp_utils.align_band(v0, v1, types, 0.2, a.band // 2, 20000, 100)
    torch.cuda.synchronize()
    t_c = time.time() - t0
    al_a, sc_a = st[0]['final_alignments'], st[0]['alignment_scores']
    same = al_a == al_c
    print(json.dumps({"n": n, "m": m, "band": a.band, "types": len(types), "gen_s": round(t_gen, 2), "mode_a_s": round(t_a, 3),
                      "mode_c_s": round(t_c, 3), "alignments_a": len(al_a), "alignments_c": len(al_c), "identical_spans": bool(same),
                      "max_abs_dscore": float(np.abs(np.asarray(sc_a) - np.asarray(sc_c)).max()) if same else None}))


if __name__ == "__main__":
    main()

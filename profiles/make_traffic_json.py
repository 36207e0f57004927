#!/usr/bin/env python3
"""rocprofv3 counter CSVs -> profiles/rNN_hbm_traffic.json, the file bench.py's `roofline.traffic` comes from.

    python profiles/make_traffic_json.py --fetch <fetch counter_collection.csv> --write <write counter_collection.csv> \
        --pairs 1024 --out profiles/r02_hbm_traffic.json

HBM bytes per launch of a stage = 2 x FETCH_SIZE + WRITE_SIZE of its kernel, averaged over the kernel's launches in the
counter runs (one warm-up + one timed step).  FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE
tallies the 128-byte requests of a wide read stream at 64 bytes (MI355X_MICROARCH.md, HBM), hence the factor 2.
Stages that launch one kernel several times per step (pyrN, band_costsN, band_dpN: the levels >= 1) get the average
launch, like the average launch time bench.py divides by."""
import argparse
import collections
import csv
import json
import re

STAGE_OF = [  # (regex on the kernel name, stage)
    (r"k_pyramid<Elem(BF16|F16|F32), \d+, 1(, (true|false))?>", "pyr0"), (r"k_pyramid<Elem(BF16|F16|F32), \d+, 2(, (true|false))?>", "pyr1"),
    (r"k_pyramid<ElemF32, \d+, 0(, (true|false))?>", "pyrN"),
    (r"k_knob_scores<Elem(BF16|F16|F32), \d+, true>", "knob_scores0"), (r"k_knob_scores<ElemF32, \d+, false>", "knob_scoresN"),
    (r"k_band_costs3<", "band_costs0"), (r"k_band_costs2<Elem(BF16|F16|F32), true", "band_costs0"),
    (r"k_band_costs2<ElemF32, false", "band_costsN"), (r"k_band_costs_batch<Elem(BF16|F16|F32), true", "band_costs0"),
    (r"k_band_costs_batch<ElemF32, false", "band_costsN"),
    (r"k_sparse_dp_fast_batch<3, 4>", "band_dp0"), (r"k_sparse_dp_fast_batch<1, 1>", "band_dpN"),
    (r"k_sparse_traceback_batch", "traceback"), (r"k_knob_sort", "knob_sort"), (r"k_del_penalty_batch", "knob"),
    (r"k_dense_costs_batch<ElemF32, false>", "dense_costs"), (r"k_dense_stage_batch", "dense_dp"), (r"k_search_path_batch", "path"),
    (r"k_band_tiles<", "tiles"),
]


PASSES = {"pyrN": 3, "band_costsN": 3, "band_dpN": 3}   # levels a stage's kernel runs over per step at the default workload (L = 4)


def table(path, counter, counts=None):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        for rx, st in STAGE_OF:
            if re.search(rx, name):
                agg[st][0] += 1
                agg[st][1] += float(r["Counter_Value"])
                break
    if counts is not None:
        counts.update({k: v[0] for k, v in agg.items()})
    return {k: v[1] / v[0] for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--pairs", type=int, required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--overlaps", type=int, default=4)
    ap.add_argument("--steps_run", type=int, default=2, help="svx_align_batch calls in each counter run (warm-up + timed)")
    a = ap.parse_args()
    counts = {}
    f, w = table(a.fetch, "FETCH_SIZE", counts), table(a.write, "WRITE_SIZE")
    # with the software pipeline a pass over the batch is several launches (half-batches, slices): pairs one launch covers
    ppl = {st: a.pairs * PASSES.get(st, 1) / (counts[st] / float(a.steps_run)) for st in counts}
    out = {"pairs_per_step": a.pairs, "workload": a.workload, "dtype": a.dtype, "N": a.N, "M": a.M, "d": a.d, "overlaps": a.overlaps,
           "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of `bench.py --steps 1 --warmup 1` at this "
                  "pairs per step, profiles/run_profiles.sh, software pipeline on); bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), launches averaged per stage",
           "launches_per_step": {st: counts[st] / float(a.steps_run) for st in sorted(counts)},
           "pairs_per_launch": {st: ppl[st] for st in sorted(ppl)},
           "hbm_bytes_per_launch": {st: (2 * f.get(st, 0.0) + w.get(st, 0.0)) * 1024 for st in sorted(set(f) | set(w))},
           "hbm_read_bytes_per_launch": {st: 2 * f[st] * 1024 for st in sorted(f)},
           "hbm_write_bytes_per_launch": {st: w[st] * 1024 for st in sorted(w)}}
    json.dump(out, open(a.out, "w"), indent=1)
    for st, b in sorted(out["hbm_bytes_per_launch"].items(), key=lambda kv: -kv[1]):
        print("%-14s %8.2f GB per launch  (%6.1f MB per pair per pass, %.0f pairs per launch)" % (st, b / 1e9, b / ppl.get(st, a.pairs) / 1e6, ppl.get(st, a.pairs)))


if __name__ == "__main__":
    main()

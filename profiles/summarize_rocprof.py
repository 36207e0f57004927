#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the per-kernel tables kept under profiles/.

    python profiles/summarize_rocprof.py --stats <*_kernel_stats.csv> --stat_pairs P [--fetch <*_counter_collection.csv>]
                                         [--write <*_counter_collection.csv>] --counter_pairs Q [--json out.json]

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; on gfx950 FETCH_SIZE reports half of the bytes of a
wide coalesced read stream (MI355X_MICROARCH.md, HBM section), so the table shows it doubled as well.
Only this repo's kernels (k_*) are listed; torch kernels that generate the synthetic inputs are dropped.
"""
import argparse
import collections
import csv
import re


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?(k_[a-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else None


def counter_table(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        if k:
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--stat_pairs", type=int, default=0, help="document pairs per launch in the --stats run")
    ap.add_argument("--counter_pairs", type=int, default=0, help="document pairs per launch in the counter runs")
    ap.add_argument("--json", help="also write {kernel: HBM bytes per pair per launch} here")
    a = ap.parse_args()
    fetch = counter_table(a.fetch, "FETCH_SIZE") if a.fetch else {}
    write = counter_table(a.write, "WRITE_SIZE") if a.write else {}
    rows = []
    for r in csv.DictReader(open(a.stats)):
        k = short(r["Name"])
        if k:
            rows.append((k, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
    tot = sum(r[2] for r in rows)
    sp, cp = a.stat_pairs, a.counter_pairs
    print("| kernel | calls | total ms | avg us | share | HBM read MiB/pair/launch (2 x FETCH_SIZE) | HBM write MiB/pair/launch | HBM GB/s |")
    print("|---|---|---|---|---|---|---|---|")
    traffic = {}
    for k, calls, ms, avg in sorted(rows, key=lambda r: -r[2]):
        f = fetch.get(k)
        w = write.get(k)
        if cp and (f is not None or w is not None):
            rd, wr = 2 * (f or 0) / 1024 / cp, (w or 0) / 1024 / cp
            traffic[k] = (rd + wr) * 1024 * 1024
            bw = "%.0f" % ((rd + wr) * 1048576 * sp / (avg * 1e-6) / 1e9) if sp else "-"
            print("| %s | %d | %.3f | %.1f | %.1f%% | %.2f | %.2f | %s |" % (k, calls, ms, avg, 100 * ms / tot, rd, wr, bw))
        else:
            print("| %s | %d | %.3f | %.1f | %.1f%% | - | - | - |" % (k, calls, ms, avg, 100 * ms / tot))
    print("\nsum over svx kernels: %.3f ms" % tot)
    if a.json:
        import json
        json.dump({"counter_pairs_per_launch": cp, "hbm_bytes_per_pair_per_launch": traffic}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()

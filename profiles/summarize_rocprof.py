#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the per-kernel tables kept under profiles/.

    python profiles/summarize_rocprof.py --stats <*_kernel_stats.csv> [--fetch <*_counter_collection.csv>]
                                         [--write <*_counter_collection.csv>] [--pairs P]

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
    ap.add_argument("--pairs", type=int, default=0, help="document pairs per launch (for the per-pair columns)")
    a = ap.parse_args()
    fetch = counter_table(a.fetch, "FETCH_SIZE") if a.fetch else {}
    write = counter_table(a.write, "WRITE_SIZE") if a.write else {}
    rows = []
    for r in csv.DictReader(open(a.stats)):
        k = short(r["Name"])
        if k:
            rows.append((k, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
    tot = sum(r[2] for r in rows)
    print("| kernel | calls | total ms | avg us | share | FETCH_SIZE MiB/launch (x2) | WRITE_SIZE MiB/launch | GB/s (2*fetch+write) |")
    print("|---|---|---|---|---|---|---|---|")
    for k, calls, ms, avg in sorted(rows, key=lambda r: -r[2]):
        f = fetch.get(k)
        w = write.get(k)
        fs = "%.1f (%.1f)" % (f / 1024, 2 * f / 1024) if f is not None else "-"
        ws = "%.1f" % (w / 1024) if w is not None else "-"
        bw = "%.0f" % ((2 * (f or 0) + (w or 0)) * 1024 / (avg * 1e-6) / 1e9) if (f is not None or w is not None) else "-"
        print("| %s | %d | %.3f | %.1f | %.1f%% | %s | %s | %s |" % (k, calls, ms, avg, 100 * ms / tot, fs, ws, bw))
    print("\nsum over svx kernels: %.3f ms" % tot)


if __name__ == "__main__":
    main()

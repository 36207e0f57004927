#!/usr/bin/env python3
"""Per-kernel table of a round-3 profile set (profiles/run_profiles.sh): rocprofv3 --kernel-trace --stats durations of the
default bench command (software pipeline on: kernels of different streams overlap, so durations include what they cost
each other) and, from the separate --pmc passes (kernels serialised by the counter collection), HBM bytes per launch =
2 x FETCH_SIZE + WRITE_SIZE averaged over the kernel's launches.  GB/s = bytes per launch / average duration.

    python profiles/summarize_rocprof_r03.py --stats <kernel_stats.csv> --fetch <counter_collection.csv> --write <counter_collection.csv>
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
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    a = ap.parse_args()
    fetch, write = counter_table(a.fetch, "FETCH_SIZE"), counter_table(a.write, "WRITE_SIZE")
    rows = []
    for r in csv.DictReader(open(a.stats)):
        k = short(r["Name"])
        if k:
            rows.append((k, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
    tot = sum(r[2] for r in rows)
    print("| kernel | calls | total ms | avg us | share of kernel time | HBM read GB/launch (2 x FETCH_SIZE) | HBM write GB/launch | GB/s |")
    print("|---|---|---|---|---|---|---|---|")
    for k, calls, ms, avg in sorted(rows, key=lambda r: -r[2]):
        f, w = fetch.get(k), write.get(k)
        if f is not None or w is not None:
            rd, wr = 2 * (f or 0) * 1024 / 1e9, (w or 0) * 1024 / 1e9
            print("| %s | %d | %.3f | %.1f | %.1f%% | %.2f | %.2f | %.0f |" % (k, calls, ms, avg, 100 * ms / tot, rd, wr, (rd + wr) / (avg * 1e-6)))
        else:
            print("| %s | %d | %.3f | %.1f | %.1f%% | - | - | - |" % (k, calls, ms, avg, 100 * ms / tot))
    print("\nsum over svx kernels: %.3f ms" % tot)


if __name__ == "__main__":
    main()

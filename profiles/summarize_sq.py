#!/usr/bin/env python3
"""Per-kernel wave-cycle breakdown and LDS bank-conflict rate from a rocprofv3 --pmc SQ_* pass (counter_collection.csv).
    python profiles/summarize_sq.py <counter_collection.csv>"""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?(k_[a-z0-9_]+(?:<[^>]*>)?)", name)
    if not m:
        continue
    k = m.group(1)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        cnt[k] += 1
print("| kernel | launches | wave quad-cycles / launch | waiting (s_waitcnt, barrier) | issue-stalled | issuing | LDS conflict cycles / LDS cycles |")
print("|---|---|---|---|---|---|---|")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = c.get("SQ_WAVE_CYCLES", 0)
    if w <= 0:
        continue
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0)
    print("| %s | %d | %.3g | %.0f%% | %.0f%% | %.0f%% | %s |" % (
        k, cnt[k], w / max(1, cnt[k]), 100 * c.get("SQ_WAIT_ANY", 0) / w, 100 * c.get("SQ_WAIT_INST_ANY", 0) / w,
        100 * c.get("SQ_ACTIVE_INST_ANY", 0) / w, ("%.1f%%" % (100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / lds)) if lds > 0 else "-"))

#!/usr/bin/env python3
"""Keep only this repo's kernels (k_*) and the useful columns of a rocprofv3 *_counter_collection.csv.
    python profiles/trim_counters.py in.csv out.csv"""
import csv
import sys

KEEP = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count",
        "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "::k_" in r["Kernel_Name"] or r["Kernel_Name"].startswith("k_")]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, KEEP, extrasaction="ignore", quoting=csv.QUOTE_MINIMAL)
    w.writeheader()
    w.writerows(rows)
print(len(rows), "rows")

#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv -> per-launch sums for the kernels whose name contains <substr>:
python scripts/pmc_per_launch.py <p_counter_collection.csv> <substr>"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
agg = collections.defaultdict(float)
for r in rows:
    agg[r["Counter_Name"]] += float(r["Counter_Value"])
n_disp = len(set(r["Dispatch_Id"] for r in rows))
k0 = rows[0]["Kernel_Name"][:70] if rows else "?"
for k, v in agg.items():
    print("%-26s %.5g   per launch (%d launches, %s)" % (k, v / max(1, n_disp), n_disp, k0))

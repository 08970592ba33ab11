#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter CSV per kernel name: launches, mean counter value per launch."""
import csv
import glob
import sys
from collections import defaultdict

root, counter = sys.argv[1], sys.argv[2]
files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
agg = defaultdict(lambda: [0, 0.0])
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
            a = agg[name]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
print(f"# {counter}: kernel, launches, mean per launch, total")
for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{name[:60]:60s} {n:7d} {tot / n:16.1f} {tot:18.1f}")

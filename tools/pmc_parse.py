#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv files per kernel (FT kernels only)."""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
res = collections.defaultdict(dict)
for f in sorted(glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)):
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ft_" not in k:
            continue
        k = re.search(r"(ft_\w+)", k).group(1)
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for (k, c), v in acc.items():
        res[k][c] = round(v / n[(k, c)])
for k, v in res.items():
    print(k)
    for c in sorted(v):
        print(f"   {c:28s} {v[c]:>14,d}")

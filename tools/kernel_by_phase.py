#!/usr/bin/env python3
"""A kernel's durations in the forward-only spin-up passes and in the training steps of one bench.py trace.
usage: tools/kernel_by_phase.py <kernel_trace.csv> <kernel name substring>"""
import csv
import statistics
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first_step = next(i for i, r in enumerate(rows) if "ft_bwd" in r["Kernel_Name"])
for name, part in (("spin-up", rows[:first_step]), ("steps", rows[first_step:])):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in part if sys.argv[2] in r["Kernel_Name"]]
    if d:
        print(f"{name:8s} n={len(d):4d} median {statistics.median(d):7.1f} us  min {min(d):7.1f}  max {max(d):7.1f}")

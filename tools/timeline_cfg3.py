#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace as a timeline: start (us), duration, gap to the previous kernel's end, queue.
usage: tools/timeline_cfg3.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# a step ends with the optimiser update: adamw_kernel, or the merge kernel that carries it (bt_net_train_step)
ad = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
if len(ad) < 2:
    ad = [i for i, r in enumerate(rows) if "ft_bwd_merge" in r["Kernel_Name"]]
a, b = ad[-2], ad[-1]
t0 = prev_end = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap={(s - prev_end) / 1e3:6.1f} q={r.get('Queue_Id', '?')} {name}")
    prev_end = max(prev_end, e)
print("step", (int(rows[b]["End_Timestamp"]) - t0) / 1e3)

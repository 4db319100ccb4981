#!/usr/bin/env python3
"""Developer aid: one line per bench.py output file -- step time and the live per-kernel table (base names)."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError) as e:
        print(path, "unreadable:", e)
        continue
    k = {n.split("<")[0].replace("_kernel", ""): round(v, 1) for n, v in d.get("kernels_us_per_step_warmup", {}).items()}
    print(f"{path}: {d['ms_per_step'] * 1000:.1f} us/step  {d['value'] / 1e6:.2f} M/s  frac {d['roofline']['frac']:.3f}  {k}")

#!/usr/bin/env python3
"""Developer experiment: per-phase cycle counts of ft_bwd_oct (workgroup 0, one row per wave), from a library built
with -DBT_OCT_TIMING (BT_LIB_PATH=tools/libbullet_amd_timing.so python tools/oct_timing.py [occupancy])."""
import ctypes as C
import os
import runpy
import sys

sys.argv = [sys.argv[0], "3"] + sys.argv[1:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runpy.run_path(os.path.join(ROOT, "tools", "ft_microbench.py"), run_name="__main__")
from bullet_amd import lib  # noqa: E402
L = lib.load()
buf = (C.c_ulonglong * 128)()
assert L.bt_oct_debug_read(buf) == 0
names = ["pre", "loop_tail", "waitA", "stage", "waitB", "prefetch", "reduce", "end"]
print("wave " + " ".join(f"{n:>9s}" for n in names))
for w in range(16):
    print(f"{w:4d} " + " ".join(f"{buf[w * 8 + i] / 3:9.0f}" for i in range(8)))

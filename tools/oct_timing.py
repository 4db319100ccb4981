#!/usr/bin/env python3
"""Developer experiment: per-phase cycle counts of ft_bwd_oct (workgroup 0, one row per wave) over the bench's
training step, from a library built with -DBT_OCT_TIMING:
BT_LIB_PATH=tools/libbullet_amd_timing.so python tools/oct_timing.py [bench args]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (before the library: both must end up on torch's HIP runtime)
import bench  # noqa: E402
from bullet_amd import lib  # noqa: E402

STEPS, WARM = 20, 2
sys.argv = [sys.argv[0], "--steps", str(STEPS), "--warmup", str(WARM), "--no-cpu-baseline", "--no-profile",
            "--no-subrecords"] + sys.argv[1:]
L = lib.load()
buf = (C.c_ulonglong * 128)()
bench.main()
assert L.bt_oct_debug_read(buf) == 0
# phases of ft_bwd_oct_body (OT(i)): before the loop | loop tail -> top | wait at the first chunk barrier | first stream
# rows + fix-up + staging | wait at the second barrier | issuing the next chunk's loads | the reduction | after the loop
names = ["prologue", "looptail", "waitA", "stage", "waitB", "prefetch", "reduce", "end"]
print("cycles per launch; wave " + " ".join(f"{n:>9s}" for n in names))
for w in range(16):
    print(f"{w:4d} " + " ".join(f"{buf[w * 8 + i] / (STEPS + WARM):9.0f}" for i in range(8)))

#!/usr/bin/env python3
"""Developer experiment: per-phase cycle counts of ft_bwd_lds (workgroup 0, thread 0) from a -DBT_EXP_TIMING build."""
import ctypes as C, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BT_LIB_PATH"] = os.path.join(ROOT, "bullet_amd", "exp", "lib_timing.so")
import runpy
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-profile"] + sys.argv[1:]
import torch
torch.cuda.init()
torch.zeros(1, device='cuda')
from bullet_amd import lib
L = lib.load()
buf = (C.c_ulonglong * 8)()
L.bt_debug_read(buf)
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
L.bt_debug_read(buf)
names = ["prologue", "top barrier wait", "staging (LDS writes, blkf)", "barrier 2 wait", "prefetch issue", "reduce loop", "phase2+epilogue-pre", "final barrier"]
tot = sum(buf)
for n, v in zip(names, buf):
    print(f"{n:28s} {v/25:12.0f} cycles/launch  {100*v/tot:5.1f}%")

if hasattr(L, "bt_debug_read2"):
    b2 = (C.c_ulonglong * 16)()
    L.bt_debug_read2(b2)
    print("reduce-loop cycles per launch by wave:", [int(v / 25) for v in b2])

#!/usr/bin/env python3
"""Developer tool: phase timing inside ft_bwd_oct from the shader-clock stamps of a -DBT_STAMPS build
(tools/build_variant_of.sh stamps ft.hip -DBT_STAMPS; BT_LIB_PATH=tools/libbullet_amd_stamps.so python3 tools/stamps_bwd.py).
Prints, per phase, the cycles between consecutive stamps: mean over chunks and waves, and the mean of the per-chunk
fastest / slowest wave."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--occupancy", default="full")
ap.add_argument("--net", default="cfg2")
ap.add_argument("--names", default="top,barrier1,write+fixup,barrier2,stage_first,reduce,rows+stage_last")
a = ap.parse_args()
args = argparse.Namespace(upload="records", spinup_ms=50.0, no_hints=False, no_profile=True, graphs="off")
import torch  # noqa: E402
from bullet_amd import lib  # noqa: E402

torch.cuda.set_device(0)
L = lib.load()
res = bench.run_case(args, a.net, a.occupancy, 16384, 16384, 0, None, 30, 5)
torch.cuda.synchronize()
WGS, WAVES, CH, N = 8, 16, 20, 8
buf = np.zeros((WGS, WAVES, CH, N), dtype=np.uint64)
lib.check(L.bt_internal_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)))
res["release"]()
st = buf.astype(np.int64)
nch = int((st[0, 4, :, 0] != 0).sum())
print(f"chunks per workgroup {nch}; cycles per chunk: {np.diff(st[:, 4:, :nch, 0], axis=2).mean():.0f}")
full = st[:, 4:, 1:nch - 1, :]  # reducer waves, steady-state chunks
t0 = full[..., 1].min(axis=1, keepdims=True)
print("reducer timeline (cycles after the first barrier opens): arrive A | leave A | fixup+meta done | leave B | rows done | next rows sent")
for w in range(12):
    rel = (full[:, w, :, :] - t0[:, 0, :, None]).mean(axis=(0, 1))
    print(f"  wave {w + 4:2d}: " + " ".join(f"{rel[i]:7.0f}" for i in (0, 1, 2, 3, 5, 6)))
sg = st[:, :4, 1:nch - 1, :]
print("stager timeline: arrive A | leave A | tile written | leave B | formed | next request sent")
for w in range(4):
    rel = (sg[:, w, :, :] - t0[:, 0, :, None]).mean(axis=(0, 1))
    print(f"  wave {w:2d}: " + " ".join(f"{rel[i]:7.0f}" for i in (0, 1, 2, 3, 4, 5)))
d = full[..., 5] - full[..., 3]
print(f"reduce: mean {d.mean():.0f}  fastest wave {d.min(axis=1).mean():.0f}  slowest wave {d.max(axis=1).mean():.0f}")

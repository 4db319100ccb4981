#!/usr/bin/env python3
"""Runs only the FT forward / backward operators of config 2 (for rocprofv3 --pmc passes).
env BT_MB_PAIR=1: pairwise table; BT_MB_NSEG=n: n king buckets (cols = 768 n) with the segment promise."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bullet_amd import lib  # noqa: E402

L = lib.load()
B, H, F, N = int(os.environ.get("BT_MB_BATCH", "16384")), 1024, 768, 32
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
occ = sys.argv[2] if len(sys.argv) > 2 else "full"
PAIR, NSEG = int(os.environ.get("BT_MB_PAIR", "0")), int(os.environ.get("BT_MB_NSEG", "1"))
rec = bench.synth_records(B, 0xB011E7002, occ)
if NSEG > 1:
    idesc = lib.make_input_desc("chess_buckets", False, [i * NSEG // 64 for i in range(64)])
    F = 768 * NSEG
else:
    idesc = lib.make_input_desc("chess768")
stm, ntm, bk, tg, w = lib.map_batch(idesc, rec, out_buckets=8, blend=0.75, threads=8)
OW = H // 2 if PAIR else H
d = lib.FtDesc(batch=B, rows=H, cols=F, stride=H, offset=0, nnz=N, act=lib.ACT["crelu" if PAIR else "screlu"],
               pairwise=PAIR, n_inputs=2, out_ld=2 * OW, segment=768 if NSEG > 1 else 0)
rng = np.random.default_rng(0)
W = torch.from_numpy((rng.standard_normal(F * H) * 0.05).astype(np.float32)).cuda()
bias = torch.full((H,), 0.1, device="cuda")
i0, i1 = torch.from_numpy(stm).cuda(), torch.from_numpy(ntm).cuda()
out = torch.zeros(B, 2 * OW, device="cuda")
saved = torch.zeros(2, B, H, device="cuda") if PAIR else None
sp = saved.data_ptr() if PAIR else None
go = torch.randn(B, 2 * OW, device="cuda")
ptrs = (C.c_void_p * 2)(i0.data_ptr(), i1.data_ptr())
wsb = L.bt_ft_backward_workspace_bytes(C.byref(d))
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
dw, db = torch.zeros(F * H, device="cuda"), torch.zeros(H, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(iters):
    lib.check(L.bt_ft_build_index(s, C.byref(d), ptrs, ws.data_ptr(), wsb))
    lib.check(L.bt_ft_forward_prebuilt(s, C.byref(d), W.data_ptr(), bias.data_ptr(), ptrs, out.data_ptr(), sp,
                                       ws.data_ptr(), wsb))
    lib.check(L.bt_ft_backward_prebuilt(s, C.byref(d), go.data_ptr(), out.data_ptr(), sp, dw.data_ptr(), db.data_ptr(),
                                        ws.data_ptr(), wsb))
torch.cuda.synchronize()
print("done", float(out.sum()), float(dw.abs().sum()))

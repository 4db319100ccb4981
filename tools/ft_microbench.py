#!/usr/bin/env python3
"""Runs only the FT forward / backward operators of config 2 (for rocprofv3 --pmc passes)."""
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
B, H, F, N = 16384, 1024, 768, 32
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
occ = sys.argv[2] if len(sys.argv) > 2 else "full"
rec = bench.synth_records(B, 0xB011E7002, occ)
stm, ntm, bk, tg, w = lib.map_batch(lib.make_input_desc("chess768"), rec, out_buckets=8, blend=0.75, threads=8)
d = lib.FtDesc(batch=B, rows=H, cols=F, stride=H, offset=0, nnz=N, act=lib.ACT["screlu"], pairwise=0, n_inputs=2,
               out_ld=2 * H)
rng = np.random.default_rng(0)
W = torch.from_numpy((rng.standard_normal(F * H) * 0.05).astype(np.float32)).cuda()
bias = torch.full((H,), 0.1, device="cuda")
i0, i1 = torch.from_numpy(stm).cuda(), torch.from_numpy(ntm).cuda()
out = torch.zeros(B, 2 * H, device="cuda")
go = torch.randn(B, 2 * H, device="cuda")
ptrs = (C.c_void_p * 2)(i0.data_ptr(), i1.data_ptr())
wsb = L.bt_ft_backward_workspace_bytes(C.byref(d))
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
dw, db = torch.zeros(F * H, device="cuda"), torch.zeros(H, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(iters):
    lib.check(L.bt_ft_forward(s, C.byref(d), W.data_ptr(), bias.data_ptr(), ptrs, out.data_ptr(), None))
    lib.check(L.bt_ft_backward(s, C.byref(d), ptrs, go.data_ptr(), out.data_ptr(), None, dw.data_ptr(), db.data_ptr(),
                               ws.data_ptr(), wsb))
torch.cuda.synchronize()
print("done", float(out.sum()), float(dw.abs().sum()))

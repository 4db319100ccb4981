#!/usr/bin/env python3
"""Pinned host -> HBM copy rate at the batch sizes of the training loop (one mapped batch 4.4 MB, one batch of
records 0.5 MB), through the C ABI: what PCIe gives the loader on this box."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bullet_amd import lib

L = lib.load()
torch.zeros(1, device="cuda")
for nbytes in (16384 * 32, 16384 * 268, 64 << 20):
    hp = C.c_void_p()
    lib.check(L.bt_host_alloc_pinned(C.byref(hp), nbytes))
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.bt_memcpy_h2d_async(dev.data_ptr(), hp, nbytes, s)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        L.bt_memcpy_h2d_async(dev.data_ptr(), hp, nbytes, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{nbytes / 1e6:8.2f} MB  {dt * 1e6:8.1f} us  {nbytes / dt / 1e9:6.1f} GB/s")
    L.bt_host_free_pinned(hp)

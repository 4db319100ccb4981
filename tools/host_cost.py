#!/usr/bin/env python3
"""Host-side cost of the calls one training step enqueues (is the loop GPU-bound or launch-bound?)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bullet_amd import lib, parallel

L = lib.load()
torch.zeros(1, device="cuda")
B = 16384


def handle(create):
    h = C.c_void_p()
    lib.check(create(C.byref(h)))
    return h


def timeit(name, fn, reps=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:34s} enqueue {1e6 * (t1 - t0) / reps:8.1f} us/call   drained after {1e6 * (t2 - t0) / reps:8.1f} us/call")


copy, ev, ev2 = handle(L.bt_stream_create), handle(L.bt_event_create), handle(L.bt_event_create)
s = torch.cuda.current_stream().cuda_stream
own = handle(L.bt_stream_create)
for nbytes in (B * 32, B * 268):
    hp = C.c_void_p()
    lib.check(L.bt_host_alloc_pinned(C.byref(hp), nbytes))
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    timeit(f"h2d {nbytes >> 10} KiB on copy stream", lambda: L.bt_memcpy_h2d_async(dev.data_ptr(), hp, nbytes, copy))
timeit("event record (copy stream)", lambda: L.bt_event_record(ev, copy))
timeit("stream wait event (null stream)", lambda: L.bt_stream_wait_event(s, ev))
timeit("event record (null stream)", lambda: L.bt_event_record(ev2, s))
timeit("stream wait event (copy stream)", lambda: L.bt_stream_wait_event(copy, ev2))

netcfg, kind, layout, _ = bench.NETS["cfg2"]
idesc = lib.make_input_desc(kind, False, layout)
rec = bench.synth_records(B, 1234, "full")
mapped = lib.map_batch(idesc, rec, out_buckets=8, blend=0.75, eval_scale=400.0, threads=8)
net = lib.Net(lib.make_net_desc(**netcfg), B)
net.init_weights()
batch = net.make_batch(*[torch.from_numpy(a).cuda() for a in mapped])
adj = torch.tensor([1.0 / B], device="cuda")
lr = torch.tensor([1e-3], device="cuda")
for stream, label in ((s, "null stream"), (own.value, "own stream")):
    def step():
        net.forward_backward(batch, stream)
        net.optimiser_step(adj, lr, stream)
    for _ in range(10):
        step()
    timeit(f"fwd+bwd+adamw ({label})", step)
    timeit(f"fwd+bwd only ({label})", lambda: net.forward_backward(batch, stream))
    timeit(f"adamw only ({label})", lambda: net.optimiser_step(adj, lr, stream))

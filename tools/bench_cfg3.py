#!/usr/bin/env python3
"""Times BASELINE configs[2] (HalfKA, 16 king buckets -> 1024x2 pairwise -> 16 -> 32 -> 1, 8 output buckets) at batch
16384 (argv: H, segment): a parity-test configuration, reported for information only."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bullet_amd import lib  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
SEG = int(sys.argv[2]) if len(sys.argv) > 2 else 768  # 0 = without the segment promise (any-shape FT kernels)
B = 16384
layout = [i // 4 for i in range(64)]  # 16 distinct king buckets
idesc = lib.make_input_desc("chess_buckets", False, layout)
if len(sys.argv) > 3:  # real positions (bulletformat file) instead of the synthetic generator: skewed king buckets
    allrec = np.fromfile(sys.argv[3], dtype=np.uint8).reshape(-1, 32)
    rec = allrec[np.arange(B) % allrec.shape[0]].reshape(-1).copy()
else:
    rec = bench.synth_records(B, 0xB011E7003, "full")
stm, ntm, bk, tg, w = lib.map_batch(idesc, rec, out_buckets=8, blend=0.75, threads=8)
desc = lib.make_net_desc(num_inputs=768 * 16, nnz=32, ft_out=H, ft_act="crelu", pairwise=True, num_buckets=8,
                         dense=[(16, "screlu", True), (32, "screlu", True), (1, "identity", True)], ft_segment=SEG)
net = lib.Net(desc, B)
net.init_weights()
tb = net.make_batch(*[torch.from_numpy(a).cuda() for a in (stm, ntm, bk, tg, w)])
adj, lr = torch.tensor([1.0 / B], device="cuda"), torch.tensor([1e-3], device="cuda")
for _ in range(5):
    net.forward_backward(tb)
    net.optimiser_step(adj, lr)
torch.cuda.synchronize()
lib.profile_start(4096)
t0 = time.perf_counter()
K = 20
HINT = os.environ.get("BT_NO_HINT", "0") != "1"  # bt_net_hint_next_batch: next step's index built during this one
for _ in range(K):
    if HINT:
        net.hint_next(tb)
    net.forward_backward(tb)
    net.optimiser_step(adj, lr)
torch.cuda.synchronize()
el = time.perf_counter() - t0
prof = lib.profile_stop()
print(f"cfg3 H={H} segment={SEG}: {B * K / el / 1e6:.2f} M pos/s, {el / K * 1e3:.3f} ms/step, loss {net.loss.item() / B:.6f}")
print({k: round(1e3 * ms / K, 1) for k, (c, ms) in prof.items()})

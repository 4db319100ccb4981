"""Stress run of the two-rank data-parallel training test (tests/test_gpu_dp.py) -- N repetitions with both mappers;
prints which parameters differ between the replicas if they ever do. Found the NULL-stream zero-fill race of bt_malloc.
Usage (GPU box): python tools/dp_stress.py 40"""
import sys, os, socket
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
import torch.multiprocessing as mp
import test_gpu_dp as t
def once(dm):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=t._worker, args=(r, 2, port, dm, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=100) for _ in range(2)], key=lambda x: x[0])
    [p.join(timeout=60) for p in procs]
    (_, l0, sb0, p0), (_, l1, sb1, p1) = res
    if not np.array_equal(p0, p1):
        d = np.nonzero(p0 != p1)[0]
        print("MISMATCH dm", dm, "count", d.size, "first", d[:5], "last", d[-5:], "p0", p0[d[:3]], "p1", p1[d[:3]], "losses equal", l0 == l1, flush=True)
        return 1
    return 0
if __name__ == "__main__":
    bad = 0
    for i in range(int(sys.argv[1])):
        bad += once(True)
        bad += once(False)
    print("runs", sys.argv[1], "bad", bad)

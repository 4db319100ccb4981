"""World-size-2 gloo test (CPU) of the data-parallel arithmetic used by bench.py --gpus N:
shard the batch, compute unscaled gradient sums per rank (the CPU oracle stands in for the device step),
all-reduce the flat gradient + loss, AdamW with adj = 1/global_batch -> same weights on every rank and equal
(within fp32 summation-order tolerance) to the single-process result on the whole batch."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
from bullet_amd import lib, parallel

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "batch.bf")
CFG = dict(num_inputs=768, nnz=32, ft_out=64, ft_act="screlu", num_buckets=8, dense=[(1, "identity", True)])
N_POS, STEPS = 1001, 3  # odd on purpose: uneven shards


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(arrs):
    return np.concatenate([a.reshape(-1) for a in arrs])


def _unflat(flat, like):
    out, o = [], 0
    for a in like:
        out.append(flat[o:o + a.size].reshape(a.shape).copy())
        o += a.size
    return out


def _load():
    rec = np.fromfile(GOLDEN, dtype=np.uint8)[:32 * N_POS]
    return lib.map_batch(lib.make_input_desc("chess768"), rec, out_buckets=8, blend=0.5, threads=2)


def _train(rank, world, use_dist):
    stm, ntm, bk, tg, w = _load()
    lo, hi = parallel.shard_range(N_POS, rank, world)
    d = ob.make_net(**CFG)
    params = ob.net_init(d)
    m = [np.zeros_like(p) for p in params]
    v = [np.zeros_like(p) for p in params]
    losses = []
    for _ in range(STEPS):
        loss, grads, _ = ob.net_fwd_bwd(d, params, stm[lo:hi].copy(), ntm[lo:hi].copy(), bk[lo:hi].copy(),
                                        tg[lo:hi].copy(), w[lo:hi].copy())
        buf = torch.from_numpy(np.concatenate([_flat(grads), np.array([loss], np.float32)]))  # Net.step_buffer layout
        parallel.allreduce_step_buffers(dist if use_dist else None, buf)
        flat, lsum = buf[:-1], buf[-1:]
        grads = _unflat(flat.numpy(), grads)
        for p_, m_, v_, g_ in zip(params, m, v, grads):
            ob.adamw(p_, m_, v_, g_, parallel.adj_factor(N_POS), 1e-3)
        losses.append(float(lsum) / N_POS)
    return params, losses


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params, losses = _train(rank, world, True)
    q.put((rank, [p.copy() for p in params], losses))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_batch():
    for n, w in [(16384, 8), (1001, 2), (7, 4), (3, 8)]:
        r = [parallel.shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_two_rank_data_parallel_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    ref_params, ref_losses = _train(0, 1, False)
    (_, p0, l0), (_, p1, l1) = res
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b)  # replicas bit-identical
    assert l0 == l1
    for a, b in zip(l0, ref_losses):
        assert abs(a - b) <= 1e-5 * abs(b)
    for t, (a, b) in enumerate(zip(p0, ref_params)):
        assert np.allclose(a, b, rtol=1e-4, atol=2e-6), f"tensor {t}"

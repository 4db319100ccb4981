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


def _plan(world, rank, shard_min):
    """bt_net_dp_plan through the C ABI (host arithmetic, no GPU) + the flat layout it refers to"""
    import ctypes as C
    L = lib.load()
    nd = lib.make_net_desc(**CFG)
    pl = lib.DpPlan()
    lib.check(L.bt_net_dp_plan(C.byref(nd), world, rank, shard_min, C.byref(pl)))
    nt = L.bt_net_num_tensors(C.byref(nd))
    sizes = [L.bt_net_tensor_size(C.byref(nd), t) for t in range(nt)]
    offs = [L.bt_net_tensor_offset(C.byref(nd), t) for t in range(nt)]
    return pl, sizes, offs, L.bt_net_param_count(C.byref(nd))


def _train_sharded(rank, world):
    """The step bt_net_dp_step takes when the plan shards the table, restated with torch.distributed on the CPU:
    all-reduce of the tail (other tensors + loss slot), reduce-scatter of the table gradient, AdamW on the own
    slice and on the tail, all-gather of the table; momentum / velocity of foreign slices are never touched."""
    stm, ntm, bk, tg, w = _load()
    lo, hi = parallel.shard_range(N_POS, rank, world)
    pl, sizes, offs, n_flat = _plan(world, rank, 1024)
    assert pl.sharded and pl.table_n == sizes[0] and pl.per_rank * world == pl.table_n and pl.tail_off == offs[1]
    assert pl.tail_off + pl.tail_n == n_flat + 1 and pl.slice_lo == rank * pl.per_rank
    d = ob.make_net(**CFG)
    flat = {k: np.zeros(n_flat + 1, np.float32) for k in "pmvg"}
    for t, a in enumerate(ob.net_init(d)):
        flat["p"][offs[t]:offs[t] + sizes[t]] = a.reshape(-1)
    views = lambda k: [flat[k][offs[t]:offs[t] + sizes[t]] for t in range(len(sizes))]
    shapes = [a.shape for a in ob.net_init(d)]
    losses = []
    for _ in range(STEPS):
        params = [v.reshape(sh) for v, sh in zip(views("p"), shapes)]
        loss, grads, _ = ob.net_fwd_bwd(d, params, stm[lo:hi].copy(), ntm[lo:hi].copy(), bk[lo:hi].copy(),
                                        tg[lo:hi].copy(), w[lo:hi].copy())
        for t, g in enumerate(grads):
            flat["g"][offs[t]:offs[t] + sizes[t]] = g.reshape(-1)
        flat["g"][n_flat] = loss
        tail = torch.from_numpy(flat["g"][pl.tail_off:pl.tail_off + pl.tail_n])
        dist.all_reduce(tail)
        for r in range(world):  # reduce-scatter (gloo has none): chunk r is reduced onto rank r
            dist.reduce(torch.from_numpy(flat["g"][r * pl.per_rank:(r + 1) * pl.per_rank]), dst=r)
        adj = parallel.adj_factor(N_POS)
        for t in range(1, len(sizes)):
            ob.adamw(views("p")[t], views("m")[t], views("v")[t], views("g")[t], adj, 1e-3)
        own = slice(pl.slice_lo, pl.slice_lo + pl.per_rank)
        ob.adamw(flat["p"][own], flat["m"][own], flat["v"][own], flat["g"][own], adj, 1e-3)
        parts = [torch.empty(pl.per_rank) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(flat["p"][own].copy()))
        flat["p"][:pl.table_n] = torch.cat(parts).numpy()
        losses.append(float(flat["g"][n_flat]) / N_POS)
    # bt_net_dp_gather_state: the owners' momentum / velocity slices, everywhere
    for k in "mv":
        parts = [torch.empty(pl.per_rank) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(flat[k][own].copy()))
        flat[k][:pl.table_n] = torch.cat(parts).numpy()
    return [v.reshape(sh).copy() for v, sh in zip(views("p"), shapes)], losses, \
        [v.copy() for v in views("m")], [v.copy() for v in views("v")]


def _worker(rank, world, port, q, sharded=False):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if sharded:
        q.put((rank,) + _train_sharded(rank, world) + (_train(rank, world, True),))
        dist.barrier()
        dist.destroy_process_group()
        return
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


def test_dp_plan_small_table_is_one_all_reduce():
    pl, sizes, offs, n_flat = _plan(8, 3, 0)  # 768 x 64 floats: far below the default threshold
    assert not pl.sharded and pl.tail_off == 0 and pl.tail_n == n_flat + 1
    pl, *_ = _plan(7, 3, 1024)  # 49152 does not divide by 7
    assert not pl.sharded


def test_two_rank_sharded_optimiser_matches_replicated_and_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, True)) for r in range(world)]
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
    (_, p0, l0, m0, v0, rep0), (_, p1, l1, m1, v1, _) = res
    for a, b in zip(p0 + m0 + v0, p1 + m1 + v1):
        assert np.array_equal(a, b)  # replicas bit-identical, optimiser state included after the gather
    assert l0 == l1
    # a sum of two terms does not depend on who adds them: the sharded step writes the replicated step's bits
    rep_params, rep_losses = rep0
    assert l0 == rep_losses
    for a, b in zip(p0, rep_params):
        assert np.array_equal(a, b)
    ref_params, ref_losses = _train(0, 1, False)
    for a, b in zip(l0, ref_losses):
        assert abs(a - b) <= 1e-5 * abs(b)
    for t, (a, b) in enumerate(zip(p0, ref_params)):
        assert np.allclose(a, b, rtol=1e-4, atol=2e-6), f"tensor {t}"

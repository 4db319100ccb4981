"""-m gpu: data-parallel bt_train_run (bt_train_config.dp_*). Two ranks share the one GPU of the test box and
exchange the step buffer over gloo (RCCL needs one GPU per rank; the driver's multi-GPU runs use "nccl"): each rank
trains on its shard of every global batch, the replicas must stay bit-identical and match -- within fp32
summation-order tolerance -- the single-process run over the whole batch. The reference has no multi-device
path; the single-process bt_train_run is itself checked against the reference's printed losses in
test_gpu_net.py."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch.bf")
CFG = dict(num_inputs=768, nnz=32, ft_out=64, ft_act="screlu", num_buckets=8, dense=[(1, "identity", True)])
# a king-bucketed multi-layer net (HalfKA-style): segment-grouped FT kernels, bucket-sorted MFMA layer, fused tail and
# the next-batch hints of the training loop, all under the data-parallel exchange
LAYOUT = [0, 0, 1, 1] * 2 + [2] * 8 + [3] * 16
CFG_MULTI = dict(num_inputs=768 * 4, nnz=32, ft_out=128, ft_act="crelu", pairwise=True, num_buckets=8, ft_segment=768,
                 dense=[(16, "screlu", True), (32, "screlu", True), (1, "identity", True)])
BATCH, BPS, SBS = 1001, 3, 2  # odd batch: uneven shards


def _run(dist, device_mapper, kind="single", mode="dist", comm=None):
    """mode "dist": the grad_sync hook over torch.distributed; "comm" / "sharded": a bt_comm (bt_net_dp_step), the
    second with the table forced onto the sharded path (reduce-scatter, optimiser on the own slice, all-gather);
    "sharded_own": the same with the collectives on a stream of the communicator's own (bt_comm_use_own_stream), i.e.
    the event-fenced sequence a native RCCL communicator takes, here with real summation between two ranks"""
    from bullet_amd import lib, parallel
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    if dist is not None and mode != "dist":
        comm, dist = parallel.Comm.from_torch(dist, own_stream=mode == "sharded_own"), None
    lo, hi = parallel.shard_range(BATCH, rank, world)
    multi = kind == "multi"
    net = lib.Net(lib.make_net_desc(**(CFG_MULTI if multi else CFG)), hi - lo)
    net.init_weights(seed=11, ft_effective_inputs=32 if multi else 0)
    losses = []
    idesc = lib.make_input_desc("chess_buckets_mirrored", False, LAYOUT) if multi else lib.make_input_desc("chess768")
    rep = net.train(idesc, [GOLDEN], BATCH, BPS, SBS,
                    lib.LrSched(lib.LR["constant"], 0.001, 0, 0, 0), lib.WdlSched(lib.WDL["constant"], 0.3, 0, 0),
                    out_buckets=8, threads=2, on_batch=lambda sb, b, l: losses.append(l),
                    device_mapper=device_mapper, dist=dist, comm=comm, shard_min_floats=1024 if mode.startswith("sharded") else 0)
    params = net.params.detach().cpu().numpy().copy()
    if mode.startswith("sharded"):  # the optimiser state was gathered before bt_train_run returned: part of the comparison
        n = net.n_flat
        state = [parallel.device_view(f(net.h), n).cpu().numpy().copy() for f in (net.L.bt_net_momentum, net.L.bt_net_velocity)]
        params = np.concatenate([params] + state)
    net.close()
    return losses, [r[1] for r in rep], params


def _worker(rank, world, port, device_mapper, q, kind="single", mode="dist"):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      GLOO_SOCKET_IFNAME="lo")  # the box's hostname may not resolve
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _run(dist, device_mapper, kind, mode))
    except Exception as e:  # report instead of hanging the parent on q.get
        q.put((rank, repr(e), None, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("device_mapper,kind,mode", [(False, "single", "dist"), (True, "single", "dist"),
                                                     (True, "multi", "dist"), (True, "single", "comm"),
                                                     (False, "single", "sharded"), (True, "multi", "sharded"),
                                                     (True, "single", "sharded_own"), (True, "multi", "sharded_own")])
def test_two_rank_train_run_matches_single_process(device_mapper, kind, mode):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, device_mapper, q, kind, mode)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=100) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:  # never leave a rank behind (a stuck rendezvous would otherwise outlive the test run)
        for p in procs:
            if p.is_alive():
                p.kill()
    for r in res:
        assert not isinstance(r[1], str), r[1]
    (_, l0, sb0, p0), (_, l1, sb1, p1) = res
    assert l0 == l1 and sb0 == sb1 and np.array_equal(p0, p1)  # replicas bit-identical
    ref_l, ref_sb, ref_p = _run(None, device_mapper, kind, mode)
    assert len(l0) == BPS * SBS == len(ref_l)
    for a, b in zip(l0, ref_l):
        assert abs(a - b) <= 1e-5 * abs(b)
    assert np.allclose(p0, ref_p, rtol=1e-4, atol=2e-6)


def test_rccl_communicator_of_one_rank():
    """The native RCCL binding (bt_comm_unique_id / bt_comm_init_rank / the three collectives) on the one GPU of the
    test box: with a single rank every collective is the identity, and a training run through the communicator --
    once replicated, once with the table forced onto the sharded path -- must write the bits of the plain run.
    More ranks need one GPU each: the driver's 8-GPU bench is the first place RCCL moves data between devices."""
    import ctypes as C
    from bullet_amd import lib, parallel
    comm = parallel.Comm.rccl()
    try:
        L = lib.load()
        assert L.bt_comm_rank(comm.h) == 0 and L.bt_comm_world(comm.h) == 1
        x = torch.arange(4096, dtype=torch.float32, device="cuda") * 0.5
        want = x.clone()
        comm.all_reduce(x)
        s = torch.cuda.current_stream().cuda_stream
        y = torch.zeros_like(x)
        lib.check(L.bt_reduce_scatter_sum_f32(comm.h, s, x.data_ptr(), y.data_ptr(), x.numel()))
        z = torch.zeros_like(x)
        lib.check(L.bt_allgather_f32(comm.h, s, y.data_ptr(), z.data_ptr(), x.numel()))
        lib.check(L.bt_allreduce_sum_f32(comm.h, s, z.data_ptr(), 0))  # empty: accepted, nothing sent
        torch.cuda.synchronize()
        assert torch.equal(x, want) and torch.equal(y, want) and torch.equal(z, want)
        ref = _run(None, True, "multi", "sharded")
        for mode in ("comm", "sharded"):
            got = _run(None, True, "multi", mode, comm=comm)
            assert got[0] == ref[0] and got[1] == ref[1]
            assert np.array_equal(got[2], ref[2][:got[2].size])
    finally:
        comm.close()

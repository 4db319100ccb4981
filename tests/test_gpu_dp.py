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


def _run(dist, device_mapper, kind="single"):
    from bullet_amd import lib, parallel
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    lo, hi = parallel.shard_range(BATCH, rank, world)
    multi = kind == "multi"
    net = lib.Net(lib.make_net_desc(**(CFG_MULTI if multi else CFG)), hi - lo)
    net.init_weights(seed=11, ft_effective_inputs=32 if multi else 0)
    losses = []
    idesc = lib.make_input_desc("chess_buckets_mirrored", False, LAYOUT) if multi else lib.make_input_desc("chess768")
    rep = net.train(idesc, [GOLDEN], BATCH, BPS, SBS,
                    lib.LrSched(lib.LR["constant"], 0.001, 0, 0, 0), lib.WdlSched(lib.WDL["constant"], 0.3, 0, 0),
                    out_buckets=8, threads=2, on_batch=lambda sb, b, l: losses.append(l),
                    device_mapper=device_mapper, dist=dist)
    params = net.params.detach().cpu().numpy().copy()
    net.close()
    return losses, [r[1] for r in rep], params


def _worker(rank, world, port, device_mapper, q, kind="single"):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      GLOO_SOCKET_IFNAME="lo")  # the box's hostname may not resolve
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _run(dist, device_mapper, kind))
    except Exception as e:  # report instead of hanging the parent on q.get
        q.put((rank, repr(e), None, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("device_mapper,kind", [(False, "single"), (True, "single"), (True, "multi")])
def test_two_rank_train_run_matches_single_process(device_mapper, kind):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, device_mapper, q, kind)) for r in range(world)]
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
    ref_l, ref_sb, ref_p = _run(None, device_mapper, kind)
    assert len(l0) == BPS * SBS == len(ref_l)
    for a, b in zip(l0, ref_l):
        assert abs(a - b) <= 1e-5 * abs(b)
    assert np.allclose(p0, ref_p, rtol=1e-4, atol=2e-6)

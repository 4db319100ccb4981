#!/usr/bin/env python3
"""bench.py -- training positions/sec of the NNUE hot path on MI355X.

Workload = BASELINE.json configs[1]: Chess768 -> 1024x2 SCReLU -> 1 with 8 output buckets,
batch 16384 per GPU, AdamW; one "step" = fwd + bwd (+ gradient all-reduce when N > 1) + AdamW
over one batch of synthetic positions that is already resident in HBM.
N > 1: one process per GPU (torchrun), data parallel, gradients all-reduced with RCCL, weak scaling
(per-GPU batch fixed at 16384, global batch 16384*N, adj = 1/global batch as run.rs:130 does).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (timed live with HIP events on the kernels' own stream) and `cpu_baseline` (the CPU oracle,
scalar single-thread port of the reference semantics, timed on a bounded sample on this host).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LDS_PEAK_GBS = 150000.0  # MI355X_MICROARCH.md, LDS: ~150 TB/s aggregate for ds_read_b64/b128 at ~2.4 GHz


# ----------------------------------------------------------------------------- synthetic positions
def splitmix64_stream(seed, n):
    """n outputs of SplitMix64 seeded with `seed` (vectorised: state_i = seed + (i+1)*gamma)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


FIXTURE_HIST = None


def synth_records(count, seed, occupancy="full"):
    """32-byte bulletformat ChessBoard records: k distinct squares with
    exactly one own king (code 5) and one opponent king (13), other pieces uniform over the legal
    non-king codes, score ~ clamp(round(400 N(0,1)), +-2000), result uniform, ksq / opp_ksq derived."""
    per = 80
    r = splitmix64_stream(seed, count * per).reshape(count, per)
    if occupancy == "full":
        k = np.full(count, 32, dtype=np.int64)
    else:  # piece-count histogram of examples/tests/batch.bf (mean 16.2), sampled by inverse CDF
        global FIXTURE_HIST
        if FIXTURE_HIST is None:
            rec = np.fromfile(os.path.join(ROOT, "tests", "golden", "batch.bf"), dtype=np.uint8).reshape(-1, 32)
            occ = rec[:, :8].copy().view(np.uint64).reshape(-1)
            pop = np.array([bin(int(o)).count("1") for o in occ])
            FIXTURE_HIST = np.cumsum(np.bincount(pop, minlength=33) / pop.size)
        u = (r[:, 70] >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        k = np.clip(np.searchsorted(FIXTURE_HIST, u), 2, 32)
    # k distinct squares = the k smallest of 64 random keys
    keys = (r[:, :64] >> np.uint64(8)).astype(np.int64)
    order = np.argsort(keys, axis=1, kind="stable")                # random permutation of squares
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.arange(64)[None, :].repeat(count, 0), axis=1)
    on = rank < k[:, None]                                          # [count][64] occupied squares
    occ = (on.astype(np.uint64) << np.arange(64, dtype=np.uint64)[None, :]).sum(axis=1, dtype=np.uint64)
    # piece codes: squares with rank 0 / 1 are the two kings, others uniform over {0..4, 8..12}
    nonking = np.array([0, 1, 2, 3, 4, 8, 9, 10, 11, 12], dtype=np.uint8)
    code = nonking[(r[:, :64] & np.uint64(0xFF)).astype(np.int64) % 10]
    code = np.where(rank == 0, 5, np.where(rank == 1, 13, code)).astype(np.uint8)
    rec = np.zeros((count, 32), dtype=np.uint8)
    rec[:, :8] = occ.view(np.uint8).reshape(count, 8)
    # nibbles in ascending-square order of the occupied squares, low nibble first
    pos_in_list = np.cumsum(on, axis=1) - 1
    pcs = np.zeros((count, 16), dtype=np.uint8)
    rows, sqs = np.nonzero(on)
    li = pos_in_list[rows, sqs]
    np.add.at(pcs, (rows, li // 2), (code[rows, sqs].astype(np.uint8) << ((li % 2) * 4).astype(np.uint8)))
    rec[:, 8:24] = pcs
    # Box-Muller normal from two uniforms
    u1 = ((r[:, 64] >> np.uint64(11)).astype(np.float64) + 1.0) / float(1 << 53)
    u2 = (r[:, 65] >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    nrm = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    score = np.clip(np.round(400.0 * nrm), -2000, 2000).astype(np.int16)
    rec[:, 24:26] = score.view(np.uint8).reshape(count, 2)
    rec[:, 26] = (r[:, 66] % np.uint64(3)).astype(np.uint8)
    rec[:, 27] = np.argmax(rank == 0, axis=1).astype(np.uint8)          # own king square
    rec[:, 28] = (np.argmax(rank == 1, axis=1) ^ 56).astype(np.uint8)   # opponent king square ^ 56
    return rec.reshape(-1)


# ----------------------------------------------------------------------------- workload
CFG2 = dict(num_inputs=768, nnz=32, ft_out=1024, ft_act="screlu", num_buckets=8, dense=[(1, "identity", True)])


def halfka(h):  # BASELINE configs[2] / [3]: 16 king buckets -> h x 2 pairwise CReLU -> 16 -> 32 -> 1, 8 output buckets
    return dict(num_inputs=768 * 16, nnz=32, ft_out=h, ft_act="crelu", pairwise=True, num_buckets=8,
                dense=[(16, "screlu", True), (32, "screlu", True), (1, "identity", True)], ft_segment=768)


# name -> (net, input set, king-bucket layout, workload text). The default `cfg2` is the contract workload; the
# others are BASELINE's parity-test configurations, runnable with the same flags (single GPU or data parallel).
NETS = {
    "cfg2": (CFG2, "chess768", None,
             "BASELINE configs[1]: Chess768 -> 1024x2 SCReLU -> 1, 8 output buckets, AdamW, fwd+bwd+optimiser step"),
    "cfg3": (halfka(1024), "chess_buckets", [i // 4 for i in range(64)],
             "BASELINE configs[2]: HalfKA 16 king buckets -> 1024x2 pairwise -> 16 -> 32 -> 1, 8 output buckets, AdamW"),
    "cfg4": (halfka(1536), "chess_buckets", [i // 4 for i in range(64)],
             "BASELINE configs[3]: HalfKA 16 king buckets -> 1536x2 pairwise -> 16 -> 32 -> 1, 8 output buckets, AdamW"),
}


def algorithmic_bytes(kernel, B, nbar, P=2, H=1024, N=32, F=768, K=2048, O=8):
    """ALGORITHMIC bytes per launch (stated in DESIGN.md). nbar = mean active features."""
    ft_fwd = P * B * (nbar * H * 4 + N * 4 + H * 4) + H * 4
    ft_bwd = P * B * (nbar * H * 4 + 2 * H * 4 + N * 4) + F * H * 4
    dense_pass = (K + 1) * B * 4 + K * O * 4
    table = {
        "ft_fwd_generic": ft_fwd, "ft_fwd_lds": ft_fwd,
        "ft_bwd_gather": P * B * nbar * H * 4 + F * H * 4,          # the gather-sum part of ft_bwd
        "ft_bwd_make_g": P * B * 3 * H * 4,                          # dout, out -> g
        "ft_bwd_lds": ft_bwd, "ft_bwd_merge": 2 * F * H * 4,
        "ft_index_kernel": P * B * N * 4 + P * B * N * 2,
        "dense_fwd_kernel": dense_pass, "dense_fwd_small": dense_pass, "dense_bwd_small": 2 * dense_pass, "dense_bwd_dx_kernel": dense_pass, "dense_bwd_dw_kernel": dense_pass,
        "adamw_kernel": 28 * (F * H + H + K * O + O),
        "sigmoid_mse_kernel": 5 * B * 4,
    }
    return table.get(kernel)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16384, help="positions per GPU per step")
    ap.add_argument("--occupancy", choices=["full", "fixture"], default="full",
                    help="pieces per synthetic position: 32 (worst case) or the batch.bf histogram (mean 16.2)")
    ap.add_argument("--records", default=None, metavar="FILE.bf",
                    help="take positions from a bulletformat file (tiled to the batch size) instead of the synthetic "
                         "generator, e.g. tests/golden/batch.bf: real feature skew (not the contract workload)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch positions per GPU (default); strong: --batch is the global batch, split over the GPUs")
    ap.add_argument("--net", choices=sorted(NETS), default="cfg2",
                    help="cfg2 = the contract workload (BASELINE configs[1]); cfg3 / cfg4 = the HalfKA multi-layer nets")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--live-loader", type=int, default=0, metavar="BATCHES",
                    help="also time bt_train_run (file reader + threaded mapper + pinned H2D + step) over this many "
                         "batches and report it as `live_loader` (not part of `value`)")
    ap.add_argument("--loader-threads", type=int, default=16)
    ap.add_argument("--host-mapper", action="store_true",
                    help="--live-loader with the threaded host mapper instead of the device mapper")
    args = ap.parse_args()

    import torch
    from bullet_amd import lib, parallel

    rank, world, local_rank = parallel.env_world()
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    dist = None
    if world > 1:
        import torch.distributed as dist
        # nccl == RCCL on ROCm. BULLET_AMD_DIST_BACKEND=gloo rehearses the N > 1 path with several ranks on ONE GPU
        # (RCCL wants a GPU per rank): same code path, the collective staged through the host.
        backend = os.environ.get("BULLET_AMD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    L = lib.load()
    if args.scaling == "strong":  # the reference's fixed global batch, contiguous shards (parallel.shard_range)
        lo, hi = parallel.shard_range(args.batch, rank, world)
        B, global_batch = hi - lo, args.batch
    else:
        B = args.batch
        global_batch = B * world
    n_distinct = 4  # distinct resident batches, rotated
    netcfg, input_kind, layout, workload = NETS[args.net]
    idesc = lib.make_input_desc(input_kind, False, layout)
    batches, nbar = [], 0.0
    for i in range(n_distinct):
        if args.records:
            allrec = np.fromfile(args.records, dtype=np.uint8).reshape(-1, 32)
            rec = allrec[(np.arange(B) + i * B) % allrec.shape[0]].reshape(-1).copy()
        else:
            rec = synth_records(B, 0xB011E7000 + 2 + 1000 * (rank * n_distinct + i), args.occupancy)
        stm, ntm, bk, tg, w = lib.map_batch(idesc, rec, out_buckets=8, blend=0.75, eval_scale=400.0, threads=8)
        cnt = np.where((stm < 0).any(1), (stm < 0).argmax(1), 32)
        nbar += cnt.mean() / n_distinct
        batches.append((stm, ntm, bk, tg, w))

    desc = lib.make_net_desc(**netcfg)
    net = lib.Net(desc, B)
    # random-init weights of the architecture: Normal(0, sqrt(2/in)) from a fixed stream, same on all ranks
    if args.net == "cfg2":
        rng = np.random.default_rng(1234)
        init = []
        for t, n in enumerate(net.sizes):
            fan_in = 768 if t == 0 else 2048
            init.append(np.zeros(n, np.float32) if t & 1 else (rng.standard_normal(n) * np.sqrt(2.0 / fan_in)).astype(np.float32))
        net.set_params(init)
    else:
        net.init_weights(ft_effective_inputs=32)  # the reference's stream and init_with_effective_input_size(32)
    dev_batches = []
    for stm, ntm, bk, tg, w in batches:
        t = [torch.from_numpy(a).cuda() for a in (stm, ntm, bk, tg, w)]
        dev_batches.append(net.make_batch(*t))
    adj = torch.tensor([parallel.adj_factor(global_batch)], device="cuda")
    lr = torch.tensor([1e-3], device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    multi_layer = args.net != "cfg2"

    def step(i):
        if multi_layer:  # the next batch is resident: its index structures are built during this step
            net.hint_next(dev_batches[(i + 1) % n_distinct])
        net.forward_backward(dev_batches[i % n_distinct], stream)
        parallel.allreduce_step_buffers(dist, net.step_buffer)  # grads + loss, one collective; replicas stay identical
        net.optimiser_step(adj, lr, stream)

    # Per-kernel HIP-event timing costs ~20 % of a 0.4 ms step when every kernel is timed, so the full per-kernel
    # table is taken over the warm-up steps and the timed region times only the dominant kernel -- and only every
    # SAMPLE-th launch of it: a timed launch (timestamps on its dispatch packet) costs the stream ~10 us, 4 % of a
    # 0.26 ms step. The average over the samples agrees with the rocprofv3 kernel trace (profiles/README.md).
    profile = not args.no_profile
    if profile and args.warmup > 0:
        lib.profile_start(args.warmup * 16 + 64)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    warm = lib.profile_stop() if (profile and args.warmup > 0) else {}
    # side-stream kernels overlap the main stream and never gate the step
    # (and the brackets of the small kernels that follow a stream join absorb the wait for the other stream: the
    # dominant kernel is looked for among the feature-transformer passes, which are >= 85 % of the critical path)
    cand = {k: v for k, v in warm.items() if k.startswith("ft_") and k != "ft_index_kernel"} or \
           {k: v for k, v in warm.items() if k != "ft_index_kernel"}
    dom = max(cand, key=lambda k: cand[k][1]) if cand else None
    if dist is not None:
        dist.barrier()
    SAMPLE = 8
    if profile and dom:
        lib.profile_start(args.steps * 2 + 64, only=[dom], every=SAMPLE)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern = lib.profile_stop() if (profile and dom) else {}
    loss_per_pos = net.loss.item() / global_batch
    if dist is not None:
        tmax = torch.tensor([elapsed], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = global_batch * args.steps / elapsed
        out = {
            "metric": "training positions/sec (whole node) HalfKA-1024 SCReLU net at 1/2/4/8 GPUs",
            "value": value, "unit": "positions/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if not args.records else "file:" + os.path.basename(args.records),
            "config": {"workload": workload, "batch_per_gpu": B, "global_batch": global_batch,
                       "occupancy": args.occupancy, "mean_active_features": round(float(nbar), 2),
                       "parallelism": f"dp{world}"},
            "final_loss": loss_per_pos,
        }
        if warm:
            out["kernels_us_per_step_warmup"] = {k: round(1e3 * ms / args.warmup, 2) for k, (c, ms) in warm.items()}
        if kern:
            c, ms = kern[dom]
            abytes = algorithmic_bytes(dom, B, nbar, H=netcfg["ft_out"], F=netcfg["num_inputs"],
                                       K=netcfg["ft_out"] * (1 if netcfg.get("pairwise") else 2),
                                       O=netcfg["dense"][0][0] * netcfg["num_buckets"])
            launches_per_step = max(1.0, round(c * SAMPLE / args.steps))
            avg_s = ms / c * 1e-3
            traffic = None
            # PMC passes, see profiles/README.md (cfg2: profiles/hbm_traffic.json; the HalfKA nets: r01_cfg3/)
            tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json" if args.net == "cfg2"
                                 else f"r01_cfg3/hbm_traffic_{args.net}.json")
            if os.path.exists(tpath):
                tab = json.load(open(tpath))
                traffic = (tab.get(dom + "_any") or tab.get(dom) or {}).get("hbm_bytes_per_launch")
            if abytes:
                per_launch = abytes / max(launches_per_step, 1.0)
                ach = per_launch / avg_s / 1e9
                out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                                   "avg_launch_us": avg_s * 1e6, "algorithmic_bytes_per_launch": per_launch,
                                   "timed_launches": c}
                if dom in ("ft_fwd_lds", "ft_bwd_lds"):
                    # The table tile lives in LDS, so the algorithmic bytes are LDS gathers, not HBM reads (frac > 1
                    # against HBM is expected). The bound that applies is the LDS array: ~150 TB/s chip-wide
                    # for ds_read_b128 (MI355X_MICROARCH.md, LDS section), 2/3 of that with the unavoidable
                    # two-random-rows-per-256-B-line conflicts.
                    out["roofline"]["lds"] = {"achieved": ach, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                                              "frac": ach / LDS_PEAK_GBS}
        if args.live_loader > 0 and world == 1:
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "synthetic.bf")
                synth_records(B * 8, 0xB011E7000 + 99, args.occupancy).tofile(path)
                nb = args.live_loader
                assert args.net == "cfg2", "--live-loader is wired for the contract workload"
                rep = net.train(idesc, [path], B, nb, 2, lib.LrSched(lib.LR["constant"], 1e-3, 0, 0, 0),
                                lib.WdlSched(lib.WDL["constant"], 0.75, 0, 0), out_buckets=8, threads=args.loader_threads,
                                device_mapper=not args.host_mapper)
                out["live_loader"] = {"positions_per_sec": rep[-1][2], "batches": nb, "threads": args.loader_threads,
                                      "mapper": "host" if args.host_mapper else "device",
                                      "note": "second superbatch of bt_train_run: reader + mapper + H2D + step"}
        if not args.no_cpu_baseline and world == 1 and args.net == "cfg2":
            out["cpu_baseline"] = cpu_baseline(batches[0], B)
        print(json.dumps(out))
    net.close()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(batch, B):
    """The CPU oracle (oracle/, a scalar C restatement of the reference's CPU semantics: the reference has no CPU
    backend of its own) on a bounded sample of the same workload: full training steps, batch-parallel
    over the host cores with the worker threads INSIDE the C library (orc_train_steps_mt: each thread runs forward +
    backward on a contiguous shard, the shard gradients are added in order, then AdamW). Timed on the optimised build
    (oracle/liboracle_fast.so, -O3 -march=x86-64-v3); the parity build (-O2 -ffp-contract=off,
    the one the tests check against) is timed beside it and reported in `sample`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    do = ob.make_net(**CFG2)
    sample = min(B, 16384)
    stm, ntm, bk, tg, w = (np.ascontiguousarray(a[:sample]) for a in batch)
    cores = max(1, min(16, os.cpu_count() or 1))  # the GPU box gives one job 16 cores (MAX_JOBS there)

    def run(so, threads, budget_s, max_steps):
        L = C.CDLL(os.path.join(ROOT, "oracle", so))
        L.orc_train_steps_mt.restype = C.c_float
        params = ob.net_init(do)
        m, v = [np.zeros_like(p) for p in params], [np.zeros_like(p) for p in params]
        args = (C.byref(do), sample, ob.ptr_array(params), ob.ptr_array(m), ob.ptr_array(v), ob.ip(stm), ob.ip(ntm),
                ob.ip(bk), ob.fp(tg), ob.fp(w), threads)
        t0 = time.perf_counter()
        L.orc_train_steps_mt(*args, 1, C.c_float(1e-3))  # one step sizes the rest of the budget
        per = time.perf_counter() - t0
        more = int(max(0, min(max_steps - 1, (budget_s - per) / per)))
        if more:
            L.orc_train_steps_mt(*args, more, C.c_float(1e-3))
        return 1 + more, time.perf_counter() - t0

    s1, e1 = run("liboracle_fast.so", 1, 3.0, 8)
    sn, en = run("liboracle_fast.so", cores, 10.0, 256)
    sp, ep = run("liboracle.so", cores, 4.0, 64)
    return {"value": sample * sn / en, "unit": "positions/sec", "cores": cores, "kind": "port",
            "single_core_value": sample * s1 / e1, "parity_build_value": sample * sp / ep,
            "sample": f"{sn} full training steps (fwd+bwd+AdamW) of {sample} positions on {cores} threads inside "
                      f"oracle/liboracle_fast.so (gcc -O3 -march=x86-64-v3, pthreads) in {en:.1f} s; + {s1} single-thread "
                      f"steps in {e1:.1f} s; + {sp} steps of the parity build (-O2 -ffp-contract=off) on {cores} threads "
                      f"in {ep:.1f} s"}


if __name__ == "__main__":
    main()

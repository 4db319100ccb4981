#!/usr/bin/env python3
"""bench.py -- training positions/sec of the NNUE hot path on MI355X.

Workload = BASELINE.json configs[1]: Chess768 -> 1024x2 SCReLU -> 1 with 8 output buckets,
batch 16384 per GPU, AdamW; one "step" = fwd + bwd (+ gradient all-reduce when N > 1) + AdamW
over one batch of synthetic positions that is already resident in HBM, with the pinned-host -> HBM upload of the NEXT
batch running beside it on a copy stream (inside the timed loop).
N > 1: one process per GPU (torchrun), data parallel through the library's own RCCL communicator (bt_comm).
`value` is the contract metric of SURVEY.md 8d -- the GLOBAL batch of 16384 split over the GPUs (strong scaling) --
and a `weak` sub-record (16384 per GPU) is printed beside it, each with ms/step and the time of the exchange alone.
At N = 1 the line also carries `fixture_occupancy` (the same net at the fixture's piece counts) and `halfka` (--net cfg3).

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
        "ft_bwd_oct": ft_bwd, "ft_bwd_merge": 2 * F * H * 4,
        "ft_index_kernel": P * B * N * 4 + P * B * N * 2, "ft_index_oct_kernel": P * B * N * 4 + P * B * N * 2,
        "dense_fwd_kernel": dense_pass, "dense_fwd_small": dense_pass, "dense_bwd_small": 2 * dense_pass, "dense_bwd_dx_kernel": dense_pass, "dense_bwd_dw_kernel": dense_pass,
        "adamw_kernel": 28 * (F * H + H + K * O + O),
        "sigmoid_mse_kernel": 5 * B * 4,
    }
    return table.get(kernel.split("<")[0])  # (the live table names kernels with their template arguments)


def roofline_record(dom, c, ms, sample, steps, B, nbar, netcfg, step_s, traffic_tab):
    """The dominant kernel against the bound that applies to it. ALGORITHMIC bytes per launch over the
    kernel's average launch time, measured live with HIP events on the kernel's own stream. The feature-transformer
    kernels keep their table tile in LDS, so those bytes are LDS gathers and the bound is the LDS array (~150 TB/s
    chip-wide, MI355X_MICROARCH.md); `hbm_frac` is the counter-measured HBM traffic of the same kernel against
    8 TB/s, and `algorithmic_vs_hbm` the 8d figure against 8 TB/s -- a ratio, not a fraction: it exceeds 1 when the
    table never leaves the chip."""
    abytes = algorithmic_bytes(dom, B, nbar, H=netcfg["ft_out"], F=netcfg["num_inputs"],
                               K=netcfg["ft_out"] * (1 if netcfg.get("pairwise") else 2),
                               O=netcfg["dense"][0][0] * netcfg["num_buckets"])
    if not abytes or c == 0:
        return None
    launches_per_step = max(1.0, round(c * sample / steps))
    avg_s = ms / c * 1e-3
    per_launch = abytes / launches_per_step
    ach = per_launch / avg_s / 1e9
    base = dom.split("<")[0]
    entry = traffic_tab.get(base) if isinstance(traffic_tab.get(base), dict) else None
    kernel_name = dom  # the kernel's own name, as rocprofv3 prints it (profiles/*kernel_stats.csv)
    measured_batch = (traffic_tab.get("_step") or {}).get("batch", 16384)  # the counters were collected at this batch
    if B != measured_batch:
        entry, traffic_tab = None, {}
    traffic = entry.get("hbm_bytes_per_launch") if entry else None
    lds = base in ("ft_fwd_lds", "ft_bwd_oct")
    peak = LDS_PEAK_GBS if lds else HBM_PEAK_GBS
    rec = {"kernel": kernel_name, "bound": "lds" if lds else "hbm", "achieved": ach, "peak": peak,
           "unit": "GB/s", "frac": ach / peak, "traffic": traffic,
           "hbm_frac": (traffic / avg_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
           "algorithmic_vs_hbm": ach / HBM_PEAK_GBS,
           "avg_launch_us": avg_s * 1e6, "algorithmic_bytes_per_launch": per_launch, "timed_launches": c}
    step = traffic_tab.get("_step") or {}
    if step.get("hbm_bytes_per_step"):
        rec["hbm_bytes_per_step"] = step["hbm_bytes_per_step"]
        rec["hbm_step_frac"] = step["hbm_bytes_per_step"] / step_s / 1e9 / HBM_PEAK_GBS
    return rec


def traffic_table(net_name):
    """HBM bytes per launch from the rocprofv3 PMC passes (profiles/README.md says how they are collected)"""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json" if net_name == "cfg2" else f"hbm_traffic_{net_name}.json")
    return json.load(open(path)) if os.path.exists(path) else {}


def run_case(args, net_name, occupancy, B, global_batch, rank, comm, steps, warmup, records=None):
    """One workload on this rank: warm-up (with the per-kernel table), then `steps` timed steps. A step = upload of the
    NEXT batch from pinned host memory on a copy stream (one packed buffer per batch, overlapped) + fwd + bwd +
    gradient exchange (N > 1) + optimiser over the batch already in HBM."""
    import torch
    from bullet_amd import lib, parallel
    L = lib.load()
    n_distinct = 4  # distinct batches, rotated
    netcfg, input_kind, layout, workload = NETS[net_name]
    idesc = lib.make_input_desc(input_kind, False, layout)
    nnz = netcfg["nnz"]
    words = B * (2 * nnz + 3)  # stm, nstm, buckets, targets, weights: 4-byte words of one batch
    host_packs, nbar, first = [], 0.0, None
    for i in range(n_distinct):
        if records:
            allrec = np.fromfile(records, dtype=np.uint8).reshape(-1, 32)
            rec = allrec[(np.arange(B) + i * B) % allrec.shape[0]].reshape(-1).copy()
        else:
            rec = synth_records(B, 0xB011E7000 + 2 + 1000 * (rank * n_distinct + i), occupancy)
        mapped = lib.map_batch(idesc, rec, out_buckets=8, blend=0.75, eval_scale=400.0, threads=8)
        stm = mapped[0]
        nbar += np.where((stm < 0).any(1), (stm < 0).argmax(1), nnz).mean() / n_distinct
        first = first or mapped
        # pinned staging straight from the runtime (bt_host_alloc_pinned), filled once: stm | nstm | buckets | targets | weights
        hp = C.c_void_p()
        if args.upload == "records":  # the 32-byte records themselves; bt_map_batch_device maps them behind the copy
            lib.check(L.bt_host_alloc_pinned(C.byref(hp), B * 32))
            np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=(B * 32,))[:] = rec
        else:
            lib.check(L.bt_host_alloc_pinned(C.byref(hp), words * 4))
            np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_int32)), shape=(words,))[:] = \
                np.concatenate([a.reshape(-1).view(np.int32) for a in mapped])
        host_packs.append(hp)

    net = lib.Net(lib.make_net_desc(**netcfg), B)
    graphs = args.graphs == "on"
    lib.check(L.bt_net_use_graphs(net.h, int(graphs)))
    if net_name == "cfg2":  # random-init weights of the architecture: Normal(0, sqrt(2/in)), same on all ranks
        rng = np.random.default_rng(1234)
        init = []
        for t, n in enumerate(net.sizes):
            fan_in = 768 if t == 0 else 2048
            init.append(np.zeros(n, np.float32) if t & 1 else (rng.standard_normal(n) * np.sqrt(2.0 / fan_in)).astype(np.float32))
        net.set_params(init)
    else:
        net.init_weights(ft_effective_inputs=32)  # the reference's stream and init_with_effective_input_size(32)

    def views(pack):
        o, out = 0, []
        for n, dt in ((B * nnz, torch.int32), (B * nnz, torch.int32), (B, torch.int32), (B, torch.float32), (B, torch.float32)):
            out.append(pack[o:o + n].view(dt).view(B, nnz) if n == B * nnz else pack[o:o + n].view(dt))
            o += n
        return out

    SETS = 3  # device batch sets: one in use, one being filled, one spare (see step())
    dev_packs = [torch.empty(words, dtype=torch.int32, device="cuda") for _ in range(SETS)]
    dev_batches = [net.make_batch(*views(p)) for p in dev_packs]
    dev_ptrs = [p.data_ptr() for p in dev_packs]
    dev_views = [[t.data_ptr() for t in views(p)] for p in dev_packs]
    dev_records = [torch.empty(B * 32, dtype=torch.uint8, device="cuda") for _ in range(SETS)]
    mopts = lib.MapOpts(8, 0.75, 400.0, 0, 0, 1)

    def handle(create):
        h = C.c_void_p()
        lib.check(create(C.byref(h)))
        return h

    # copy stream + events through the C ABI (a torch copy_ from pinned memory costs ~0.1 ms of HOST time per call on
    # ROCm -- the pointer-attribute query -- which would make the loop host-bound)
    uploaded = [handle(L.bt_event_create) for _ in range(SETS)]
    consumed = [handle(L.bt_event_create) for _ in range(SETS)]
    copy = handle(L.bt_stream_create)
    X = os.environ.get("BENCH_X", "").split(",")  # "devwait": the experiment described in step()
    # the step's own (non-blocking) compute stream: nothing of it rides on the legacy NULL stream, whose implicit
    # joins with other streams are no business of an RCCL collective
    stream_h = handle(L.bt_stream_create)
    stream = stream_h.value
    adj = torch.tensor([parallel.adj_factor(global_batch)], device="cuda")
    lr = torch.tensor([1e-3], device="cuda")
    torch.cuda.synchronize()  # parameters, adj, lr: written on torch's stream
    hints = not args.no_hints
    nbytes = words * 4

    def upload(k, dst, on):
        """batch k of the rotation into device set `dst`, in `on` stream order"""
        if args.upload == "records":
            lib.check(L.bt_memcpy_h2d_async(dev_records[dst].data_ptr(), host_packs[k], B * 32, on))
            lib.check(L.bt_map_batch_device(on, C.byref(idesc), C.byref(mopts), dev_records[dst].data_ptr(), B, *dev_views[dst]))
        else:
            lib.check(L.bt_memcpy_h2d_async(dev_ptrs[dst], host_packs[k], nbytes, on))

    if args.upload == "none":  # (an experiment, not the contract: the sets are resident, nothing moves)
        for k in range(SETS):
            lib.check(L.bt_memcpy_h2d_async(dev_ptrs[k], host_packs[k], nbytes, stream))
    else:
        upload(0, 0, stream)
    lib.check(L.bt_event_record(uploaded[0], stream))
    counter = [0]
    waited = [0.0]
    parts = [0.0, 0.0, 0.0]  # (hosttime aid: upload + map | hint + query | train step)
    landed = C.c_int(0)

    def step():
        i = counter[0]
        counter[0] += 1
        cur, nxt = i % SETS, (i + 1) % SETS
        if args.upload != "none":
            # The set being filled was last read two steps ago. The HOST waits for that step (it then runs at most two
            # steps ahead of the GPU, which always has a full step queued); a device-side wait of the copy stream on the
            # compute stream instead costs ~0.12 ms per step here: the copy then starts late enough to stall the
            # step after it (tools/step_timeline_copies.sh).
            if i >= 2:
                if "devwait" in X:
                    L.bt_stream_wait_event(copy, consumed[nxt])
                elif "hosttime" in X:  # (developer aid: how long the host waits for the GPU, i.e. its slack per step)
                    t_w = time.perf_counter()
                    L.bt_event_sync(consumed[nxt])
                    waited[0] += time.perf_counter() - t_w
                else:
                    L.bt_event_sync(consumed[nxt])
            t_a = time.perf_counter()
            upload((i + 1) % n_distinct, nxt, copy)
            L.bt_event_record(uploaded[nxt], copy)
            parts[0] += time.perf_counter() - t_a
        t_b = time.perf_counter()
        if hints:  # multi-layer nets: its index structures are built during this step, once the upload has landed;
            # the single-layer net: the next step's index build is ordered behind this step without a fork packet
            net.hint_next(dev_batches[nxt], uploaded[nxt] if args.upload != "none" else None)
        if args.upload != "none":
            L.bt_event_query(uploaded[cur], C.byref(landed))
            if not landed.value:  # (the usual case: the upload was enqueued a whole step ago -- no wait packet then)
                L.bt_stream_wait_event(stream, uploaded[cur])
        if comm is not None:
            net.forward_backward(dev_batches[cur], stream)
            net.dp_step(comm, adj, lr, stream)  # exchange (grads + loss slot) and optimiser, replicas stay identical
            L.bt_event_record(consumed[cur], stream)
        else:  # fwd + bwd + AdamW; `consumed` rides on the last kernel's packet
            t_c = time.perf_counter()
            parts[1] += t_c - t_b
            net.train_step(dev_batches[cur], adj, lr, stream, consumed[cur])
            parts[2] += time.perf_counter() - t_c

    def release():
        torch.cuda.synchronize()
        lib.check(L.bt_stream_sync(copy))
        for e in uploaded + consumed:
            L.bt_event_destroy(e)
        L.bt_stream_destroy(copy)
        L.bt_stream_destroy(stream_h)
        for hp in host_packs:
            L.bt_host_free_pinned(hp)
        net.close()

    # Per-kernel HIP-event timing costs ~20 % of a 0.25 ms step when every kernel is timed, so the full per-kernel
    # table is taken over the warm-up steps and the timed region times only the dominant kernel -- and only every
    # SAMPLE-th launch of it. The average over the samples agrees with the rocprofv3 kernel trace (profiles/README.md).
    # Clocks: the GPU needs tens of milliseconds of load to leave its idle power state, and a driver may ask for as
    # few as W = 5 warm-up steps (1 ms). Forward-only passes (bt_net_forward: no parameter, gradient or optimiser
    # state changes hands) keep it busy for --spinup-ms first; the W warm-up steps and the K timed steps follow.
    if args.spinup_ms > 0:
        upl = uploaded[0]
        L.bt_stream_wait_event(stream, upl)
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
            for _ in range(16):
                lib.check(L.bt_net_forward(net.h, stream, C.byref(dev_batches[0]), None, None))
            lib.check(L.bt_stream_sync(stream))
    profile = not args.no_profile
    if profile and warmup > 0:
        lib.profile_start(warmup * 24 + 64)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    warm = lib.profile_stop() if (profile and warmup > 0) else {}
    # side-stream kernels overlap the main stream and never gate the step: the dominant kernel is looked for among
    # the feature-transformer passes (>= 85 % of the critical path)
    side = ("ft_index", "ft_group", "map_batch", "bucket_")  # (index builds and the mapper run beside the step)
    cand = {k: v for k, v in warm.items() if k.startswith("ft_") and not k.startswith(side)} or \
           {k: v for k, v in warm.items() if not k.startswith(side)}
    dom = max(cand, key=lambda k: cand[k][1]) if cand else None
    SAMPLE = 8 if steps >= 64 else max(1, -(-steps // 3))  # (a short run gets three timed launches: each costs ~10 us)
    if comm is not None:
        comm.barrier(stream)
    # Graph steps carry no per-kernel timestamps (and an armed profiler makes the library run steps the ordinary way):
    # with graphs on, the timed region runs unprofiled and the dominant kernel is timed over as many ordinary steps
    # right after it, outside the timed region.
    if profile and dom and not graphs:
        lib.profile_start(steps * 2 + 64, only=[dom], every=SAMPLE)
    torch.cuda.synchronize()
    waited[0] = 0.0
    parts[:] = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    if "hosttime" in X:
        print(f"host: {(t_host - waited[0]) / steps * 1e6:.1f} us busy, {waited[0] / steps * 1e6:.1f} us waiting per step; "
              f"upload+map {parts[0] / steps * 1e6:.1f}, hint+query {parts[1] / steps * 1e6:.1f}, "
              f"train step {parts[2] / steps * 1e6:.1f}", file=sys.stderr)
    if comm is not None:
        comm.barrier(stream)
    elapsed = time.perf_counter() - t0
    kern = lib.profile_stop() if (profile and dom and not graphs) else {}
    n_graph = L.bt_net_graph_launches(net.h)
    if profile and dom and graphs:
        lib.profile_start(steps * 2 + 64, only=[dom], every=SAMPLE)
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        kern = lib.profile_stop()
    res = {"elapsed": elapsed, "loss": net.loss.item() / global_batch, "nbar": float(nbar), "warm": warm, "dom": dom,
           "kern": kern.get(dom) if dom else None, "sample": SAMPLE, "netcfg": netcfg, "workload": workload,
           "first_batch": first, "net": net, "idesc": idesc, "release": release, "graph_steps": n_graph}
    if comm is not None and comm.world > 1:  # the exchange alone, same buffers, back to back
        reps = 50
        torch.cuda.synchronize()
        comm.barrier(stream)
        t1 = time.perf_counter()
        for _ in range(reps):
            comm.all_reduce(net.step_buffer, stream)
        torch.cuda.synchronize()
        res["allreduce_us"] = (time.perf_counter() - t1) / reps * 1e6
        res["allreduce_bytes"] = net.step_buffer.numel() * 4
    return res


def case_record(res, B, global_batch, steps, warmup, net_name, with_kernels=True):
    """the part of the JSON line one run_case contributes"""
    ms = res["elapsed"] / steps * 1e3
    out = {"value": global_batch * steps / res["elapsed"], "unit": "positions/sec", "ms_per_step": ms,
           "batch_per_gpu": B, "global_batch": global_batch, "mean_active_features": round(res["nbar"], 2),
           "final_loss": res["loss"]}
    if "allreduce_us" in res:
        out["allreduce_us"], out["allreduce_bytes"] = res["allreduce_us"], res["allreduce_bytes"]
    if with_kernels and res["warm"]:
        out["kernels_us_per_step_warmup"] = {k: round(1e3 * t / warmup, 2) for k, (c, t) in res["warm"].items()}
    if res["kern"]:
        c, t = res["kern"]
        rl = roofline_record(res["dom"], c, t, res["sample"], steps, B, res["nbar"], res["netcfg"],
                             res["elapsed"] / steps, traffic_table(net_name))
        if rl:
            if res.get("graph_steps"):
                rl["timed"] = "over as many ordinary steps right after the timed region (steps replayed as hipGraphs carry no per-kernel timestamps)"
            out["roofline"] = rl
    if res.get("graph_steps"):
        out["graph_steps"] = res["graph_steps"]  # steps of warm-up + timed region that ran as one hipGraph launch
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16384, help="the GLOBAL batch (every example of the reference: 16384)")
    ap.add_argument("--occupancy", choices=["full", "fixture"], default="full",
                    help="pieces per synthetic position: 32 (worst case) or the batch.bf histogram (mean 16.2)")
    ap.add_argument("--records", default=None, metavar="FILE.bf",
                    help="take positions from a bulletformat file (tiled to the batch size) instead of the synthetic "
                         "generator, e.g. tests/golden/batch.bf: real feature skew (not the contract workload)")
    ap.add_argument("--net", choices=sorted(NETS), default="cfg2",
                    help="cfg2 = the contract workload (BASELINE configs[1]); cfg3 / cfg4 = the HalfKA multi-layer nets")
    ap.add_argument("--upload", choices=["records", "mapped", "none"], default="records",
                    help="what crosses PCIe for the next batch inside the timed loop: the 32-byte records (mapped on the "
                         "device by bt_map_batch_device behind the copy; 32 B/position, the training loop's device_mapper "
                         "path), the five mapped tensors (268 B/position), or nothing (an experiment, not the contract)")
    ap.add_argument("--spinup-ms", type=float, default=150.0,
                    help="milliseconds of forward-only passes before the warm-up steps (GPU clock ramp; no training state "
                         "is touched)")
    ap.add_argument("--no-hints", action="store_true",
                    help="do not announce the next batch (bt_net_hint_next_batch): multi-layer nets then build its index "
                         "structures on the step's own critical path, the single-layer net forks its side stream "
                         "with an event packet per step")
    ap.add_argument("--graphs", choices=("on", "off"), default="off",
                    help="bt_net_use_graphs: steady-state steps replayed as hipGraphs. Off by default: measured on this "
                         "stack (ROCm 7.2) one hipGraphLaunch of the step's six kernels in two branches costs the host "
                         "~100 us against ~39 us for the ten ordinary calls (profiles/r03/bench_lines.txt), so it loses "
                         "at every batch size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-subrecords", action="store_true",
                    help="only the main workload: no fixture_occupancy / halfka (N = 1) or weak (N > 1) sub-records")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--live-loader", type=int, default=0, metavar="BATCHES",
                    help="also time bt_train_run (file reader + mapper + pinned H2D + step) over this many batches and "
                         "report it as `live_loader` (not part of `value`)")
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
    dist = comm = None
    comm_kind = None
    if world > 1:
        import torch.distributed as dist
        # torch.distributed is the launcher's rendezvous (it carries the RCCL id and the final max-over-ranks); the
        # step's collectives are the library's own RCCL communicator. BULLET_AMD_DIST_BACKEND=gloo rehearses the N > 1
        # path with several ranks on ONE GPU (RCCL wants a GPU per rank): hooks, staged through the host.
        backend = os.environ.get("BULLET_AMD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            try:  # (Comm.rccl has every rank agree that librccl loads before any of them enters ncclCommInitRank; an init
                # failure after that agreement cannot be recovered from collectively and ends the job)
                comm = parallel.Comm.rccl(dist)
                comm_kind = "native RCCL communicator (bt_comm_init_rank)"
            except RuntimeError as e:  # the same collectives through torch.distributed's RCCL, as hooks: slower host path
                print(f"rank {rank}: native communicator unavailable ({e}); falling back to torch.distributed hooks",
                      file=sys.stderr)
                comm = parallel.Comm.from_torch(dist)
                comm_kind = "torch.distributed (RCCL) through bt_comm_from_hooks"
            # every rank must have taken the same path
            flag = torch.tensor([0 if comm_kind.startswith("native") else 1], device="cuda")
            dist.all_reduce(flag)
            if 0 < flag.item() < world:
                comm.close()
                comm = parallel.Comm.from_torch(dist)
                comm_kind = "torch.distributed (RCCL) through bt_comm_from_hooks"
        else:
            dist.init_process_group(backend)
            comm = parallel.Comm.from_torch(dist)
            comm_kind = f"torch.distributed ({backend}) through bt_comm_from_hooks"
    lib.load()

    def tmax(x):
        if dist is None:
            return x
        t = torch.tensor([x], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    # the contract metric: fixed GLOBAL batch, split into contiguous shards over the GPUs
    lo, hi = parallel.shard_range(args.batch, rank, world)
    B = hi - lo
    main_res = run_case(args, args.net, args.occupancy, B, args.batch, rank, comm, args.steps, args.warmup, args.records)
    main_res["elapsed"] = tmax(main_res["elapsed"])
    out = None
    if rank == 0:
        rec = case_record(main_res, B, args.batch, args.steps, args.warmup, args.net)
        out = {
            "metric": "training positions/sec (whole node) HalfKA-1024 SCReLU net at 1/2/4/8 GPUs",
            "value": rec["value"], "unit": "positions/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak" if world == 1 else "strong",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not args.records else "file:" + os.path.basename(args.records),
            "config": {"workload": main_res["workload"], "batch_per_gpu": B, "global_batch": args.batch,
                       "occupancy": args.occupancy, "mean_active_features": rec["mean_active_features"],
                       "parallelism": f"dp{world}", "collective": comm_kind,
                       "upload": args.upload, "spinup_ms": args.spinup_ms,
                       "step": "pinned-host -> HBM upload of the next batch (copy stream"
                               + (", 32-byte records + bt_map_batch_device" if args.upload == "records" else "")
                               + ") || fwd + bwd"
                               + (" + RCCL gradient exchange" if world > 1 else "") + " + AdamW"},
            "final_loss": rec["final_loss"],
        }
        for k in ("kernels_us_per_step_warmup", "roofline", "allreduce_us", "allreduce_bytes"):
            if k in rec:
                out[k] = rec[k]
    if args.live_loader > 0 and world == 1:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "synthetic.bf")
            synth_records(B * 8, 0xB011E7000 + 99, args.occupancy).tofile(path)
            nb = args.live_loader
            assert args.net == "cfg2", "--live-loader is wired for the contract workload"
            rep = main_res["net"].train(main_res["idesc"], [path], B, nb, 2, lib.LrSched(lib.LR["constant"], 1e-3, 0, 0, 0),
                                        lib.WdlSched(lib.WDL["constant"], 0.75, 0, 0), out_buckets=8,
                                        threads=args.loader_threads, device_mapper=not args.host_mapper)
            out["live_loader"] = {"positions_per_sec": rep[-1][2], "batches": nb, "threads": args.loader_threads,
                                  "mapper": "host" if args.host_mapper else "device",
                                  "note": "second superbatch of bt_train_run: reader + mapper + H2D + step"}
    first_batch = main_res["first_batch"]
    main_res["release"]()
    del main_res

    if not args.no_subrecords and not args.records:
        if world == 1:
            # the same contract workload at the fixture's piece counts, and the HalfKA-1024 net the metric is named for
            for key, net_name, occ in (("fixture_occupancy", args.net, "fixture"), ("halfka", "cfg3", "full")):
                if (net_name, occ) == (args.net, args.occupancy):
                    continue
                r = run_case(args, net_name, occ, B, args.batch, rank, None, args.steps, args.warmup)
                sub = case_record(r, B, args.batch, args.steps, args.warmup, net_name)
                sub["workload"], sub["occupancy"] = r["workload"], occ
                out[key] = sub
                r["release"]()
        else:
            # weak scaling beside it: the full batch on every GPU (global batch = N x 16384)
            r = run_case(args, args.net, args.occupancy, args.batch, args.batch * world, rank, comm, args.steps, args.warmup)
            r["elapsed"] = tmax(r["elapsed"])
            if rank == 0:
                out["weak"] = case_record(r, args.batch, args.batch * world, args.steps, args.warmup, args.net, with_kernels=False)
            r["release"]()
    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and args.net == "cfg2":
            out["cpu_baseline"] = cpu_baseline(first_batch, min(B, 16384))
        print(json.dumps(out))
    if comm is not None:
        comm.close()
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

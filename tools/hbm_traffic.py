#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the two rocprofv3 --pmc passes collected by tools/collect_pmc.sh.

FETCH_SIZE / WRITE_SIZE are in KB. On gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read
(MI355X_MICROARCH.md, HBM / rocprofv3 section), so it is doubled; WRITE_SIZE is exact for 16-B stores."""
import collections
import csv
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_hbm"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/hbm_traffic.json"
KERNELS = ["ft_fwd_lds", "ft_fwd_decode_kernel", "ft_index_kernel", "ft_index_oct_kernel", "ft_bwd_oct", "ft_bwd_merge_head", "map_batch_kernel",
           "ft_bwd_merge", "head_finish_kernel", "dense_fwd_small",
           "dense_bwd_small_light", "dense_bwd_small", "sum_partials_seq4_kernel", "sum_partials4_kernel", "adamw_kernel",
           # the HalfKA multi-layer nets (bench.py --net cfg3 / cfg4)
           "ft_bwd_merge_grp", "ft_group_count", "ft_group_place", "bucket_count_kernel", "bucket_place_kernel", "ft_bwd_merge_step",
           "dense_mfma_fwd", "dense_mfma_dx", "dense_mfma_dw", "dense_mfma_sum", "dense_tail_kernel", "dense_tail_sum",
           "dense_gz_kernel", "factor_expand_kernel", "factor_reduce_kernel"]


def avg(path, counter):
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for k in KERNELS:
            if k + "<" in r["Kernel_Name"] or k + "(" in r["Kernel_Name"]:
                acc[k] += float(r["Counter_Value"])
                n[k] += 1
    return {k: acc[k] / n[k] for k in acc}, n


(f, nf), (w, _) = avg(f"{src}/FETCH_SIZE.csv", "FETCH_SIZE"), avg(f"{src}/WRITE_SIZE.csv", "WRITE_SIZE")
out = {"_comment": __doc__.strip().replace("\n", " ")}
steps = max(nf.get("ft_fwd_lds", 0), 1)  # one forward launch per step
total = 0.0
for k in KERNELS:
    if k in f and k in w:
        per_launch = (2 * f[k] + w[k]) * 1024
        lps = nf[k] / steps
        lps = float(round(lps)) if abs(lps - round(lps)) < 0.2 and round(lps) >= 1 else lps  # (+1 launch: the first upload)
        out[k] = {"kernel": k, "FETCH_SIZE_KB": round(f[k], 1), "WRITE_SIZE_KB": round(w[k], 1),
                  "hbm_bytes_per_launch": round(per_launch, 1), "launches_per_step": round(lps, 3)}
        total += per_launch * lps
# uploads of the next batch are copies, not kernels: B * (2 N + 3) * 4 bytes written per step, not in the counters
out["_step"] = {"hbm_bytes_per_step": round(total, 1), "steps_profiled": steps, "batch": 16384,
                "note": "sum over the kernels above of hbm_bytes_per_launch * launches_per_step (all streams)"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))

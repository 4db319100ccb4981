#!/usr/bin/env python3
"""Per-kernel averages of the SQ counter passes of tools/pmc_sq_bench.sh, with the busy fractions DESIGN.md quotes.
SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); SQ_BUSY_CYCLES
is per shader engine; the per-CU LDS figures are SQ_LDS_IDX_ACTIVE over the kernel's cycles on 256 CUs."""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
want = sys.argv[2:] or ["ft_bwd_oct", "ft_fwd_lds", "ft_index_oct_kernel", "ft_bwd_merge", "head_finish", "dense_mfma", "dense_tail"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
for f in sorted(glob.glob(f"{root}/*.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+)(<[^>]*>)?\(", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
        k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40]
        if not any(w in k for w in want):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    v = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    print(k, f"({max(cnt[k].values())} launches)")
    for c in sorted(v):
        print(f"   {c:30s} {v[c]:16,.0f}")
    wc = v.get("SQ_WAVE_CYCLES")
    if wc:
        def pct(name):
            return f"{100 * v[name] / wc:5.1f} %" if name in v else "   n/a"
        print(f"   -> of the waves' cycles: instruction active {pct('SQ_ACTIVE_INST_ANY')}, waiting to issue {pct('SQ_WAIT_INST_ANY')}, "
              f"parked (s_waitcnt / barrier) {pct('SQ_WAIT_ANY')}, VALU {pct('SQ_ACTIVE_INST_VALU')}, LDS instr {pct('SQ_ACTIVE_INST_LDS')}, "
              f"LDS issue stall {pct('SQ_WAIT_INST_LDS')}")
    if "GRBM_GUI_ACTIVE" in v and "SQ_LDS_IDX_ACTIVE" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        print(f"   -> kernel cycles {cyc:,.0f}; LDS array busy {100 * v['SQ_LDS_IDX_ACTIVE'] / (cyc * 256):5.1f} % of CU-cycles, "
              f"bank conflicts {100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v['SQ_LDS_IDX_ACTIVE'], 1):5.1f} % of the busy cycles"
              + (f", MFMA pipe busy {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):5.1f} % of SIMD-cycles" if "SQ_VALU_MFMA_BUSY_CYCLES" in v else ""))

#!/bin/bash
# MFMA utilisation of the matrix-core kernels of the HalfKA multi-layer step (dense_mfma_*): one --pmc pass with
# SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs) and GRBM_GUI_ACTIVE (kernel cycles, summed over the 8 XCDs).
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_mfma; mkdir -p gpurun_out/pmc_mfma
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- python3 tools/bench_cfg3.py 1024 768 tests/golden/batch.bf > gpurun_out/pmc_mfma/log.txt 2>&1
f=$(find gpurun_out/pmc_mfma -name "*counter_collection.csv" | head -1)
cp "$f" gpurun_out/pmc_mfma/mfma_counter_collection.csv
python3 - "$f" <<'PY'
import collections, csv, re, sys
acc, n = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(?:\)::|^|\s)([A-Za-z_][A-Za-z_0-9]*)(?:<[^(]*>)?\(", r["Kernel_Name"])
    k = m.group(1) if m else r["Kernel_Name"][:30]
    acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted({k for k, _ in acc}):
    busy = acc.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), 0) / max(n[(k, "SQ_VALU_MFMA_BUSY_CYCLES")], 1)
    gui = acc.get((k, "GRBM_GUI_ACTIVE"), 0) / max(n[(k, "GRBM_GUI_ACTIVE")], 1)
    if busy > 0:
        print(f"{k:28s} mfma_busy_cycles={busy:14.0f} kernel_cycles={gui / 8:10.0f} util={busy / (gui / 8 * 1024):.4f}")
PY

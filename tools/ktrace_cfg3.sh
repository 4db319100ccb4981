#!/bin/bash
# true per-kernel durations of the HalfKA multi-layer step (tools/bench_cfg3.py)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf gpurun_out/kt3; mkdir -p gpurun_out/kt3
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt3 -- python3 tools/bench_cfg3.py "$@" > gpurun_out/kt3/log.txt 2>&1
f=$(find gpurun_out/kt3 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 20:
        print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.2f}')
PY

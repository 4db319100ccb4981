#!/bin/bash
# true per-kernel durations of the whole training step (bench.py without its own event profiling)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf gpurun_out/ktb; mkdir -p gpurun_out/ktb
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ktb -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-profile > gpurun_out/ktb/log.txt 2>&1
f=$(find gpurun_out/ktb -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/ktb/kernel_stats.csv
python3 - "$f" $KT_FILTER <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 50 and (len(sys.argv) < 3 or sys.argv[2] in r["Name"]):
        print(f'{r["Name"][:60]:60s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={r["Percentage"]}')
PY
tail -1 gpurun_out/ktb/log.txt | cut -c 1-200

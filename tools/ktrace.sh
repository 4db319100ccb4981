#!/bin/bash
# per-kernel durations of tools/ft_microbench.py (FT forward + index + backward only)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf gpurun_out/kt; mkdir -p gpurun_out/kt
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 tools/ft_microbench.py 5 "$@" > gpurun_out/kt/log.txt 2>&1
f=$(find gpurun_out/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ft_" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.2f} min_us={float(r["MinNs"])/1e3:9.2f}')
PY

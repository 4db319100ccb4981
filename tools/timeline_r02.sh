#!/bin/bash
# one step of bench.py as a kernel timeline (start, duration, gap to the previous end, queue); BENCH_ARGS picks the net
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf gpurun_out/tl2; mkdir -p gpurun_out/tl2
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl2 -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-subrecords $BENCH_ARGS > gpurun_out/tl2/log.txt 2>&1
python3 tools/timeline_cfg3.py "$(find gpurun_out/tl2 -name '*kernel_trace.csv' | head -1)"

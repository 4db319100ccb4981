#!/bin/bash
# Round artifacts: the default bench line and the rocprofv3 kernel-trace summary of the same command.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf gpurun_out/final; mkdir -p gpurun_out/final
timeout -k 10 300 python3 bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/trace -- python3 bench.py > gpurun_out/final/bench_traced.json 2> gpurun_out/final/trace.err
cp "$(find gpurun_out/final/trace -name '*kernel_stats.csv' | head -1)" gpurun_out/final/kernel_stats.csv
rm -rf gpurun_out/final/trace
cut -c 1-400 gpurun_out/final/bench.json
head -12 gpurun_out/final/kernel_stats.csv

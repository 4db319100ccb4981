#!/bin/bash
# HBM-side traffic of every kernel of the training step: FETCH_SIZE and WRITE_SIZE in separate passes
# (rocprofv3 cannot hold both in one), over a short bench.py run. Output: gpurun_out/pmc_hbm/{fetch,write}.csv
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_hbm; mkdir -p gpurun_out/pmc_hbm
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_hbm/$c -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile $BENCH_ARGS > gpurun_out/pmc_hbm/$c.log 2>&1
  cp "$(find gpurun_out/pmc_hbm/$c -name '*counter_collection.csv' | head -1)" gpurun_out/pmc_hbm/$c.csv
done
ls -la gpurun_out/pmc_hbm/*.csv

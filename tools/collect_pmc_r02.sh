#!/bin/bash
# only the two PMC passes of tools/collect_r02.sh (no spin-up: the counters then cover exactly 8 training steps)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
net=${1:-cfg2}
out=gpurun_out/r02_$net
mkdir -p $out
args="--net $net --no-subrecords --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_$c
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py $args --steps 6 --warmup 2 --no-profile --spinup-ms 0 > $out/pmc_$c.log 2>&1
  cp "$(find $out/pmc_$c -name '*counter_collection.csv' | head -1)" $out/$c.csv
  rm -rf $out/pmc_$c
done
ls -la $out/*.csv

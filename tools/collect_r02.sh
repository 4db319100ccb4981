#!/bin/bash
# Round-2 artifacts. For each net: the bench line, the rocprofv3 kernel-trace summary of the same command, and the
# two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) the HBM traffic figures come from.
# usage (GPU box): tools/collect_r02.sh [cfg2|cfg3]
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
net=${1:-cfg2}
out=gpurun_out/r02_$net
rm -rf $out; mkdir -p $out
args="--net $net --no-subrecords --no-cpu-baseline"
timeout -k 10 300 python3 bench.py $args > $out/bench.json 2> $out/bench.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $args > $out/bench_traced.json 2> $out/trace.err
cp "$(find $out/trace -name '*kernel_stats.csv' | head -1)" $out/kernel_stats.csv
rm -rf $out/trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py $args --steps 6 --warmup 2 --no-profile --spinup-ms 0 > $out/pmc_$c.log 2>&1
  cp "$(find $out/pmc_$c -name '*counter_collection.csv' | head -1)" $out/$c.csv
  rm -rf $out/pmc_$c
done
cut -c 1-300 $out/bench.json
head -14 $out/kernel_stats.csv
ls -la $out

#!/bin/bash
# Round-3 artifacts, one GPU-box call per net: tools/collect_r03.sh [cfg2|cfg3]
#   bench line, rocprofv3 kernel-trace summary of the same command, the two HBM-traffic PMC passes (separate runs),
#   the SQ counter passes over bench.py (tools/pmc_sq_bench.sh), and for cfg2 the small-batch lines and timelines.
# Everything lands in gpurun_out/r03_<net>/; tools/file_r03.sh copies what is judged into profiles/r03/.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
net=${1:-cfg2}
out=gpurun_out/r03_$net
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
bash tools/pmc_sq_bench.sh $net $out/sq > $out/sq_log.txt 2>&1
cp $out/sq/summary.txt $out/sq_summary.txt
if [ $net = cfg2 ]; then
  for b in 2048 4096; do
    timeout -k 10 200 python3 bench.py $args --batch $b > $out/bench_b$b.json 2> /dev/null
    BENCH_ARGS="--batch $b" bash tools/timeline_r02.sh > $out/timeline_b$b.txt 2>&1
  done
  timeout -k 10 200 python3 bench.py $args --occupancy fixture > $out/bench_fixture.json 2> /dev/null
else
  BENCH_ARGS="--net $net" bash tools/timeline_r02.sh > $out/timeline.txt 2>&1
fi
python3 tools/show_bench.py $out/bench*.json
head -14 $out/kernel_stats.csv

#!/bin/bash
# copies what tools/collect_r03.sh left under gpurun_out/r03_<net>/ into profiles/r03/ and rebuilds the HBM traffic tables
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/r03
for net in cfg2 cfg3; do
  src=gpurun_out/r03_$net
  [ -d $src ] || continue
  cp $src/bench.json profiles/r03/${net}_bench.json
  cp $src/bench_traced.json profiles/r03/${net}_bench_traced.json
  cp $src/kernel_stats.csv profiles/r03/${net}_kernel_stats.csv
  cp $src/FETCH_SIZE.csv profiles/r03/${net}_FETCH_SIZE_counter_collection.csv
  cp $src/WRITE_SIZE.csv profiles/r03/${net}_WRITE_SIZE_counter_collection.csv
  cp $src/sq_summary.txt profiles/r03/${net}_sq_counters.txt
  for f in bench_b2048.json bench_b4096.json bench_fixture.json timeline_b2048.txt timeline_b4096.txt; do
    [ -f $src/$f ] && cp $src/$f profiles/r03/${net}_$f
  done
  [ -f $src/timeline.txt ] && cp $src/timeline.txt profiles/r03/${net}_timeline.txt
done
python3 tools/hbm_traffic.py gpurun_out/r03_cfg2 profiles/hbm_traffic.json > /dev/null
python3 tools/hbm_traffic.py gpurun_out/r03_cfg3 profiles/hbm_traffic_cfg3.json > /dev/null
ls profiles/r03

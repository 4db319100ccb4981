#!/bin/bash
# the builder-run lines of profiles/r03/bench_lines.txt (round 2: profiles/r02/): other nets, occupancies, loaders and shard sizes
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/bench_lines.txt
: > $out
run() {
  echo "== bench.py --no-subrecords --no-cpu-baseline $*" >> $out
  timeout -k 10 300 python3 bench.py --no-subrecords --no-cpu-baseline "$@" 2> gpurun_out/bench_lines.err | tail -n 1 >> $out || { tail -n 5 gpurun_out/bench_lines.err; exit 1; }
  tail -n 1 $out | cut -c 1-190
}
run --net cfg3 --records tests/golden/batch.bf &&
run --net cfg3 --records tests/golden/batch.bf --no-hints &&
run --net cfg3 --no-hints &&
run --net cfg4 &&
run --no-hints &&
run --live-loader 400 &&
run --live-loader 400 --host-mapper &&
run --upload mapped &&
run --batch 2048 &&
run --batch 4096 &&
run --batch 8192 &&
for b in 2048 4096; do
  echo "== host time per step, BENCH_X=hosttime bench.py --batch $b --no-profile" >> $out
  BENCH_X=hosttime timeout -k 10 200 python3 bench.py --no-subrecords --no-cpu-baseline --no-profile --batch $b 2>&1 >/dev/null | grep "^host:" >> $out
done
tail -n 4 $out | cut -c 1-200

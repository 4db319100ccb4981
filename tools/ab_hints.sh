#!/bin/bash
# developer A/B: the default bench line with and without next-batch hints
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp
for rep in 1 2; do
for h in "" "--no-hints"; do
  for occ in full fixture; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-subrecords --occupancy $occ $h > gpurun_out/exp/h_$occ$h.txt 2>&1 || { tail -n 5 gpurun_out/exp/h_$occ$h.txt; exit 1; }
    tail -n 1 gpurun_out/exp/h_$occ$h.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('hints=$h', '$occ', round(d['value'] / 1e6, 2), 'M pos/s', round(d['ms_per_step'], 4), 'ms')"
  done
done
done

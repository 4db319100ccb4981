#!/bin/bash
# developer A/B of an environment switch: tools/ab_env.sh NAME v1 v2 ... (bench.py, full + fixture occupancy)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp
name=$1; shift
for v in "$@"; do
  for occ in full fixture; do
    env $name=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-subrecords --occupancy $occ > gpurun_out/exp/env_${v}_$occ.txt 2>&1 || { tail -n 5 gpurun_out/exp/env_${v}_$occ.txt; exit 1; }
    tail -n 1 gpurun_out/exp/env_${v}_$occ.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d.get('kernels_us_per_step_warmup', {})
print('$name=$v $occ', round(d['value'] / 1e6, 2), 'M pos/s', round(d['ms_per_step'], 4), 'ms', 'fwd', k.get('ft_fwd_lds'), 'idx', k.get('ft_index_kernel'), 'bwd', k.get('ft_bwd_lds'))"
  done
done

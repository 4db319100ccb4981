#!/bin/bash
# developer A/B: bench.py (no CPU baseline) for each tools/libbullet_amd_<name>.so given, plus the real library
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp
for name in real "$@"; do
  for occ in full fixture; do
    if [ $name = real ]; then unset BT_LIB_PATH; else export BT_LIB_PATH=tools/libbullet_amd_$name.so; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-subrecords --occupancy $occ > gpurun_out/exp/bench_${name}_$occ.txt 2>&1 || { tail -n 5 gpurun_out/exp/bench_${name}_$occ.txt; exit 1; }
    tail -n 1 gpurun_out/exp/bench_${name}_$occ.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d.get('kernels_us_per_step_warmup', {})
print('$name $occ', round(d['value'] / 1e6, 2), 'M pos/s', round(d['ms_per_step'], 4), 'ms', 'bwd', round(d.get('roofline', {}).get('avg_launch_us', 0), 1), 'fwd', next((v for n, v in k.items() if n.startswith('ft_fwd_lds')), None), 'idx', next((v for n, v in k.items() if n.startswith('ft_index')), None))"
  done
done

#!/bin/bash
# developer A/B over the nets: bench.py --net cfg3 (synthetic + real positions) and cfg4 for each library given
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp
for name in real "$@"; do
  if [ $name = real ]; then unset BT_LIB_PATH; else export BT_LIB_PATH=tools/libbullet_amd_$name.so; fi
  for a in "--net cfg3" "--net cfg3 --records tests/golden/batch.bf" "--net cfg4"; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-subrecords $a > gpurun_out/exp/bn.txt 2>&1 || { tail -n 5 gpurun_out/exp/bn.txt; exit 1; }
    tail -n 1 gpurun_out/exp/bn.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d.get('kernels_us_per_step_warmup', {})
print('$name', '$a', round(d['value'] / 1e6, 2), 'M pos/s', round(d['ms_per_step'], 4), 'ms', {n: v for n, v in k.items() if v > 15})"
  done
done

#!/bin/bash
# developer A/B on the multi-layer nets: bench.py --net cfg3 on real positions (with / without hints) and cfg4, for
# the real library and each tools/libbullet_amd_<name>.so given
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp
for name in real "$@"; do
  if [ $name = real ]; then unset BT_LIB_PATH; else export BT_LIB_PATH=tools/libbullet_amd_$name.so; fi
  for cfg in "--net cfg3 --records tests/golden/batch.bf" "--net cfg3 --records tests/golden/batch.bf --no-hints" "--net cfg3" "--net cfg4"; do
    tag=$(echo "$cfg" | tr -d ' /.-')
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-subrecords $cfg > gpurun_out/exp/ab_${name}_$tag.txt 2>&1 || { tail -n 5 gpurun_out/exp/ab_${name}_$tag.txt; exit 1; }
    tail -n 1 gpurun_out/exp/ab_${name}_$tag.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', '$cfg', round(d['value'] / 1e6, 2), 'M pos/s', round(d['ms_per_step'], 4), 'ms')"
  done
done

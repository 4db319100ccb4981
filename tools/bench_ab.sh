#!/bin/bash
# developer A/B: bench.py lines (no CPU baseline) for the given env/arg variants; prints value, ms/step, kernel table
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp
run() {  # name, env assignments..., -- bench args
    name=$1; shift
    envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/exp/bench_$name.txt 2>&1 || { tail -n 5 gpurun_out/exp/bench_$name.txt; return 1; }
    tail -n 1 gpurun_out/exp/bench_$name.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['value'] / 1e6, 2), 'M pos/s', round(d['ms_per_step'], 4), 'ms', d.get('kernels_us_per_step_warmup'), round(d.get('roofline', {}).get('avg_launch_us', 0), 1))"
}
run oct X=1 -- && run old BT_FT_OLD_BWD=1 -- && run oct_fix X=1 -- --occupancy fixture && run old_fix BT_FT_OLD_BWD=1 -- --occupancy fixture

#!/bin/bash
# developer A/B of ONE kernel's duration (rocprofv3 kernel trace of a short cfg3 bench on real positions; wrong
# results are fine: timing-only variants): tools/ab_kernel.sh <kernel substring> <variant> ...
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
k=$1; shift
for name in real "$@"; do
  if [ $name = real ]; then unset BT_LIB_PATH; else export BT_LIB_PATH=tools/libbullet_amd_$name.so; fi
  rm -rf gpurun_out/abk; mkdir -p gpurun_out/abk
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abk -- python3 bench.py ${BENCH_ARGS:---net cfg3 --records tests/golden/batch.bf} --steps 30 --warmup 5 --spinup-ms 20 --no-cpu-baseline --no-profile --no-subrecords > gpurun_out/abk/log.txt 2>&1 || { tail -n 5 gpurun_out/abk/log.txt; exit 1; }
  echo "== $name"; python3 tools/kernel_by_phase.py "$(find gpurun_out/abk -name '*kernel_trace.csv' | head -1)" $k
done

#!/bin/bash
# SQ counter passes over bench.py itself (the training step, not a microbenchmark): tools/pmc_sq_bench.sh [net] [outdir]
# One rocprofv3 run per pass (8 SQ counters at most each); no spin-up, so the counters cover whole training steps only.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
net=${1:-cfg2}
out=${2:-gpurun_out/pmc_sq_$net}
rm -rf $out; mkdir -p $out
args="--net $net --no-subrecords --no-cpu-baseline --steps 6 --warmup 2 --no-profile --spinup-ms 0 $BENCH_ARGS"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py $args > $out/$name.log 2>&1
  cp "$(find $out/$name -name '*counter_collection.csv' | head -1)" $out/$name.csv
  rm -rf $out/$name
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
pass b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INSTS_SMEM
pass c SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE
python3 tools/pmc_sq_table.py $out | tee $out/summary.txt

#!/bin/bash
# SQ counter passes over tools/ft_microbench.py (FT forward/backward only); results under gpurun_out/pmc/
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc/a -- python3 tools/ft_microbench.py 3 > gpurun_out/pmc/a.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc/b -- python3 tools/ft_microbench.py 3 > gpurun_out/pmc/b.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc/c -- python3 tools/ft_microbench.py 3 > gpurun_out/pmc/c.log 2>&1

#!/bin/bash
# item-size sweep of the segment-grouped FT kernels on BASELINE configs[2] (tools/bench_cfg3.py); $1 = optional .bf file
cd "$GRAFT_REPO_ROOT"
for fi in 1024 2048 4096; do for gc in 4 8 16; do
  echo "FWD_ITEM=$fi BWD_CHUNKS=$gc"
  BT_FT_FWD_ITEM=$fi BT_FT_BWD_CHUNKS=$gc timeout -k 10 100 python tools/bench_cfg3.py 1024 768 $1 | grep cfg3 || exit 1
done; done

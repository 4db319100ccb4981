#!/bin/bash
# FT kernel durations for the table variants of tools/ft_microbench.py: (pairwise, king buckets)
cd "$GRAFT_REPO_ROOT"
for v in "0 1" "1 1" "0 16" "1 16"; do
  set -- $v
  echo "== PAIR=$1 NSEG=$2"
  BT_MB_PAIR=$1 BT_MB_NSEG=$2 bash tools/ktrace.sh || exit 1
done

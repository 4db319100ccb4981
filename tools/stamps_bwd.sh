#!/bin/bash
# developer run on the GPU box: phase stamps of ft_bwd_oct (build first: tools/build_variant_of.sh stamps ft.hip -DBT_STAMPS)
cd "$GRAFT_REPO_ROOT"
for occ in full fixture; do
  echo "== $occ"
  BT_LIB_PATH=tools/libbullet_amd_${1:-stamps}.so timeout -k 10 200 python3 tools/stamps_bwd.py --occupancy $occ 2>&1 | tail -n 14 || exit 1
done

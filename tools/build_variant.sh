#!/bin/bash
# developer build: tools/libbullet_amd_<name>.so with ft.hip compiled with extra -D flags (A/B through BT_LIB_PATH)
# usage: tools/build_variant.sh name -DBT_OCT_WAVES=8 ...
set -e
name=$1; shift
cd "$(dirname "$0")/../bullet_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c ft.hip -o /tmp/ft_$name.o
objs=$(ls obj/*.o | grep -v ft.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libbullet_amd_$name.so /tmp/ft_$name.o $objs -lpthread

#!/bin/bash
# developer build: tools/libbullet_amd_<name>.so with ONE source of csrc compiled with extra -D flags
# usage: tools/build_variant_of.sh name dense.hip -DFLAG ...
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../bullet_amd/csrc"
make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c $src -o /tmp/${src}_$name.o
objs=$(ls obj/*.o | grep -v "$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libbullet_amd_$name.so /tmp/${src}_$name.o $objs -lpthread -ldl

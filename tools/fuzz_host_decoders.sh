#!/bin/bash
# Host-only AddressSanitizer + UBSan run of the data decoders (Stockfish binpack, viriformat with and without its built-in
# filter, montyformat, FEN text): valid files written by the tests' encoders are mutated byte-wise 9000 times, plus 20000 random text lines and 40000 batches of
# random / bit-flipped plain bulletformat records through bt_map_batch. CPU only.
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/bt_fuzz
python3 - <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_sfbinpack as ts, test_viribinpack as tv, test_montybinpack as tm
open("/tmp/bt_fuzz/seed.mf", "wb").write(tm.encode_game(tv.GAME1, 2, tv.EV1) + tm.encode_game(tv.GAME2, 0, tv.EV2))
open("/tmp/bt_fuzz/seed.binpack", "wb").write(ts.build_file(True))
open("/tmp/bt_fuzz/seed.vf", "wb").write(tv.encode_game(tv.GAME1, {0, 7, 56, 63}, 2, tv.EV1) + tv.encode_game(tv.GAME2, {56}, 0, tv.EV2))
PY
/opt/rocm/bin/hipcc -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-gpu-sanitize \
    tools/fuzz_host_decoders.cpp bullet_amd/csrc/sfbinpack.cpp bullet_amd/csrc/viribinpack.cpp bullet_amd/csrc/montybinpack.cpp bullet_amd/csrc/host_inputs.cpp \
    -lpthread -o /tmp/bt_fuzz/drv
/tmp/bt_fuzz/drv

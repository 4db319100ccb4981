#!/bin/bash
# developer check of the driver's short form: what do warm-up length, spin-up time and the live kernel timing change?
cd "$GRAFT_REPO_ROOT"
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --no-profile" "--steps 20 --warmup 5" "--steps 20 --warmup 5 --no-profile" "--steps 200 --warmup 20" "--steps 200 --warmup 20 --no-profile"; do
  timeout -k 10 300 python bench.py --gpus 1 --no-cpu-baseline --no-subrecords $args 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$args', round(d['value']/1e6,2), round(d['ms_per_step'],4))"
done

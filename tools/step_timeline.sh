#!/bin/bash
# timeline of one training step (kernel start/end relative to the step's first kernel), from rocprofv3 --kernel-trace
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile $BENCH_ARGS > gpurun_out/tl/log.txt 2>&1
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# steps start at ft_index_kernel / ft_fwd_lds; take the 20th step
starts = [i for i, n in enumerate(names) if "ft_fwd_lds" in n]
i0, i1 = starts[20], starts[21]
# include the index kernel launched just before the forward
j0 = i0 - 1 if "ft_index" in names[i0 - 1] else i0
t0 = int(rows[j0]["Start_Timestamp"])
prev_end = None
for r in rows[j0:i1 + (1 if "ft_index" in names[i1 - 1] else 0) - 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f'{r["Kernel_Name"].split("(")[0][-34:]:34s} start {s/1e3:8.1f} us  end {e/1e3:8.1f} us  dur {(e-s)/1e3:7.1f}')
print("step period:", (int(rows[i1]["Start_Timestamp"]) - int(rows[i0]["Start_Timestamp"])) / 1e3, "us")
PY

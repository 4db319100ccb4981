#!/bin/bash
# timeline of one training step with the upload beside it: kernels (rocprofv3 --kernel-trace) and copies
# (--memory-copy-trace) on one time axis, relative to the forward kernel of the step
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf gpurun_out/tlc; mkdir -p gpurun_out/tlc
timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tlc -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-profile --no-subrecords $BENCH_ARGS > gpurun_out/tlc/log.txt 2>&1
k=$(find gpurun_out/tlc -name "*kernel_trace.csv" | head -1)
m=$(find gpurun_out/tlc -name "*memory_copy_trace.csv" | head -1)
python3 - "$k" "$m" <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("anonymous namespace)::", "")[-40:], "q" + r.get("Queue_Id", "?")))
for r in csv.DictReader(open(sys.argv[2])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") , ""))
ev.sort()
fw = [i for i, e in enumerate(ev) if "ft_fwd_lds" in e[2]]
i0, i1 = fw[20], fw[22]
t0 = ev[i0][0]
for s, e, n, q in ev[i0 - 3:i1 + 1]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {q:4s} {n}")
PY

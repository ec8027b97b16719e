#!/bin/bash
# rocprofv3 kernel trace of the training step + per-stream timeline + the kernels of one step in start order.
#   gpurun -- 'bash tools/trace_step.sh <tag> [bench.py args]'   ->  gpurun_out/<tag>_{timeline,step_kernels}.txt, gpurun_out/<tag>_trace/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -o tr -- python3 bench.py --steps 8 --warmup 3 --no-predict --no-cpu-baseline "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}.err
f=$(find gpurun_out/${tag}_trace -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f > gpurun_out/${tag}_timeline.txt 2>&1
python - "$f" "gpurun_out/${tag}_step_kernels.txt" <<'PY'
import csv, sys
rows = [(r["Kernel_Name"], int(r["Stream_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: r[2])
marks = [r[2] for r in rows if "dice_partial" in r[0]]
k = len(marks) // 2
a, b = marks[k], marks[k + 1]
win = [r for r in rows if a <= r[2] < b]
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
with open(sys.argv[2], "w") as o:
    for r in win:
        o.write(f"{(r[2] - a) / 1e3:9.1f} {(r[3] - r[2]) / 1e3:7.1f} s{r[1]} {short(r[0])}\n")
PY
python tools/wg_time_step.py "$f" > gpurun_out/${tag}_wg_time.txt 2>&1

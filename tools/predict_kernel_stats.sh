#!/bin/bash
# rocprofv3 kernel stats of one 512^3 12-direction prediction -> gpurun_out/<tag>_predict_kernel_stats.csv + a top-20 table on stdout
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_pk -o p -- python3 tools/predict_once.py 128 > gpurun_out/${tag}_pred.txt 2> gpurun_out/${tag}_pred.err
cp gpurun_out/${tag}_pk/p_kernel_stats.csv gpurun_out/${tag}_predict_kernel_stats.csv
rm -rf gpurun_out/${tag}_pk
python3 - "gpurun_out/${tag}_predict_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"kernel time total {tot / 1e6:.1f} ms")
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:16]:
    print(f"{float(r['TotalDurationNs']) / tot * 100:5.1f}% {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:8.1f}us  {r['Name'][:100]}")
PY
cat gpurun_out/${tag}_pred.txt

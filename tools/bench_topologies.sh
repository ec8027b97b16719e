#!/bin/bash
# one default-size training-step bench line per topology / encoder (256^2, batch 32, bf16): gpurun -- 'bash tools/bench_topologies.sh r2'
set -e
tag=${1:-rX}
out=gpurun_out/${tag}_topologies.jsonl
: > $out
for t in unet unetplusplus linknet fpn deeplabv3plus deeplabv3 manet pan; do
  python bench.py --topology $t --steps 10 --warmup 3 --no-cpu-baseline --no-predict >> $out 2>> gpurun_out/${tag}_topologies.err
done
for e in resnet18 resnet50 resnext50_32x4d efficientnet-b3 efficientnet-b4 timm-resnest50d timm-resnest101e; do
  python bench.py --encoder $e --steps 10 --warmup 3 --no-cpu-baseline --no-predict >> $out 2>> gpurun_out/${tag}_topologies.err
done
python - "$out" <<'P'
import json, sys
print("topology\tencoder\tms_per_step\tslices_per_s\tstep_mode")
for line in open(sys.argv[1]):
    d = json.loads(line)
    model = d["config"]["workload"].split(", ")[2]
    print(f"{model.split('/')[0]}\t{model.split('/')[1]}\t{d['ms_per_step']:.3f}\t{d['value']:.0f}\t{d['step_timing']['mode']}")
P

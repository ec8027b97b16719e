# One short training-step bench at the full bench shape (256^2, batch 32, bf16) for encoder / decoder pairs the parity tests only see at 64^2,
# batch 4: size-dependent tile picks and buffer sizes (a 128-channel max-pool, a split data gradient at 32 + 40 channels) show up here.
#   gpurun -- "bash tools/smoke_matrix.sh"
set -e
for combo in "unetplusplus efficientnet-b3" "fpn efficientnet-b4" "manet efficientnet-b4" "pan efficientnet-b3" "deeplabv3 efficientnet-b3" "unetplusplus timm-resnest50d" "linknet timm-resnest101e" "manet timm-resnest50d" "fpn timm-resnest101e"; do
  set -- $combo
  timeout -k 10 240 python bench.py --topology $1 --encoder $2 --steps 3 --warmup 2 --no-cpu-baseline --no-predict > gpurun_out/smoke_$1_$2.json 2> gpurun_out/smoke_$1_$2.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/smoke_$1_$2.json')); print('$1 $2', d['ms_per_step'], d['config']['final_loss'])"
done

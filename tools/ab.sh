run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-predict 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items() if k.startswith('conv')}, d['roofline']['kernel'], d['roofline']['achieved'])"; }
run "VS_WGRAD_TARGET=128"
run "VS_WGRAD_TARGET=256"
run "VS_WGRAD_TARGET=512"
run "VS_WGRAD_TARGET=768"
run "VS_WGRAD_FAST=0"

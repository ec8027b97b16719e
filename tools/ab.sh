run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-predict 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items() if k.startswith('conv')}, d['roofline']['kernel'], d['roofline']['achieved'])"; }
run "VS_WGRAD_SLAB_MB=0"
run "VS_WGRAD_SLAB_MB=8"
run "VS_WGRAD_SLAB_MB=16"
run "VS_WGRAD_SLAB_MB=32"

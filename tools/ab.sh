run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-predict 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items() if k.startswith('bn')})"; }
run "VS_RECOMPUTE_MASK=0"
run "VS_RECOMPUTE_MASK=1"
run "VS_RECOMPUTE_MASK=1 VS_BN_BLOCKS=512"
run "VS_RECOMPUTE_MASK=1 VS_BN_BLOCKS=2048"

run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-predict 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items() if k.startswith('bn')})"; }
run "VS_BN_INLINE_ROWS=0"
run "VS_BN_INLINE_ROWS=64"
run "VS_BN_INLINE_ROWS=128"
run "VS_BN_INLINE_ROWS=0"
run "VS_BN_INLINE_ROWS=64"

run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-predict 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items() if k.startswith('conv')})"; }
run "VS_CONV_MIN_WGS=512"
run "VS_CONV_MIN_WGS=1000000000"
run "VS_CONV_MIN_WGS=2048"

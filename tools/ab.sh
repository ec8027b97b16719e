run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items()})"; }
run "VS_X=0"
run "VS_WGRAD_TARGET=256"
run "VS_WGRAD_TARGET=1024"
run "VS_CONV_MIN_WGS=0"
run "VS_CONV_MIN_WGS=1024"
run "VS_FUSE_STATS=0"
run "VS_RECOMPUTE_MASK=1"

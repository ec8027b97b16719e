run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-predict 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:v['ms_per_step'] for k,v in d['kernel_classes'].items() if k in ('conv_dgrad','bn_bwd')})"; }
run "VS_FUSE_BN_BWD=0"
run "VS_FUSE_BN_BWD=1"
run "VS_FUSE_BN_BWD=2"

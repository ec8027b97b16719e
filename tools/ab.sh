run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-predict 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run "VS_SIDE_STREAM=1 VS_WGRAD_TARGET=256"
run "VS_SIDE_STREAM=1 VS_WGRAD_TARGET=128"
run "VS_SIDE_STREAM=1 VS_WGRAD_TARGET=192"
run "VS_SIDE_STREAM=2 VS_WGRAD_TARGET=128"
run "VS_SIDE_STREAM=1 VS_WGRAD_TARGET=256 VS_CONV_MIN_WGS=256"

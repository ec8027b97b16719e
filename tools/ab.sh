run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-predict 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run "VS_FORK_EVERY=1"
run "VS_FORK_EVERY=2"
run "VS_FORK_EVERY=3"
run "VS_FORK_EVERY=4"
run "VS_FORK_EVERY=6"
run "VS_FORK_EVERY=1"

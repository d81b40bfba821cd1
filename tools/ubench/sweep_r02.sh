# A/B sweeps of kernel-plan switches on the headline workload (ResNet-50, B=256); one line per setting
run() { env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
run A=0
run TOK_WGRAD_SIDE_MAX_ROWS=0
run TOK_WGRAD_SIDE_MAX_ROWS=30000
run TOK_SHORT_K=200
run A=1

# A/B sweeps of kernel-plan switches on the headline workload (ResNet-50, B=256); one line per setting
run() { env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
run A=0
run TOK_SUBSAMPLE_MIN_ROWS=40000
run A=1
run TOK_SUBSAMPLE_MIN_ROWS=40000
run TOK_SUBSAMPLE_MIN_ROWS=10000

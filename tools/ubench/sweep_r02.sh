# A/B sweeps of the engine / kernel-plan switches on the headline workload (ResNet-50, B=256); one line per setting
run() { env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
run TOK_WGRAD_DEFER=0
run TOK_WGRAD_DEFER=3
run TOK_WGRAD_DEFER=5
run TOK_WGRAD_DEFER=8
run TOK_WGRAD_DEFER=0
run TOK_WGRAD_DEFER=3

cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $1 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
run "A=0"
run "TOK_PW_RING_MIN_ROWS=40000"
run "TOK_PW_RING_MIN_ROWS=10000"
run "TOK_SHORT_K=200"
run "TOK_SHORT_K=600"
run "TOK_WGRAD_SIDE_WHICH=all"
run "TOK_UNIT3_MIN_ROWS=40000"
run "TOK_WGRAD_SIDE_MAX_ROWS=300000"
run "TOK_BN_BLOCKS=2048"
run "A=1"

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 15 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$name', d['ms_per_step'])"; }
run base A=1
run rows0 TOK_WGRAD_SIDE_MAX_ROWS=0
run rows20k TOK_WGRAD_SIDE_MAX_ROWS=20000
run rows60k TOK_WGRAD_SIDE_MAX_ROWS=60000
run rows300k TOK_WGRAD_SIDE_MAX_ROWS=300000
run which1x1 TOK_WGRAD_SIDE_WHICH=1x1
run defer2 TOK_WGRAD_DEFER=2
run defer6 TOK_WGRAD_DEFER=6
run base2 A=1

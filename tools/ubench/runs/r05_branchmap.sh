#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { env "$@" timeout 400 python bench.py --no-cpu-baseline --no-secondary --backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$*', d['ms_per_step'])"; }
run A=1
run TOK_BRANCH_MAP=0,1,2,2
run TOK_BRANCH_MAP=0,1,1,2
run TOK_BRANCH_MAP=0,1,1,1
run TOK_BRANCH_MAP=0,0,1,2
run TOK_BRANCH_MAP=0,1,2,1
run A=1

#!/bin/bash
# ablation by skipping launch classes (TOK_DBG_SKIP, results garbage): upper bounds on what removing each class can buy
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05b; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 15"
run() { name=$1; shift; env "$@" timeout 300 $B ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'])
except Exception as e: print('$name', 'FAILED', e)
PY
}
run base A=1
run skip_finalize TOK_DBG_SKIP=1
run skip_reduce TOK_DBG_SKIP=2
run skip_unit3help TOK_DBG_SKIP=4
run skip_bnapply TOK_DBG_SKIP=8
run skip_wgrad TOK_DBG_SKIP=16
run skip_wgrad_side0 TOK_DBG_SKIP=16 TOK_WGRAD_SIDE=0
run noside TOK_WGRAD_SIDE=0
run skip_1_2_4 TOK_DBG_SKIP=7
run base2 A=1
EXTRA="--backbone swinv2_custom"
run swin_base A=1
run swin_skip_reduce_colsum TOK_DBG_SKIP=6
run swin_skip_wgrad TOK_DBG_SKIP=16

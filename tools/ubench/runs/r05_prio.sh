#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05prio; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-secondary ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d.get('final_loss'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 20 --warmup 5"
run hr_base A=1
run hr_prio TOK_STREAM_PRIO=-1
run hr_base2 A=1
run hr_prio2 TOK_STREAM_PRIO=-1
run hr_nobranch TOK_BRANCH_STREAMS=0
EXTRA="--steps 60 --warmup 15"
run res_base A=1
run res_prio TOK_STREAM_PRIO=-1
EXTRA="--backbone swinv2_custom --steps 40 --warmup 10"
run swin_base A=1
run swin_prio TOK_STREAM_PRIO=-1

#!/bin/bash
# which weight gradients go to the side stream, re-measured on the round-5 tree (same box, ms/step)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05side2; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-secondary ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'])
except Exception as e: print('$name', 'FAILED', e)
PY
}
EXTRA="--steps 60 --warmup 15"
run r50_base A=1
run r50_all TOK_WGRAD_SIDE_WHICH=all
run r50_rows200k TOK_WGRAD_SIDE_MAX_ROWS=250000
run r50_rows900k TOK_WGRAD_SIDE_MAX_ROWS=900000
run r50_rows30k TOK_WGRAD_SIDE_MAX_ROWS=30000
run r50_base2 A=1
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4"
run hr_base A=1
run hr_all TOK_WGRAD_SIDE_WHICH=all
run hr_base2 A=1
EXTRA="--backbone swinv2_custom --steps 40 --warmup 10"
run sw_base A=1
run sw_rows30k TOK_WGRAD_SIDE_MAX_ROWS=30000
run sw_rows250k TOK_WGRAD_SIDE_MAX_ROWS=250000
run sw_base2 A=1

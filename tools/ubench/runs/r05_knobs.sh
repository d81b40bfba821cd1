#!/bin/bash
# engine knobs whose defaults were picked in rounds 2-3, re-measured on the round-5 tree (same box per block, ms/step)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05knobs; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-secondary ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'])
except Exception as e: print('$name', 'FAILED', e)
PY
}
EXTRA="--steps 60 --warmup 15"
run r50_base A=1
run r50_unit3_rows0 TOK_UNIT3_MIN_ROWS=0
run r50_unit3_rows40k TOK_UNIT3_MIN_ROWS=40000
run r50_unit3_rows250k TOK_UNIT3_MIN_ROWS=250000
run r50_subsample_rows0 TOK_SUBSAMPLE_MIN_ROWS=0
run r50_shortcut_early TOK_SHORTCUT_LATE=0
run r50_shortcut_branch TOK_SHORTCUT_BRANCH=1
run r50_dgrad2_off TOK_DGRAD2=0
run r50_colsum_in_act_off TOK_COLSUM_IN_ACT=0
run r50_pooled_stats_off TOK_STEM_POOLED_STATS=0
run r50_wgrad_defer3 TOK_WGRAD_DEFER=3
run r50_base2 A=1
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4"
run hr_base A=1
run hr_pick_streams_off TOK_PICK_STREAMS=0
run hr_lazy_events_off TOK_LAZY_EVENTS=0
run hr_wgrad_defer3 TOK_WGRAD_DEFER=3
run hr_base2 A=1
EXTRA="--backbone swinv2_custom --steps 40 --warmup 10"
run sw_base A=1
run sw_dgrad_first TOK_DGRAD_FIRST=1
run sw_bias_in_wgrad_off TOK_BIAS_IN_WGRAD=0
run sw_base2 A=1

#!/bin/bash
# split targets of the weight-gradient kernels, re-swept on the round-5 tree (same box, ms/step)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05wgsweep; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-secondary ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'])
except Exception as e: print('$name', 'FAILED', e)
PY
}
EXTRA="--steps 60 --warmup 15"
run r50_base A=1
for v in 256 384 768; do run r50_ring$v TOK_WGRAD_WGS=$v; done
for v in 128 384 512; do run r50_taps$v TOK_WGRAD_TAPS_WGS=$v; done
run r50_base2 A=1
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4"
run hr_base A=1
for v in 128 384 512; do run hr_taps$v TOK_WGRAD_TAPS_WGS=$v; done
for v in 256 768; do run hr_ring$v TOK_WGRAD_WGS=$v; done
run hr_base2 A=1
EXTRA="--backbone swinv2_custom --steps 40 --warmup 10"
run sw_base A=1
for v in 256 384 768; do run sw_ring$v TOK_WGRAD_WGS=$v; done
run sw_base2 A=1

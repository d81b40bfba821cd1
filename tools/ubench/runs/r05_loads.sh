#!/bin/bash
# unconditional (batched) loads in the pooled-stem forward, the depthwise 3x3 and the quarter-tile epilogue: parity + A/B
# (libtok_ab.so = those four files from the previous commit)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05loads; mkdir -p $O
timeout 1500 python -m pytest tests/test_pooling.py tests/test_davit.py tests/test_kernels_gpu.py tests/test_resnet_gpu.py tests/test_golden_gpu.py tests/test_hrnet.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-secondary ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d.get('final_loss'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
for i in 1 2; do
EXTRA="--steps 60 --warmup 15"
run r50_new_$i A=1
run r50_old_$i TOK_LIB=torchok_amd/lib/libtok_ab.so
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4"
run hr_new_$i A=1
run hr_old_$i TOK_LIB=torchok_amd/lib/libtok_ab.so
EXTRA="--backbone davit_t --steps 30 --warmup 10"
run dv_new_$i A=1
run dv_old_$i TOK_LIB=torchok_amd/lib/libtok_ab.so
done

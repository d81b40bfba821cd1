#!/bin/bash
# in-step A/B of the stride-2 data-gradient window kernel (TOK_CONV_S2D) on ResNet-50 and HRNet-W48, same box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05s2d; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-secondary ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d.get('final_loss'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
for i in 1 2; do
EXTRA="--steps 60 --warmup 15"
run r50_s2d$i A=1
run r50_igemm$i TOK_CONV_S2D=0
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4"
run hr_s2d$i A=1
run hr_igemm$i TOK_CONV_S2D=0
done
timeout 1200 python -m pytest tests/test_golden_gpu.py tests/test_resnet_gpu.py tests/test_hrnet.py tests/test_units_real_gpu.py tests/test_default_dispatch_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt

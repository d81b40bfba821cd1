#!/bin/bash
# same-box A/B of the working-tree library against torchok_amd/lib/libtok_ab.so (tools/ab_lib.sh); optional kernel tests first
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05ab; mkdir -p $O
if [ -n "$KTESTS" ]; then timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "$KTESTS" > $O/ktest.txt 2>&1; tail -3 $O/ktest.txt; fi
B="python bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-60} --warmup 15"
run() { name=$1; shift; env "$@" timeout 300 $B ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d.get('final_loss'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
for i in 1 2; do
run new$i A=1
run old$i TOK_LIB=torchok_amd/lib/libtok_ab.so
done
if [ -n "$SWIN" ]; then
EXTRA="--backbone swinv2_custom"
for i in 1 2; do
run swin_new$i A=1
run swin_old$i TOK_LIB=torchok_amd/lib/libtok_ab.so
done
fi
if [ -n "$HRNET" ]; then
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 20 --warmup 5"
run hr_new A=1
run hr_old TOK_LIB=torchok_amd/lib/libtok_ab.so
fi
if [ -n "$TESTS" ]; then timeout 2400 python -m pytest $TESTS -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt; fi

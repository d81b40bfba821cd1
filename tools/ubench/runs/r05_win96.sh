#!/bin/bash
# 96-channel tiles of the shared-window 3x3 kernel (TOK_CONV_WIN_96): parity tests, per-layer A/B, HRNet-W48 step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05win96; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv_fwd or conv_dgrad" > $O/ktest.txt 2>&1; tail -3 $O/ktest.txt
for v in 0 1; do echo "== TOK_CONV_WIN_96=$v"; TOK_CONV_WIN_96=$v timeout 300 python tools/bench_conv.py --net hrnet_w48 --batch 24 --what fwd,dgrad 2>&1 | grep -v amdgpu.ids; done | tee $O/layers.txt
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-secondary ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d.get('final_loss'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4"
for i in 1 2; do
run hr_96_$i A=1
run hr_128_$i TOK_CONV_WIN_96=0
done

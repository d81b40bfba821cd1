#!/bin/bash
# ring depth of the shared-window weight-gradient kernel (TOK_WGRAD_WIN_NST=3|4): parity, per-layer A/B, steps
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05wgnst; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wgrad" > $O/ktest.txt 2>&1; tail -3 $O/ktest.txt
for v in 3 4; do echo "== TOK_WGRAD_WIN_NST=$v"; TOK_WGRAD_WIN_NST=$v timeout 300 python tools/bench_conv.py --net hrnet_w48 --batch 24 --what wgrad 2>&1 | grep -v amdgpu.ids;  TOK_WGRAD_WIN_NST=$v timeout 300 python tools/bench_conv.py --what wgrad 2>&1 | grep ", 3, 1)\|wgrad:"; done | tee $O/layers.txt
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-secondary ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d.get('final_loss'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
for i in 1 2; do
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4"
run hr_nst4_$i A=1
run hr_nst3_$i TOK_WGRAD_WIN_NST=3
EXTRA="--steps 60 --warmup 15"
run r50_nst4_$i A=1
run r50_nst3_$i TOK_WGRAD_WIN_NST=3
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05wreg; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv" > $O/ktest.txt 2>&1; tail -3 $O/ktest.txt
for v in 1 0; do
  echo "== WREG=$v hrnet fwd"; TOK_CONV_WIN_WREG=$v python tools/bench_conv.py --what fwd,dgrad --net hrnet_w48 --batch 24 2>&1 | grep -v amdgpu | tail -7
  echo "== WREG=$v resnet 3x3"; TOK_CONV_WIN_WREG=$v python tools/bench_conv.py --what fwd,dgrad 2>&1 | grep ", 3, 1)"
done
B="python bench.py --no-cpu-baseline --no-secondary"
run() { name=$1; shift; env "$@" timeout 300 $B ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d.get('final_loss'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
EXTRA="--steps 60 --warmup 15"
run res_wreg A=1
run res_lds TOK_CONV_WIN_WREG=0
run res_wreg2 A=1
run res_lds2 TOK_CONV_WIN_WREG=0
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 20 --warmup 5"
run hr_wreg A=1
run hr_lds TOK_CONV_WIN_WREG=0

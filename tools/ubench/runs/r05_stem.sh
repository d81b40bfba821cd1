#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05stem; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "stem" > $O/ktest.txt 2>&1; tail -4 $O/ktest.txt
for v in 1 0; do echo "== TOK_STEM_WIN=$v"; TOK_STEM_WIN=$v python tools/bench_conv.py --what fwd,wgrad 2>&1 | grep "(224, 224"; done
B="python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 15"
run() { name=$1; shift; env "$@" timeout 300 $B 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d.get('final_loss'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
for i in 1 2; do
run stem_all$i A=1
run stem_fwd_only$i TOK_STEM_WGRAD=0
run stem_none$i TOK_STEM_WIN=0
done
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_resnet_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt

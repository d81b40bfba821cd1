#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "folded_into_apply or bn_chain" > $O/ktest.txt 2>&1; tail -4 $O/ktest.txt
B="python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 15"
run() { name=$1; shift; env "$@" timeout 300 $B ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d.get('phase_sync_errors'), d.get('final_loss'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
run fused A=1
run unfused TOK_FUSE_FIN_APPLY=0
run fused2 A=1
run unfused2 TOK_FUSE_FIN_APPLY=0
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 20 --warmup 5"
run hr_fused A=1
run hr_unfused TOK_FUSE_FIN_APPLY=0
timeout 1200 python -m pytest tests/test_golden_gpu.py tests/test_resnet_gpu.py tests/test_fullsize_properties_gpu.py tests/test_hrnet.py -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt

#!/bin/bash
# batched loads in bilinear_bwd / fuse_sum_relu_{fwd,bwd}: parity, then HRNet-W48 steps (libtok_ab.so = resample.hip of the previous commit)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05resample; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_units_gpu.py tests/test_hrnet.py tests/test_golden_gpu.py -x -q -m gpu -k "bilinear or fuse or hrnet or golden or seg or upsample" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-secondary ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d.get('final_loss'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4"
for i in 1 2 3; do
run hr_new_$i A=1
run hr_old_$i TOK_LIB=torchok_amd/lib/libtok_ab.so
done

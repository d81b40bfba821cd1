#!/bin/bash
# one BatchNorm-sum butterfly per kernel (narrow tiles) vs one per tile: libtok_ab.so = conv_win / conv_s2d of the previous commit
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05pst; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv_fwd or conv_dgrad" > $O/ktest.txt 2>&1; tail -2 $O/ktest.txt
for l in libtok_ab libtok_gfx950; do echo "== $l"; timeout 300 python tools/bench_conv.py --lib torchok_amd/lib/$l.so --net hrnet_w48 --batch 24 --what fwd,dgrad 2>&1 | grep -v amdgpu.ids; done | tee $O/layers.txt
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-secondary ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d.get('final_loss'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
for i in 1 2; do
EXTRA="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4"
run hr_new_$i A=1
run hr_old_$i TOK_LIB=torchok_amd/lib/libtok_ab.so
EXTRA="--steps 60 --warmup 15"
run r50_new_$i A=1
run r50_old_$i TOK_LIB=torchok_amd/lib/libtok_ab.so
done

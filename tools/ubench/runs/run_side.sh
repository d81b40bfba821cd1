#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/side
for rep in 1 2; do
for cfg in "base:TOK_X=0" "all:TOK_WGRAD_SIDE_WHICH=all" "rows250k:TOK_WGRAD_SIDE_MAX_ROWS=250000" "rows50k:TOK_WGRAD_SIDE_MAX_ROWS=40000"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  ( env $envs timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary ) > gpurun_out/side/r_${name}_$rep.json 2> gpurun_out/side/r_${name}_$rep.err
  python - <<PY
import json
try:
    j=json.loads(open('gpurun_out/side/r_${name}_$rep.json').read().strip().splitlines()[-1]); print('resnet50 ${name} rep $rep', j['ms_per_step'])
except Exception as e: print('${name} failed', e)
PY
done; done

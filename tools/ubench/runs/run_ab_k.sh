#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab_k
SW="--backbone swinv2_custom --steps 30 --warmup 10 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
for cfg in "base:TOK_GEMM256=1" "k192:TOK_GEMM256=2 TOK_GEMM256_MIN_K=192" "k96:TOK_GEMM256=2 TOK_GEMM256_MIN_K=96" "all:TOK_GEMM256=3"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  ( env $envs timeout 300 python bench.py $SW ) > gpurun_out/ab_k/swin_${name}_$rep.json 2> gpurun_out/ab_k/swin_${name}_$rep.err
  python - <<PY
import json
try:
    j=json.loads(open('gpurun_out/ab_k/swin_${name}_$rep.json').read().strip().splitlines()[-1]); print('swin ${name} rep $rep', j['ms_per_step'])
except Exception as e: print('swin ${name} failed', e)
PY
done; done

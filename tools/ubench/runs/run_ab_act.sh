#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab_act
echo skip check
SW="--backbone swinv2_custom --steps 30 --warmup 10 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
for cfg in "base:TOK_GEMM256_ACT=0" "act:TOK_GEMM256_ACT=1" "s3unfused:TOK_MLP_MAX_C=192" "s23unfused:TOK_MLP_MAX_C=96" ; do
  name=${cfg%%:*}; envs=${cfg#*:}
  ( env $envs timeout 300 python bench.py $SW ) > gpurun_out/ab_act/swin_${name}_$rep.json 2> gpurun_out/ab_act/swin_${name}_$rep.err
  python - <<PY
import json
try:
    j=json.loads(open('gpurun_out/ab_act/swin_${name}_$rep.json').read().strip().splitlines()[-1]); print('swin ${name} rep $rep', j['ms_per_step'])
except Exception as e: print('swin ${name} failed', e)
PY
done; done
DV="--backbone davit_t --steps 30 --warmup 10 --no-cpu-baseline --no-secondary"
for cfg in "base:TOK_GEMM256_ACT=0" "act:TOK_GEMM256_ACT=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  ( env $envs timeout 300 python bench.py $DV ) > gpurun_out/ab_act/davit_${name}.json 2> gpurun_out/ab_act/davit_${name}.err
  python - <<PY
import json
try:
    j=json.loads(open('gpurun_out/ab_act/davit_${name}.json').read().strip().splitlines()[-1]); print('davit ${name}', j['ms_per_step'])
except Exception as e: print('davit ${name} failed', e)
PY
done



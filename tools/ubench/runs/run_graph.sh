#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/graph
for cfg in "b256_eager:--batch 256 --graph 0" "b256_graph:--batch 256 --graph 1" "b128_eager:--batch 128 --graph 0" "b128_graph:--batch 128 --graph 1" "b64_eager:--batch 64 --graph 0" "b64_graph:--batch 64 --graph 1"; do
  name=${cfg%%:*}; args=${cfg#*:}
  ( timeout 300 python bench.py $args --steps 40 --warmup 10 --no-cpu-baseline --no-secondary ) > gpurun_out/graph/${name}.json 2> gpurun_out/graph/${name}.err
  python - <<PY
import json
try:
    j=json.loads(open('gpurun_out/graph/${name}.json').read().strip().splitlines()[-1]); print('resnet50 ${name}', j['ms_per_step'], j['config'].get('launch_mode'))
except Exception as e: print('${name} failed', e)
PY
done

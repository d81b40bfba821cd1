#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05env; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-secondary ${EXTRA} 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'])
except Exception as e: print('$name', 'FAILED', e)
PY
}
EXTRA="--steps 60 --warmup 15"
run base A=1
run expandable PYTORCH_HIP_ALLOC_CONF=expandable_segments:True
run noxnack HSA_XNACK=0
run sdma0 HSA_ENABLE_SDMA=0
run nocoop HIP_LAUNCH_BLOCKING=0 AMD_SERIALIZE_KERNEL=0
run base2 A=1
EXTRA="--backbone swinv2_custom --steps 40 --warmup 10"
run swin_base A=1
run swin_expandable PYTORCH_HIP_ALLOC_CONF=expandable_segments:True

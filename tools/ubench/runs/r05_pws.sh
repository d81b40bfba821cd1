#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05pws; mkdir -p $O
timeout 900 python -m pytest tests/test_pw_stream_gpu.py -x -q -m gpu > $O/test.txt 2>&1; tail -5 $O/test.txt
python tools/ubench/pw_probe.py stream > $O/probe_stream.txt 2>&1; TOK_PW_STREAM=0 python tools/ubench/pw_probe.py ring > $O/probe_ring.txt 2>&1
grep "M=" $O/probe_stream.txt $O/probe_ring.txt
B="python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 15"
run() { name=$1; shift; env "$@" timeout 300 $B 2>$O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d.get('final_loss'))
except Exception as e: print('$name', 'FAILED', e)
PY
}
for i in 1 2; do
run stream$i A=1
run ring$i TOK_PW_STREAM=0
done
run stream_all TOK_PW_STREAM=2

#!/bin/bash
# round 5, first GPU call: changed tests + baseline bench + cheap knob A/B (same box)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05a
O=gpurun_out/r05a
timeout 900 python -m pytest tests/test_dice.py tests/test_ddp_two_proc_gpu.py tests/test_swin.py tests/test_cabi.py -x -q -m gpu > $O/tests.txt 2>&1
tail -5 $O/tests.txt
B="python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 15"
run() { name=$1; shift; env "$@" timeout 300 $B 2>/dev/null | tail -1 > $O/$name.json; python - <<PY
import json
d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['roofline']['frac'])
PY
}
run base A=1
run unit3_40k TOK_UNIT3_MIN_ROWS=40000
run unit3_10k TOK_UNIT3_MIN_ROWS=10000
run wgs384 TOK_WGRAD_WGS=384
run wgs256 TOK_WGRAD_WGS=256
run taps128 TOK_WGRAD_TAPS_WGS=128
run bn2048 TOK_BN_BLOCKS=2048
run bn768 TOK_BN_BLOCKS=768
run base2 A=1

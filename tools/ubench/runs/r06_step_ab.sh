#!/bin/bash
# same-box A/B of an env switch on the three workloads: r06_step_ab.sh "<env A>" "<env B>"  (e.g. "TOK_WGRAD_WINP=0" "TOK_WGRAD_WINP=1")
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_step_ab; mkdir -p $O
run() { env $1 python bench.py $2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', '$3', d['ms_per_step'], 'ms/step', d['roofline']['frac'])"; }
for rep in 1 2; do
for e in "$1" "$2"; do
  run "$e" "--steps 60 --warmup 15" resnet50
  run "$e" "--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 12 --warmup 4" hrnet_w48
  run "$e" "--backbone swinv2_custom --steps 30 --warmup 10" swinv2_t
done; done 2>&1 | tee $O/last.txt

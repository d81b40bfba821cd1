#!/bin/bash
# same-box sweep of env settings on one workload: r06_env_sweep.sh <resnet50|hrnet_w48|swinv2_t> "<env1>" "<env2>" ...   (two rounds, interleaved)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_env_sweep; mkdir -p $O
w=$1; shift
case $w in
  resnet50) A="--steps 60 --warmup 15";;
  hrnet_w48) A="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 12 --warmup 4";;
  swinv2_t) A="--backbone swinv2_custom --steps 30 --warmup 10";;
esac
for rep in 1 2; do
  for e in "$@"; do
    env $e python bench.py $A --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$w', '$e', d['ms_per_step'], 'ms/step', d['roofline']['frac'])"
  done
done 2>&1 | tee -a $O/$w.txt

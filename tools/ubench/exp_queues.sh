#!/bin/bash
# GPU_MAX_HW_QUEUES sweep (ROCclr: hardware queues per process; default 4)
cd $GRAFT_REPO_ROOT; export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
S="--backbone swinv2_custom --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
H="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 10 --warmup 4 --no-cpu-baseline --no-secondary"
R="--steps 40 --warmup 10 --no-cpu-baseline --no-secondary"
for q in ${QUEUES:-4 8 16}; do
  export GPU_MAX_HW_QUEUES=$q
  python bench.py $S 2>/dev/null | ms "queues=$q swin"
  python bench.py $H 2>/dev/null | ms "queues=$q hrnet"
  python bench.py $R 2>/dev/null | ms "queues=$q resnet"
  TOK_BENCH_FORCE_DIST=1 python bench.py $R 2>/dev/null | ms "queues=$q resnet dist"
done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/h1; mkdir -p $o
H="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 10 --warmup 4 --no-cpu-baseline --no-secondary"
for i in 1 2; do
for v in 1 0; do TOK_LAZY_EVENTS=$v python bench.py $H 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lazy=$v', d['ms_per_step'])"; done
done
timeout 300 rocprofv3 --kernel-trace -d $o/raw -o kt -- python bench.py $H --steps 4 --warmup 3 > $o/kt.log 2>&1
db=$(ls $o/raw/*results.db | head -1); python tools/timeline.py $db --dump > $o/hrnet_dump.txt 2>&1; rm -rf $o/raw
head -12 $o/hrnet_dump.txt

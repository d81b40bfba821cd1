#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r05dump; mkdir -p $o
run() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace -d $o/raw_$tag -o kt -- "$@" > $o/$tag.log 2>&1; db=$(ls $o/raw_$tag/*results.db | head -1); python tools/timeline.py $db --dump > $o/${tag}_dump.txt 2>&1; rm -rf $o/raw_$tag; }
run res2 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary
TOK_WGRAD_SIDE=0 run res1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary
run swin2 python bench.py --backbone swinv2_custom --steps 6 --warmup 3 --no-cpu-baseline --no-secondary
TOK_WGRAD_SIDE=0 run swin1 python bench.py --backbone swinv2_custom --steps 6 --warmup 3 --no-cpu-baseline --no-secondary
ls -la $o

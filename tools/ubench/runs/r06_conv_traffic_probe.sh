#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per dispatch of the 3x3 window kernels alone (counter units: see tools/pmc_traffic.py — FETCH_SIZE in 32-byte... raw values here)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/pmc_probe; mkdir -p $o
run() { tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 150 rocprofv3 --pmc $c -d $o/$tag-$c -o p -- "$@" > $o/$tag-$c.log 2>&1
    db=$(ls $o/$tag-$c/*results.db | head -1)
    for k in conv_win conv_wgrad_winp wgrad_reduce; do python tools/pmc_kernel.py $db $k 2>/dev/null | sed "s/^/$tag $k: /"; done
    rm -rf $o/$tag-$c
  done; }
run fwd48 python tools/ubench/one_conv.py 12 256 48 48 3 1 fwd 5
run fwd64 python tools/ubench/one_conv.py 12 256 64 64 3 1 fwd 5
run fwd96 python tools/ubench/one_conv.py 12 128 96 96 3 1 fwd 5
run wgrad48 python tools/ubench/one_conv.py 12 256 48 48 3 1 wgrad 5

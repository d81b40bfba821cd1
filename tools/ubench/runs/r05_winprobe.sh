#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05win; mkdir -p $O
for m in ${PROBES:-base}; do
  lib=torchok_amd/lib/libtok_winprobe_$m.so; [ $m = base ] && lib=torchok_amd/lib/libtok_gfx950.so
  echo "== probe $m (hrnet B=24 fwd)"; python tools/bench_conv.py --lib $lib --what fwd --net hrnet_w48 --batch 24 2>&1 | grep -v amdgpu | tail -6
  echo "== probe $m (resnet50 fwd 3x3)"; python tools/bench_conv.py --lib $lib --what fwd 2>&1 | grep ", 3, 1)" 
done > $O/probe2.txt 2>&1
cat $O/probe2.txt

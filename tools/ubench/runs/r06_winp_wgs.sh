#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_winp_wgs; mkdir -p $O
for t in 256 384 512; do
  TOK_WGRAD_TAPS_WGS=$t timeout 600 python tools/bench_conv.py --what wgrad --net resnet50 > $O/resnet50_$t.txt 2>&1
  TOK_WGRAD_TAPS_WGS=$t timeout 600 python tools/bench_conv.py --what wgrad --net hrnet_w48 --batch 24 > $O/hrnet_$t.txt 2>&1
  echo "== target $t"; grep -h ', 3, 1)' $O/resnet50_$t.txt $O/hrnet_$t.txt
done

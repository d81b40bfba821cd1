#!/bin/bash
# PMC passes over the pipelined window weight-gradient kernel on ResNet-50's 14x14x256 layer (and the round-5 kernel beside it)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_winp_pmc; mkdir -p $O
for v in 1 0; do
  TOK_WGRAD_WINP=$v bash tools/ubench/pmc_conv.sh $O/raw$v 256 14 256 256 3 1 wgrad 5 > /dev/null 2>&1
  { echo "# TOK_WGRAD_WINP=$v, (256,14,14,256,256,3,1) wgrad; per-dispatch averages";
    for g in 1 2 3 4 5 6 7; do db=$(ls $O/raw$v/g$g/*results.db 2>/dev/null | head -1); [ -n "$db" ] && python tools/pmc_kernel.py $db conv_wgrad_win; done; } > $O/pmc_winp$v.txt
  rm -rf $O/raw$v
done
cat $O/pmc_winp1.txt; cat $O/pmc_winp0.txt

#!/bin/bash
# fused-MLP timing probes (libtok_probeN.so built with -DTOK_MLP_PROBE=N)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/mlp
{
for n in 0 1 2 3 4; do
  lib=$GRAFT_REPO_ROOT/torchok_amd/lib/libtok_probe$n.so; [ $n = 0 ] && lib=$GRAFT_REPO_ROOT/torchok_amd/lib/libtok_gfx950.so
  echo "== probe $n"
  for shape in "50176 384" "200704 192" "802816 96"; do TOK_LIB=$lib timeout 300 python tools/ubench/mlp_check.py $shape 2>&1 | grep -v amdgpu.ids | sed 's/.*| fused/fused/'; done
done
} > gpurun_out/mlp/probes.txt 2>&1
cat gpurun_out/mlp/probes.txt

#!/bin/bash
# profiles/r06_layer_tables.txt: isolated per-layer numbers of the final library
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O
{ echo "# tools/bench_conv.py on the final round-6 library, isolated launches (8 iterations each): us per call, algorithmic GB/s (|X|+|Y|), TFLOP/s; forward includes the BatchNorm partial sums, wgrad its slab fold";
  echo "## ResNet-50 B=256"; python tools/bench_conv.py --net resnet50 2>&1 | grep -v amdgpu.ids;
  echo "## HRNet-W48 512x1024 B=24 (the 3x3 / stride-1 layers of the four branches)"; python tools/bench_conv.py --net hrnet_w48 --batch 24 2>&1 | grep -v amdgpu.ids;
  echo "## SwinV2-T B=256 (token GEMMs as 1x1 layers; the Mlp pair runs fused in the step)"; python tools/bench_conv.py --net swinv2t 2>&1 | grep -v amdgpu.ids;
  echo "## stride-2 3x3 data gradients (tools/ubench/s2d_ab.py)"; python tools/ubench/s2d_ab.py 2>&1 | grep -v amdgpu.ids | head -30;
  echo "## window attention / LayerNorm / fused Mlp launches of a SwinV2-T step in isolation"; python tools/ubench/attn_time.py 2>&1 | grep -v amdgpu.ids | tail -6; python tools/ubench/ln_time.py 2>&1 | grep -v amdgpu.ids | tail -8; for a in "802816 96" "200704 192" "50176 384"; do python tools/ubench/mlp_time.py torchok_amd/lib/libtok_gfx950.so $a 2>&1 | grep -v amdgpu.ids | tail -2; done;
} > $O/r06_layer_tables.txt 2>&1
tail -5 $O/r06_layer_tables.txt

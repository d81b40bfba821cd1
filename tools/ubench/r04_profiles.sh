#!/bin/bash
# round-4 artifact set: bench line + kernel stats + timeline + main-queue gaps + PMC traffic (calibrated on the optimizer kernel) / MFMA
# for the three workloads, the MLP launches in isolation (recompute plan evidence), the token-GEMM budget, host phases
cd $GRAFT_REPO_ROOT; o=gpurun_out/r04; mkdir -p $o
bash tools/profile_workload.sh r04_resnet50_bs256 $o --steps 50 --warmup 10 > $o/p1.log 2>&1
bash tools/profile_workload.sh r04_swinv2t_224_bs256 $o --backbone swinv2_custom --steps 30 --warmup 10 > $o/p2.log 2>&1
bash tools/profile_workload.sh r04_hrnet_w48_512x1024_bs24 $o --backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 10 --warmup 4 > $o/p3.log 2>&1
python bench.py --backbone davit_t --steps 30 --warmup 10 --no-cpu-baseline --no-secondary > $o/r04_davit_t_224_bs256_bench.json 2> /dev/null
{ echo "# isolated per-call times of the Mlp launches at the SwinV2-T B=256 stage shapes (tools/ubench/mlp_dw_time.py)"; python tools/ubench/mlp_dw_time.py 2>&1 | grep -v amdgpu.ids;
  echo "# SwinV2-T B=256 step, recompute plan off (default) / on";
  for v in 0 1; do TOK_MLP_RECOMPUTE=$v python bench.py --backbone swinv2_custom --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('TOK_MLP_RECOMPUTE=$v', d['ms_per_step'], 'ms/step')"; done; } > $o/r04_mlp_recompute_ab.txt 2>&1
{ echo "# isolated token GEMMs of one SwinV2-T block per stage (tools/ubench/swin_budget.py)"; python tools/ubench/swin_budget.py 2>&1 | grep -v amdgpu.ids; } > $o/r04_swin_token_gemm_budget.txt 2>&1
{ echo "# launch-thread time per phase of train_step at batch 2 (tools/ubench/host_phases.py)"; for w in resnet50 swinv2_custom hrnet_w48; do python tools/ubench/host_phases.py $w 2 2>&1 | grep "host ms"; done; } > $o/r04_host_phases.txt 2>&1
{ echo "# HRNet-W48 B=24 step with the head's interpolation fused into the loss (default) / TOK_FUSE_UPSAMPLE_CE=0";
  for v in 1 0; do TOK_FUSE_UPSAMPLE_CE=$v python bench.py --backbone hrnet_w48 --res 512 --width 1024 --classes 19 --batch 24 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('TOK_FUSE_UPSAMPLE_CE=$v', d['ms_per_step'], 'ms/step')"; done; } > $o/r04_upsample_ce_ab.txt 2>&1
ls $o

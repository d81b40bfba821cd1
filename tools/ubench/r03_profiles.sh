#!/bin/bash
# round-3 artifact set: bench line + kernel stats + timeline + PMC traffic / MFMA for the four workloads
cd $GRAFT_REPO_ROOT; o=gpurun_out/r03; mkdir -p $o
bash tools/profile_workload.sh r03_resnet50_bs256 $o --steps 50 --warmup 10 > $o/p1.log 2>&1
bash tools/profile_workload.sh r03_swinv2t_224_bs256 $o --backbone swinv2_custom --steps 30 --warmup 10 > $o/p2.log 2>&1
bash tools/profile_workload.sh r03_hrnet_w48_512x1024_bs24 $o --backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 10 --warmup 4 > $o/p3.log 2>&1
python bench.py --backbone davit_t --steps 30 --warmup 10 --no-cpu-baseline --no-secondary > $o/r03_davit_t_224_bs256_bench.json 2> /dev/null
ls $o

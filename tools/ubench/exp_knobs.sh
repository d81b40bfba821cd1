cd $GRAFT_REPO_ROOT
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
S="--backbone swinv2_custom --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
H="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 10 --warmup 4 --no-cpu-baseline --no-secondary"
R="--steps 40 --warmup 10 --no-cpu-baseline --no-secondary"
for e in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_HIP_DYNAMIC_QUEUES=1" "AMD_SERIALIZE_KERNEL=0 HSA_NO_SCRATCH_RECLAIM=1"; do
  env $e python bench.py $R 2>/dev/null | ms "$e resnet"
  env $e python bench.py $S 2>/dev/null | ms "$e swin"
  env $e python bench.py $H 2>/dev/null | ms "$e hrnet"
done

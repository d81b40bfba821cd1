#!/bin/bash
# round-6 artifact set: bench line + kernel stats + timeline + main-queue gaps + PMC traffic (calibrated on the optimizer kernel) / MFMA
# for the three workloads, the default bench line, the launch-class ablation (TOK_DBG_SKIP)
cd $GRAFT_REPO_ROOT; o=gpurun_out/r06; mkdir -p $o
bash tools/profile_workload.sh r06_resnet50_bs256 $o --steps 50 --warmup 10 > $o/p1.log 2>&1
bash tools/profile_workload.sh r06_swinv2t_224_bs256 $o --backbone swinv2_custom --steps 30 --warmup 10 > $o/p2.log 2>&1
bash tools/profile_workload.sh r06_hrnet_w48_512x1024_bs24 $o --backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 10 --warmup 4 > $o/p3.log 2>&1
python bench.py > $o/r06_bench_default_line.json 2> $o/bench_default.err
python bench.py --backbone davit_t --steps 30 --warmup 10 --no-cpu-baseline --no-secondary > $o/r06_davit_t_224_bs256_bench.json 2> /dev/null
{ echo "# launch-class ablation on the ResNet-50 B=256 step (TOK_DBG_SKIP: the named launches are not issued, results garbage, ms/step = upper bound on what removing the class can buy; bench.py --steps 60 --warmup 15, same box)";
  for v in "0 base" "1 BatchNorm_finalizes(88_launches)" "2 wgrad_reduce_flat(61)" "4 fused-unit_helpers+colsum_f32(46)" "7 all_three" "8 BatchNorm_apply_passes(86)" "16 every_weight_gradient"; do set -- $v
    TOK_DBG_SKIP=$1 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('TOK_DBG_SKIP=$1 $2', d['ms_per_step'], 'ms/step')"; done
  TOK_DBG_SKIP=16 TOK_WGRAD_SIDE=0 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('TOK_DBG_SKIP=16 TOK_WGRAD_SIDE=0 (no weight gradients, no side-stream forks)', d['ms_per_step'], 'ms/step')"
  TOK_WGRAD_SIDE=0 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('TOK_WGRAD_SIDE=0 (single stream)', d['ms_per_step'], 'ms/step')"
  echo "# SwinV2-T B=256";
  for v in "0 base" "6 wgrad_reduce_flat+colsum_f32" "16 every_weight_gradient"; do set -- $v
    TOK_DBG_SKIP=$1 python bench.py --backbone swinv2_custom --steps 40 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('TOK_DBG_SKIP=$1 $2', d['ms_per_step'], 'ms/step')"; done
  echo "# HRNet-W48 B=24";
  for v in "0 base" "1 BatchNorm_finalizes" "2 wgrad_reduce_flat" "8 BatchNorm_apply_passes" "16 every_weight_gradient"; do set -- $v
    TOK_DBG_SKIP=$1 python bench.py --backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('TOK_DBG_SKIP=$1 $2', d['ms_per_step'], 'ms/step')"; done
  TOK_BRANCH_STREAMS=0 python bench.py --backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('TOK_BRANCH_STREAMS=0 (no branch streams)', d['ms_per_step'], 'ms/step')"
  TOK_STREAM_PRIO=-1 python bench.py --backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('TOK_STREAM_PRIO=-1 (side / branch streams at high priority)', d['ms_per_step'], 'ms/step')"
} > $o/r06_launch_class_ablation.txt 2>&1
# HRNet-W48 scheduling switches of round 6, each turned back to its round-5 behaviour alone (same box, two interleaved rounds)
H="--backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 15 --warmup 4 --no-cpu-baseline --no-secondary"
{ echo "# HRNet-W48 512x1024 B=24, ms/step, same box, two interleaved rounds; each line turns ONE round-6 scheduling change back";
  for rep in 1 2; do
  for e in "X=0(round-6_defaults)" "TOK_HRNET_BRANCH0_FIRST=1(branch_0_enqueued_first:_forks_recorded_behind_it)" "TOK_HRNET_FUSE_STREAMS=0(fuse_rows_on_the_main_stream)" \
           "TOK_HRNET_FUSE_STREAMS=1(fuse_rows_on_streams,_their_backward_on_main)" "TOK_HRNET_FUSE_STREAMS=2(fuse_backward_on_the_row_streams)" \
           "TOK_WGRAD_SIDE_TAGS=0(main_stream_forks_its_weight_gradients_to_a_fifth_stream)" "TOK_WGRAD_SIDE_TAGS=0,1,2,3(every_stream_forks:_round_5)" \
           "TOK_WGRAD_TAPS_WGS=192(192-way_split_of_the_3x3_weight_gradients)" "TOK_BRANCH_STREAMS=0(no_branch_streams)"; do
    env ${e%%(*} python bench.py $H 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$e', d['ms_per_step'], 'ms/step', d['roofline']['frac'])"
  done; done
} > $o/r06_hrnet_scheduling_ab.txt 2>&1
ls $o
# HRNet segmentation neck alone, commuted vs direct order (tools/ubench/neck_time.py) + kernel stats of both
{ echo "# HRNet-W48 segmentation neck alone at 512x1024 B=24 (tools/ubench/neck_time.py): forward + backward, commuted order (engine/neck.py) vs direct order";
  python tools/ubench/neck_time.py 2>/dev/null
  for m in commuted direct; do
    cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
    timeout 300 rocprofv3 --kernel-trace --stats -d $o/raw/neck_$m -o kt -- python tools/ubench/neck_time.py --only $m --iters 5 > $o/raw/neck_$m.log 2>&1
    db=$(ls $o/raw/neck_$m/*results.db | head -1)
    echo "## rocprofv3 --kernel-trace --stats, $m order (5 timed + 5 warm-up / check calls)"; python tools/prof_summary.py $db 10 | head -16; rm -rf $o/raw/neck_$m
  done
  echo "## HRNet-W48 step, same box, two interleaved rounds"
  for rep in 1 2; do for e in "X=0(commuted)" "TOK_NECK_COMMUTE=0(direct)"; do
    env ${e%%(*} python bench.py $H 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$e', d['ms_per_step'], 'ms/step', d['roofline']['frac'])"
  done; done
} > $o/r06_hrnet_neck_commuted_ab.txt 2>&1
ls $o

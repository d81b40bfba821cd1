#!/bin/bash
cd $GRAFT_REPO_ROOT; o=gpurun_out/r05; mkdir -p $o
bash tools/profile_workload.sh r05_resnet50_bs256 $o --steps 50 --warmup 10 > $o/p1.log 2>&1
python bench.py > $o/r05_bench_default_line.json 2> $o/bench_default.err
{ echo "# the stem kernels of round 5 (csrc/stem.hip), isolated, tools/bench_conv.py (ResNet-50 B=256: 224 x 224 x 4 -> 112 x 112 x 64): us per call, algorithmic GB/s, TFLOP/s";
  echo "# columns: forward (+ BatchNorm partial sums) | weight gradient (incl. the fixed-order fold of its 512 slabs)";
  for v in 1 0; do echo "TOK_STEM_WIN=$v"; TOK_STEM_WIN=$v python tools/bench_conv.py --what fwd,wgrad 2>&1 | grep "(224, 224"; done
  echo "# ResNet-50 B=256 step, same box, ms/step (bench.py --steps 60 --warmup 15): both stem kernels / forward only / neither";
  for e in "A=1" "TOK_STEM_WGRAD=0" "TOK_STEM_WIN=0"; do env $e python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$e', d['ms_per_step'], 'ms/step', d['roofline']['frac'])"; done
} > $o/r05_stem_window_ab.txt 2>&1
cat $o/r05_stem_window_ab.txt
python -c "
import json
d=json.load(open('$o/r05_bench_default_line.json')); print('default line', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
d=json.load(open('$o/r05_resnet50_bs256_bench.json')); print('profiled', d['ms_per_step'], d['roofline']['frac'])
"

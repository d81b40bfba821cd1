#!/bin/bash
# SwinV2-T: GPU tests of the transformer path, bench lines, per-dispatch dump of one step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/s7; mkdir -p $o
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_swin.py tests/test_davit.py tests/test_real_geometry_gpu.py tests/test_golden_gpu.py tests/test_units_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -q -k "swin or davit or Swin or transformer or window_attention or hrnet" > $o/tests.txt 2>&1; tail -5 $o/tests.txt
python tools/ubench/swin_repro.py 256 2>&1 | grep "vs" 
python bench.py --backbone swinv2_custom --steps 30 --warmup 10 --no-cpu-baseline --no-secondary > $o/swin_bench.json 2> $o/swin_bench.err; cut -c1-330 $o/swin_bench.json
python bench.py --backbone davit_t --steps 30 --warmup 10 --no-cpu-baseline --no-secondary > $o/davit_bench.json 2> $o/davit_bench.err; cut -c1-330 $o/davit_bench.json
python bench.py --backbone hrnet_w48 --res 512 --width 1024 --batch 24 --classes 19 --steps 10 --warmup 4 --no-cpu-baseline --no-secondary > $o/hrnet_bench.json 2> $o/hrnet_bench.err; cut -c1-330 $o/hrnet_bench.json

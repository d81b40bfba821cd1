#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "window_attention" 2>&1 | tail -2
for i in 1 2; do python bench.py --backbone swinv2_custom --steps 30 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-200; done
python bench.py --backbone davit_t --steps 30 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-200
o=gpurun_out/a2; mkdir -p $o
timeout 300 rocprofv3 --kernel-trace --stats -d $o/raw -o kt -- python bench.py --backbone swinv2_custom --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $o/kt.log 2>&1
db=$(ls $o/raw/*results.db | head -1); python tools/prof_summary.py $db 9 | grep -i "attn\|ms/step"; rm -rf $o/raw
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $o/raw -o p -- python bench.py --backbone swinv2_custom --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $o/pf.log 2>&1
python tools/pmc_kernel.py $(ls $o/raw/*results.db | head -1) attn_fwd; python tools/pmc_kernel.py $(ls $o/raw/*results.db | head -1) attn_bwd; rm -rf $o/raw

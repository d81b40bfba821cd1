#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r05davit; mkdir -p $o
timeout 400 rocprofv3 --kernel-trace --stats -d $o/kt -o kt -- python bench.py --backbone davit_t --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $o/kt.log 2>&1
db=$(ls $o/kt/*results.db | head -1)
{ echo "# rocprofv3 --kernel-trace --stats of bench.py --backbone davit_t --steps 6 --warmup 3; tools/prof_summary.py"; python tools/prof_summary.py $db 9; } > $o/r05_davit_t_224_bs256_kernel_stats.txt
python tools/timeline.py $db > $o/r05_davit_t_224_bs256_timeline.txt 2>&1
rm -rf $o/kt
head -30 $o/r05_davit_t_224_bs256_kernel_stats.txt | cut -c1-170

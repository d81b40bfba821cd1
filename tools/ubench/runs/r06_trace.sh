#!/bin/bash
# kernel trace of one workload -> gpurun_out/r06_trace/<tag>_{kernel_stats,timeline,gaps}.txt :  r06_trace.sh <tag> [bench.py args...]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
O=gpurun_out/r06_trace; mkdir -p $O/raw
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $O/raw/${tag}_kt -o kt -- python bench.py "$@" --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $O/raw/${tag}_kt.log 2>&1
db=$(ls $O/raw/${tag}_kt/*results.db 2>/dev/null | head -1)
if [ -n "$db" ]; then
  { echo "# rocprofv3 --kernel-trace --stats of bench.py $* --steps 6 --warmup 3; tools/prof_summary.py"; python tools/prof_summary.py $db 9; } > $O/${tag}_kernel_stats.txt
  { echo "# one training step out of the same trace; tools/timeline.py"; python tools/timeline.py $db; } > $O/${tag}_timeline.txt 2>&1
  { echo "# where the main queue idles in that step; tools/gaps.py"; python tools/gaps.py $db --min-us 8 --top 30; } > $O/${tag}_gaps.txt 2>&1
fi
rm -rf $O/raw/${tag}_kt
head -30 $O/${tag}_timeline.txt

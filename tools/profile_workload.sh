#!/bin/bash
# Round artifacts of one workload on the GPU box (every rocprofv3 run under its own timeout; PMC passes separate, --pmc only):
#   tools/profile_workload.sh <tag> <outdir> [bench.py args...]
# writes <outdir>/<tag>_{bench.json,kernel_stats.txt,timeline.txt,pmc_traffic.{txt,json},pmc_mfma.{txt,json}}
tag=$1; out=$2; shift 2
mkdir -p $out/raw
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py "$@" > $out/${tag}_bench.json 2> $out/raw/${tag}_bench.err
timeout 400 rocprofv3 --kernel-trace --stats -d $out/raw/${tag}_kt -o kt -- python bench.py "$@" --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $out/raw/${tag}_kt.log 2>&1
db=$(ls $out/raw/${tag}_kt/*results.db 2>/dev/null | head -1)
if [ -n "$db" ]; then
  { echo "# rocprofv3 --kernel-trace --stats of bench.py $* --steps 6 --warmup 3; tools/prof_summary.py"; python tools/prof_summary.py $db 9; } > $out/${tag}_kernel_stats.txt
  { echo "# one training step out of the same trace; tools/timeline.py"; python tools/timeline.py $db; } > $out/${tag}_timeline.txt 2>&1
  { echo "# where the main queue idles in that step; tools/gaps.py"; python tools/gaps.py $db --min-us 8 --top 30; } > $out/${tag}_gaps.txt 2>&1
fi
timeout 500 rocprofv3 --pmc FETCH_SIZE -d $out/raw/${tag}_pf -o p -- python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $out/raw/${tag}_pf.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE -d $out/raw/${tag}_pw -o p -- python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $out/raw/${tag}_pw.log 2>&1
f=$(ls $out/raw/${tag}_pf/*results.db 2>/dev/null | head -1); w=$(ls $out/raw/${tag}_pw/*results.db 2>/dev/null | head -1)
if [ -n "$f" ] && [ -n "$w" ]; then
  ae=$(python -c "import json,sys; print(json.load(open(sys.argv[1]))['config'].get('arena_elements', 0))" $out/${tag}_bench.json 2>/dev/null || echo 0)
  { echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py $* --steps 2 --warmup 1; tools/pmc_traffic.py (calibration: optimizer kernel over $ae fp32 arena elements)"; python tools/pmc_traffic.py $f $w 3 $out/${tag}_pmc_traffic.json $ae; } > $out/${tag}_pmc_traffic.txt 2>&1
fi
timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/raw/${tag}_pm -o p -- python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $out/raw/${tag}_pm.log 2>&1
m=$(ls $out/raw/${tag}_pm/*results.db 2>/dev/null | head -1)
if [ -n "$m" ]; then
  { echo "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of bench.py $* --steps 2 --warmup 1; tools/pmc_mfma.py"; python tools/pmc_mfma.py $m 3 $out/${tag}_pmc_mfma.json; } > $out/${tag}_pmc_mfma.txt 2>&1
fi
rm -rf $out/raw/${tag}_kt $out/raw/${tag}_pf $out/raw/${tag}_pw $out/raw/${tag}_pm
ls -la $out | head -20

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_run7; mkdir -p $o/raw
HR="--backbone hrnet_w48 --res 512 --width 1024 --classes 19 --batch 24 --steps 3 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $o/raw/hr -o kt -- python bench.py $HR > $o/raw/hr.log 2>&1
db=$(ls $o/raw/hr/*results.db 2>/dev/null | head -1)
python tools/prof_summary.py $db 5 > $o/hr_stats.txt 2>&1
grep -i "upce\|ce_\|bilinear" $o/hr_stats.txt
rm -rf $o/raw/hr

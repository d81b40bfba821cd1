cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_run2; mkdir -p $o/raw
timeout 1200 python -m pytest tests/test_swin.py tests/test_units_real_gpu.py -m gpu -x -q -s > $o/t.txt 2>&1; grep -n "passed\|failed" $o/t.txt; grep "real unit" $o/t.txt | sort -t'H' -k2 | awk '{print}' | head -150 > $o/real_units.txt; tail -3 $o/t.txt
timeout 600 rocprofv3 --kernel-trace -d $o/raw/hr -o kt -- python bench.py --backbone hrnet_w48 --res 512 --width 1024 --classes 19 --batch 24 --steps 4 --warmup 3 --no-cpu-baseline --no-secondary > $o/raw/hr.log 2>&1
db=$(ls $o/raw/hr/*results.db 2>/dev/null | head -1)
python tools/gaps.py $db --min-us 20 --top 60 > $o/hr_gaps.txt 2>&1
python tools/timeline.py $db > $o/hr_timeline.txt 2>&1
rm -rf $o/raw/hr
head -50 $o/hr_gaps.txt

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_run8; mkdir -p $o/raw
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "upsample_ce" -x -q > $o/k.txt 2>&1; tail -3 $o/k.txt
TOK_UPCE_TILED=0 timeout 900 python -m pytest tests/test_kernels_gpu.py -k "upsample_ce" -x -q > $o/k0.txt 2>&1; tail -1 $o/k0.txt
timeout 900 python -m pytest tests/test_hrnet.py -m gpu -x -q > $o/h.txt 2>&1; tail -1 $o/h.txt
HR="--backbone hrnet_w48 --res 512 --width 1024 --classes 19 --batch 24 --steps 3 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $o/raw/hr -o kt -- python bench.py $HR > $o/raw/hr.log 2>&1
db=$(ls $o/raw/hr/*results.db 2>/dev/null | head -1)
python tools/prof_summary.py $db 5 > $o/hr_stats.txt 2>&1
grep -i "upce" $o/hr_stats.txt
rm -rf $o/raw/hr
HR="--backbone hrnet_w48 --res 512 --width 1024 --classes 19 --batch 24 --steps 20 --warmup 5 --no-cpu-baseline"
python bench.py $HR > $o/b_fused.json 2>$o/b_fused.err; cut -c1-230 $o/b_fused.json
TOK_FUSE_UPSAMPLE_CE=0 python bench.py $HR > $o/b_unfused.json 2>$o/b_unfused.err; cut -c1-230 $o/b_unfused.json

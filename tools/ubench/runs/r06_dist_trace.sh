cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/dist_trace; mkdir -p $o
export TOK_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout 300 rocprofv3 --kernel-trace --stats -d $o/raw -o kt -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $o/log.txt 2>&1
db=$(ls $o/raw/*results.db | head -1)
python tools/prof_summary.py $db 9 > $o/stats.txt
python tools/timeline.py $db > $o/timeline.txt 2>&1
rm -rf $o/raw

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/ubench/neck_time.py
o=gpurun_out/neck; mkdir -p $o
for m in commuted direct; do
timeout 300 rocprofv3 --kernel-trace --stats -d $o/raw_$m -o kt -- python tools/ubench/neck_time.py --only $m --iters 5 > $o/$m.log 2>&1
db=$(ls $o/raw_$m/*results.db | head -1)
python tools/prof_summary.py $db 10 > $o/${m}_stats.txt; rm -rf $o/raw_$m
done

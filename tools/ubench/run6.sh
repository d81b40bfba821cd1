cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_run9; mkdir -p $o
HR="--backbone hrnet_w48 --res 512 --width 1024 --classes 19 --batch 24 --steps 20 --warmup 5 --no-cpu-baseline"
for m in "" "0,1,2,2" "0,1,1,2" "0,1,2,1" "0,1,1,1"; do
  TOK_BRANCH_MAP=$m python bench.py $HR 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('TOK_BRANCH_MAP=$m', d['ms_per_step'], 'ms/step')"
done | tee $o/branch_map.txt

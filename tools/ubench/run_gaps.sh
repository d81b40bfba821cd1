cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_gaps; mkdir -p $o/raw
for wl in "swin --backbone swinv2_custom" "r50 " ; do
  set -- $wl; tag=$1; shift
  timeout 400 rocprofv3 --kernel-trace -d $o/raw/$tag -o kt -- python bench.py "$@" --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $o/raw/$tag.log 2>&1
  db=$(ls $o/raw/$tag/*results.db 2>/dev/null | head -1)
  python tools/gaps.py $db --min-us 8 --top 45 > $o/${tag}_gaps.txt 2>&1
  python tools/timeline.py $db > $o/${tag}_timeline.txt 2>&1
  rm -rf $o/raw/$tag
done
head -60 $o/swin_gaps.txt

cd $GRAFT_REPO_ROOT
for cfg in "0 200" "1 200" "1 128"; do set -- $cfg
  for wl in "" "--backbone hrnet_w48 --res 512 --width 1024 --classes 19 --batch 24"; do
    TOK_CONV_WIN=$1 TOK_CONV_WIN_MIN_TILES=$2 python bench.py $wl --steps 12 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('win=$1 min_tiles=$2', d['metric'][18:40], d['ms_per_step'], d['final_loss'])"
  done
done

cd $GRAFT_REPO_ROOT
echo "== correctness (kernel tests, conv)"
python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" 2>&1 | tail -3
echo "== benches: (win, ring) = (0,0) (1,0) (1,1)"
for cfg in "0 0" "1 0" "1 1"; do set -- $cfg
  for wl in "" "--backbone hrnet_w48 --res 512 --width 1024 --classes 19 --batch 24" "--backbone swinv2_custom"; do
    TOK_CONV_WIN=$1 TOK_CONV_RING=$2 python bench.py $wl --steps 12 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('win=$1 ring=$2', d['metric'][18:40], d['ms_per_step'], d['final_loss'])"
  done
done

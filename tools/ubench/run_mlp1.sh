cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_mlp1; mkdir -p $o
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "mlp" -x -q > $o/k.txt 2>&1; tail -15 $o/k.txt
timeout 300 python tools/ubench/mlp_dw_time.py > $o/time.txt 2>&1; cat $o/time.txt | tail -8
timeout 900 python -m pytest tests/test_swin.py tests/test_davit.py tests/test_real_geometry_gpu.py -m gpu -x -q > $o/swin.txt 2>&1; tail -8 $o/swin.txt
python bench.py --backbone swinv2_custom --steps 20 --warmup 5 --no-cpu-baseline > $o/b_new.json 2>$o/b_new.err; cut -c1-400 $o/b_new.json
TOK_MLP_RECOMPUTE=0 python bench.py --backbone swinv2_custom --steps 20 --warmup 5 --no-cpu-baseline > $o/b_old.json 2>$o/b_old.err; cut -c1-400 $o/b_old.json

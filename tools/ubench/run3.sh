cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_run6; mkdir -p $o
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "upsample_ce" -x -q > $o/k.txt 2>&1; tail -12 $o/k.txt
timeout 900 python -m pytest tests/test_hrnet.py tests/test_real_geometry_gpu.py tests/test_dice.py tests/test_ocr.py -m gpu -x -q > $o/h.txt 2>&1; tail -6 $o/h.txt
HR="--backbone hrnet_w48 --res 512 --width 1024 --classes 19 --batch 24 --steps 20 --warmup 5 --no-cpu-baseline"
python bench.py $HR > $o/b_fused.json 2>$o/b_fused.err; cut -c1-330 $o/b_fused.json
TOK_FUSE_UPSAMPLE_CE=0 python bench.py $HR > $o/b_unfused.json 2>$o/b_unfused.err; cut -c1-330 $o/b_unfused.json
python bench.py $HR > $o/b_fused2.json 2>$o/b_fused2.err; cut -c1-330 $o/b_fused2.json

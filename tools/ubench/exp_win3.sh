cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" 2>&1 | tail -3
for r in 0 1; do echo "-- TOK_CONV_WIN=$r resnet"; TOK_CONV_WIN=$r python tools/bench_conv.py --what fwd,dgrad 2>&1 | grep ", 3, 1)\|^fwd\|^dgrad"; done
for r in 0 1; do echo "-- TOK_CONV_WIN=$r hrnet B=24"; TOK_CONV_WIN=$r python tools/bench_conv.py --net hrnet_w48 --batch 24 --what fwd,dgrad 2>&1 | grep ", 3, 1)\|^fwd\|^dgrad"; done

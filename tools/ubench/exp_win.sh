cd $GRAFT_REPO_ROOT
echo "== correctness"
python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" 2>&1 | tail -6
echo "== per-layer: win off / on (ring off in both)"
for r in 0 1; do echo "-- TOK_CONV_WIN=$r"; TOK_CONV_RING=0 TOK_CONV_WIN=$r python tools/bench_conv.py --what fwd,dgrad 2>&1 | grep ", 3, 1)\|^fwd\|^dgrad"; done

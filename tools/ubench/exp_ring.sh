cd $GRAFT_REPO_ROOT
echo "== correctness: every eligible conv shape of the kernel tests on the ring"
TOK_CONV_RING_MIN_TILES=1 TOK_CONV_RING_MIN_K=32 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv or linear or dgrad" 2>&1 | tail -4
echo "== per-layer: ring off / on"
for r in 0 1; do echo "-- TOK_CONV_RING=$r resnet50"; TOK_CONV_RING=$r python tools/bench_conv.py --what fwd,dgrad 2>&1 | tail -28; done
for r in 0 1; do echo "-- TOK_CONV_RING=$r swin"; TOK_CONV_RING=$r TOK_CONV_RING_MIN_K=64 python tools/bench_conv.py --net swinv2t --what fwd,dgrad 2>&1 | tail -22; done

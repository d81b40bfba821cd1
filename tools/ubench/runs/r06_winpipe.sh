#!/bin/bash
# round 6: conv_win with the software-pipelined stage (TOK_CONV_WIN_PIPE=1) vs the round-5 stage (=0): isolated layers, kernel tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_winpipe; mkdir -p $O
for v in 0 1; do
  TOK_CONV_WIN_PIPE=$v timeout 600 python tools/bench_conv.py --what fwd,dgrad --net resnet50 > $O/resnet50_pipe$v.txt 2>&1
  TOK_CONV_WIN_PIPE=$v timeout 600 python tools/bench_conv.py --what fwd,dgrad --net hrnet_w48 --batch 24 > $O/hrnet_pipe$v.txt 2>&1
  echo "== TOK_CONV_WIN_PIPE=$v"; grep -h ', 3, 1)' $O/resnet50_pipe$v.txt $O/hrnet_pipe$v.txt
done
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_fwd or conv_dgrad or colsum" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt

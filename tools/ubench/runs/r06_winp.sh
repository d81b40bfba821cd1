#!/bin/bash
# round 6: the software-pipelined window weight-gradient kernel vs the round-5 one (TOK_WGRAD_WINP=0)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_winp; mkdir -p $O
TOK_WGRAD_WINP=0 timeout 600 python tools/ubench/winp_check.py --save $O/a.pt > $O/check_a.txt 2>&1
TOK_WGRAD_WINP=1 timeout 600 python tools/ubench/winp_check.py --save $O/b.pt --cmp $O/a.pt > $O/check_b.txt 2>&1
tail -3 $O/check_b.txt
rm -f $O/a.pt $O/b.pt
for v in 0 1; do
  TOK_WGRAD_WINP=$v timeout 600 python tools/bench_conv.py --what wgrad --net resnet50 > $O/resnet50_winp$v.txt 2>&1
  TOK_WGRAD_WINP=$v timeout 600 python tools/bench_conv.py --what wgrad --net hrnet_w48 --batch 24 > $O/hrnet_winp$v.txt 2>&1
done
grep -h ', 3, 1)' $O/resnet50_winp0.txt $O/resnet50_winp1.txt
cat $O/hrnet_winp0.txt $O/hrnet_winp1.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt

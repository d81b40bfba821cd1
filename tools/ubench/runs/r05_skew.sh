#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for sk in "0,0" "1,2" "1,4" "1,8" "2,2" "2,4" "2,8"; do
  echo "== TOK_WIN_SKEW=$sk"; TOK_WIN_SKEW=$sk python tools/bench_conv.py --what fwd 2>&1 | grep ", 3, 1)"
  TOK_WIN_SKEW=$sk python tools/bench_conv.py --what fwd --net hrnet_w48 --batch 24 2>&1 | grep "fwd:"
done

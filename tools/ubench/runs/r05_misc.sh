#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05misc; mkdir -p $O
timeout 1500 python -m pytest tests/test_default_dispatch_gpu.py -x -q -m gpu -s > $O/default_dispatch.txt 2>&1; tail -5 $O/default_dispatch.txt
grep "default dispatch\]" $O/default_dispatch.txt | head
timeout 1500 python tools/ubench/host_contention.py > $O/host_contention.txt 2>&1; cat $O/host_contention.txt | tail -12

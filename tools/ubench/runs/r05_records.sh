#!/bin/bash
# record runs of the tree as it stands: the whole GPU suite (+ parity distances), then the profile set
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/parity_distances.jsonl
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r05_gpu_tests.txt 2>&1
tail -5 gpurun_out/r05_gpu_tests.txt
bash tools/ubench/r05_profiles.sh > gpurun_out/r05_profiles.log 2>&1
tail -3 gpurun_out/r05_profiles.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/parity_distances.jsonl
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r05_gpu_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r05_gpu_tests.txt | tail -2
python bench.py > gpurun_out/r05_bench_default_line.json 2> gpurun_out/bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_default_line.json')); print('default', d['ms_per_step'], d['value'], d['roofline']['frac'])"

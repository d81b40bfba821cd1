#!/bin/bash
# triage of the stress-run failures: each arm under its own short timeout
mkdir -p gpurun_out/triage
cd /root/repo
( timeout 900 python tools/ubench/g256_check.py ) > gpurun_out/triage/g256_check.txt 2>&1; echo "g256_check rc=$?"
( TOK_GEMM256=3 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "fused_bn_finalize" -x ) > gpurun_out/triage/fin_g3.txt 2>&1; echo "fin_g3 rc=$?"
( TOK_GEMM256=3 timeout 900 python -m pytest tests/test_fullsize_properties_gpu.py -q -k "bit_reproducible or c5_step" ) > gpurun_out/triage/full_g3.txt 2>&1; echo "full_g3 rc=$?"
( TOK_WGRAD_256=2 timeout 900 python -m pytest tests/test_fullsize_properties_gpu.py -q -k "bit_reproducible or c5_step" ) > gpurun_out/triage/full_w2.txt 2>&1; echo "full_w2 rc=$?"
tail -3 gpurun_out/triage/*.txt

#!/bin/bash
# whole GPU suite with the 256-tile kernels forced onto every layer they can run; per-test and overall timeouts
mkdir -p gpurun_out/stress
cd /root/repo
( TOK_GEMM256=3 TOK_WGRAD_256=2 timeout 1500 python -m pytest tests -m gpu -q --timeout=240 \
    --deselect tests/test_kernels_gpu.py::test_gemm256_tile_kernel_is_bit_identical_to_the_default_kernels \
    --deselect tests/test_kernels_gpu.py::test_wgrad_256_tile_kernel_matches_the_128_tile_plan ) > gpurun_out/stress/suite.txt 2>&1
echo "stress rc=$?"
tail -n 15 gpurun_out/stress/suite.txt

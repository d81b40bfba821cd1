#!/bin/bash
# stride-2 data gradient on the shared dY window: parity tests, then the A/B on the network shapes
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "stride2_on_the_shared or test_conv_dgrad" -x 2>&1 | tail -15 > gpurun_out/s2d_tests.txt
cat gpurun_out/s2d_tests.txt
(TOK_CONV_S2D=0 timeout 300 python tools/ubench/s2d_ab.py; TOK_CONV_S2D=1 timeout 300 python tools/ubench/s2d_ab.py) > gpurun_out/s2d_ab.txt 2>&1
cat gpurun_out/s2d_ab.txt

#!/usr/bin/env python
"""Isolated time of the pooled-stem forward (ResNet-50 B=256: 112 x 112 x 64) and of DaViT-T's depthwise 3x3 (56 x 56 x 96, B=256):
    python tools/ubench/pool_dw_time.py [--lib path]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from torchok_amd import _C  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=None)
    a = ap.parse_args()
    lib = _C.load_library(a.lib)
    st = torch.cuda.current_stream().cuda_stream
    n, h, w, c = 256, 112, 112, 64
    y = torch.randn(n, h, w, c, device='cuda').to(torch.bfloat16)
    sc, sh = torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda') * 0.1
    pooled = torch.empty(n, 56, 56, c, device='cuda', dtype=torch.bfloat16)
    ypool = torch.empty_like(pooled)
    arg = torch.empty(n, 56, 56, c, device='cuda', dtype=torch.uint8)
    us = timeit(lambda: lib.tok_bn_relu_maxpool_fwd(y.data_ptr(), sc.data_ptr(), sh.data_ptr(), n, h, w, c, pooled.data_ptr(),
                                                    arg.data_ptr(), ypool.data_ptr(), st), iters=20, warm=3)
    by = y.numel() * 2 + pooled.numel() * 5
    print(f'bn_relu_maxpool_fwd {us:7.1f} us  {by / us / 1e3:6.0f} GB/s')
    n, h, w, c = 256, 56, 56, 96
    x = torch.randn(n, h, w, c, device='cuda').to(torch.bfloat16)
    wt, b = torch.randn(c, 9, device='cuda') * 0.1, torch.randn(c, device='cuda') * 0.1
    out = torch.empty_like(x)
    for flip in (0, 1):
        us = timeit(lambda: lib.tok_dwconv3x3(x.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), 0, flip, n, h, w, c, c, st),
                    iters=20, warm=3)
        print(f'dwconv3x3 flip={flip} {us:7.1f} us  {x.numel() * 4 / us / 1e3:6.0f} GB/s')


if __name__ == '__main__':
    main()

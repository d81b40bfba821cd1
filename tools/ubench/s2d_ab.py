#!/usr/bin/env python
"""Stride-2 3x3 data gradients of ResNet-50 (B=256) and HRNet-W48 (B=24, 512x1024): run once per TOK_CONV_S2D setting
(the switch is read once per process):  TOK_CONV_S2D=0 python tools/ubench/s2d_ab.py ; TOK_CONV_S2D=1 python ..."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from torchok_amd import _C  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402

SHAPES = [('r50.l2', 256, 56, 56, 128, 128), ('r50.l3', 256, 28, 28, 256, 256), ('r50.l4', 256, 14, 14, 512, 512),
          ('hr.48-48', 24, 128, 256, 48, 48), ('hr.48-96', 24, 128, 256, 48, 96), ('hr.96-96', 24, 64, 128, 96, 96),
          ('hr.96-192', 24, 64, 128, 96, 192), ('hr.192-192', 24, 32, 64, 192, 192), ('hr.192-384', 24, 32, 64, 192, 384)]


def main():
    lib = _C.load_library(os.environ.get('TOK_LIB'))
    st = torch.cuda.current_stream().cuda_stream
    print(f'TOK_CONV_S2D={os.environ.get("TOK_CONV_S2D", "(default)")}')
    for mode in ('plain', 'bnstats'):
        for name, n, h, w, c, k in SHAPES:
            p, q = h // 2, w // 2
            d = _C.ConvDesc(n, h, w, c, k, 3, 3, p, q, 2, 1, 3)
            dy = torch.randn(n, p, q, k, device='cuda').to(torch.bfloat16)
            wd = (torch.randn(c, 3, 3, k, device='cuda') * 0.05).to(torch.bfloat16)
            dx = torch.empty(n, h, w, c, device='cuda', dtype=torch.bfloat16)
            bn_y = torch.randn(n, h, w, c, device='cuda').to(torch.bfloat16)
            mask = torch.randint(0, 256, (n * h * w, c // 8), dtype=torch.uint8, device='cuda')
            rows = lib.tok_conv_dgrad_stat_rows(ctypes.byref(d))
            part = torch.empty(2, rows, c, device='cuda')
            if mode == 'plain':
                fn = lambda: lib.tok_conv_dgrad(ctypes.byref(d), dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), 0, st)  # noqa
            else:
                fn = lambda: lib.tok_conv_dgrad_bnstats(ctypes.byref(d), dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), 0,  # noqa
                                                        bn_y.data_ptr(), mask.data_ptr(), part.data_ptr(), st)
            us = timeit(fn, iters=10, warm=3)
            flops = 2.0 * n * h * w * c * k * 9 / 4
            print(f'{mode:8s} {name:12s} rows {rows:4d}  {us:8.1f} us  {flops / us / 1e6:6.0f} TF/s')


if __name__ == '__main__':
    main()

"""Isolated time of the pointwise data-gradient launches a ResNet-50 step makes on its 56 / 28 px maps, per epilogue mode, ring
kernel vs stream kernel (TOK_PW_STREAM).   python tools/ubench/pw_modes.py <tag>"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchok_amd import _C  # noqa: E402

# (n, h, w, c = channels of dx, k = channels of dy)
SHAPES = [(256, 56, 56, 256, 64), (256, 28, 28, 512, 128), (256, 56, 56, 256, 128)]


def timeit(f, n=20):
    for _ in range(3):
        assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main(tag):
    lib = _C.load_library()
    st = torch.cuda.current_stream().cuda_stream
    BF = torch.bfloat16
    P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    for n, h, w, c, k in SHAPES:
        m = n * h * w
        d = _C.ConvDesc(n, h, w, c, k, 1, 1, h, w, 1, 0, 1)
        dy = torch.randn(m, k, device='cuda').to(BF)
        wd = (torch.randn(c, k, device='cuda') * k ** -0.5).to(BF)
        dx = torch.randn(m, c, device='cuda').to(BF)
        bn_y = torch.randn(m, c, device='cuda').to(BF)
        mask = torch.randint(0, 256, (m, c // 8), dtype=torch.uint8, device='cuda')
        rows = lib.tok_conv_dgrad_stat_rows(ctypes.byref(d))
        part = torch.zeros(2, rows, c, device='cuda')
        t_plain = timeit(lambda: lib.tok_conv_dgrad(ctypes.byref(d), P(dy), P(wd), P(dx), 0, st))
        t_acc = timeit(lambda: lib.tok_conv_dgrad(ctypes.byref(d), P(dy), P(wd), P(dx), 1, st))
        t_ms = timeit(lambda: lib.tok_conv_dgrad_maskstore(ctypes.byref(d), P(dy), P(wd), P(dx), 1, P(mask), P(part), st))
        t_bn = timeit(lambda: lib.tok_conv_dgrad_bnstats(ctypes.byref(d), P(dy), P(wd), P(dx), 1, P(bn_y), P(mask), P(part), st))
        mb = lambda *t: sum(x.numel() * x.element_size() for x in t) / 1e6  # noqa: E731
        print(f'[{tag}] M={m} {k}->{c}: plain {t_plain:6.1f} ({mb(dy, dx):.0f} MB)  acc {t_acc:6.1f} ({mb(dy, dx, dx):.0f} MB)  '
              f'acc+maskstore {t_ms:6.1f} ({mb(dy, dx, dx, mask):.0f} MB)  acc+bnstats {t_bn:6.1f} ({mb(dy, dx, dx, bn_y, mask):.0f} MB)',
              flush=True)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '')

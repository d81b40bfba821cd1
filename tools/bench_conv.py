#!/usr/bin/env python
"""Per-layer micro-benchmark of the conv kernels on the ResNet-50 shapes (B=256, 224x224):
    python tools/bench_conv.py [--lib path/to/libtok_gfx950.so] [--what fwd,dgrad,wgrad,bn] [--batch 256]
Prints per distinct shape: avg us, algorithmic GB/s (|X|+|Y| resp. operands), TFLOP/s, and per-step totals.
Used to iterate on kernel variants; numbers quoted in DESIGN.md / profiles come from here and rocprofv3."""
import argparse
import ctypes
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from torchok_amd import _C  # noqa: E402

BF16 = torch.bfloat16


def resnet50_convs(B):
    out = []

    def conv(name, h, cin, cout, r, stride):
        pad = (r - 1) // 2
        out.append(dict(name=name, n=B, h=h, w=h, c=cin, k=cout, r=r, stride=stride, pad=pad))
        return (h + 2 * pad - r) // stride + 1
    h = conv('stem', 224, 4, 64, 7, 2)
    h = 56
    inp = 64
    for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
        for b in range(blocks):
            s = stride if b == 0 else 1
            conv(f'l{li + 1}.{b}.c1', h, inp, planes, 1, 1)
            h2 = conv(f'l{li + 1}.{b}.c2', h, planes, planes, 3, s)
            conv(f'l{li + 1}.{b}.c3', h2, planes, planes * 4, 1, 1)
            if b == 0:
                conv(f'l{li + 1}.{b}.ds', h, inp, planes * 4, 1, s)
            h, inp = h2, planes * 4
    return out


def swinv2t_linears(B):
    """The token GEMMs of SwinV2-T at 224 (window 7): 1x1 'convs' over (B, H, W, C) token maps."""
    out = []
    for res, c, blocks in ((56, 96, 2), (28, 192, 2), (14, 384, 6), (7, 768, 2)):
        for _ in range(blocks):
            for name, cin, cout in (('qkv', c, 3 * c), ('proj', c, c), ('fc1', c, 4 * c), ('fc2', 4 * c, c)):
                out.append(dict(name=name, n=B, h=res, w=res, c=cin, k=cout, r=1, stride=1, pad=0))
        if res > 7:
            out.append(dict(name='merge', n=B, h=res // 2, w=res // 2, c=4 * c, k=2 * c, r=1, stride=1, pad=0))
    return out


def hrnet_w48_convs(B):
    """The 3x3 / stride-1 convolutions of HRNet-W48's four branches at 512x1024 input (BasicBlocks: 2 per block, 4 blocks per
    module; modules 1 / 4 / 3 for stages 2 / 3 / 4) — the layers the shared-window kernel serves."""
    out = []
    for (h, w, c), cnt in (((128, 256, 48), 64), ((64, 128, 96), 64), ((32, 64, 192), 56), ((16, 32, 384), 24)):
        for _ in range(cnt):
            out.append(dict(name=f'b{c}', n=B, h=h, w=w, c=c, k=c, r=3, stride=1, pad=1))
    return out


def timeit(fn, iters=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=None)
    ap.add_argument('--what', default='fwd,dgrad,wgrad')
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--net', default='resnet50', choices=['resnet50', 'swinv2t', 'hrnet_w48'])
    args = ap.parse_args()
    lib = _C.load_library(args.lib)
    what = args.what.split(',')
    st = torch.cuda.current_stream().cuda_stream
    shapes = OrderedDict()
    nets = {'resnet50': resnet50_convs, 'swinv2t': swinv2t_linears, 'hrnet_w48': hrnet_w48_convs}
    for c in nets[args.net](args.batch):
        key = (c['h'], c['w'], c['c'], c['k'], c['r'], c['stride'])
        shapes.setdefault(key, [c, 0])
        shapes[key][1] += 1
    totals = {w: 0.0 for w in what}
    ideal = {w: 0.0 for w in what}
    print(f'{"shape(h,w,c,k,r,s)":>28} {"cnt":>3} ' + ' '.join(f'{w + "_us":>9} {"GB/s":>6} {"TF/s":>6}' for w in what))
    for key, (c, cnt) in shapes.items():
        n, h, w_, cin, k, r, stride, pad = c['n'], c['h'], c['w'], c['c'], c['k'], c['r'], c['stride'], c['pad']
        p = (h + 2 * pad - r) // stride + 1
        q = (w_ + 2 * pad - r) // stride + 1
        s_pad = 8 if cin == 4 else r
        d = _C.ConvDesc(n, h, w_, cin, k, r, r, p, q, stride, pad, s_pad)
        x = torch.randn(n, h, w_, cin, device='cuda').to(BF16)
        y = torch.randn(n, p, q, k, device='cuda').to(BF16)
        wf = (torch.randn(k, r, s_pad, cin, device='cuda') * 0.05).to(BF16)
        wd = (torch.randn(cin, r, r, k, device='cuda') * 0.05).to(BF16) if cin != 4 else None
        rows = lib.tok_conv_fwd_stat_rows(ctypes.byref(d))
        stats = torch.empty(2, rows, k, device='cuda')
        dw = torch.empty(k, r, r, 3 if cin == 4 else cin, device='cuda')
        wsb = lib.tok_conv_wgrad_ws_bytes(ctypes.byref(d))
        ws = torch.empty(max(wsb // 4, 16), device='cuda')
        xb, yb = x.numel() * 2, y.numel() * 2
        flops = 2.0 * n * p * q * k * r * r * (3 if cin == 4 else cin)
        cols = []
        for wname in what:
            if wname == 'fwd':
                fn = lambda: lib.tok_conv_fwd(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), None, y.data_ptr(),  # noqa
                                              stats.data_ptr(), st)
            elif wname == 'dgrad':
                if wd is None:
                    cols.append(f'{"-":>9} {"-":>6} {"-":>6}')
                    continue
                fn = lambda: lib.tok_conv_dgrad(ctypes.byref(d), y.data_ptr(), wd.data_ptr(), x.data_ptr(), 0, st)  # noqa
            else:
                fn = lambda: lib.tok_conv_wgrad(ctypes.byref(d), x.data_ptr(), y.data_ptr(), dw.data_ptr(), k,  # noqa
                                                3 if cin == 4 else cin, ws.data_ptr(), wsb, 0, st)
            us = timeit(fn)
            totals[wname] += us * cnt
            ideal[wname] += max((xb + yb) / 5.5e12, flops / 1.0e15) * 1e6 * cnt
            cols.append(f'{us:9.1f} {(xb + yb) / us / 1e3:6.0f} {flops / us / 1e6:6.0f}')
        print(f'{str(key):>28} {cnt:3d} ' + ' '.join(cols))
    for wname in what:
        print(f'{wname}: {totals[wname] / 1e3:.3f} ms/step   (bound max(5.5 TB/s, 1 PF/s): {ideal[wname] / 1e3:.3f} ms)')


if __name__ == '__main__':
    main()

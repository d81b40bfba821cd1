#!/usr/bin/env python
"""Bit-for-bit A/B of the 3x3 / stride-1 weight-gradient window kernels: run once per library setting with --save, then --cmp.
    TOK_WGRAD_WINP=0 python tools/ubench/winp_check.py --save gpurun_out/winp_a.pt
    TOK_WGRAD_WINP=1 python tools/ubench/winp_check.py --save gpurun_out/winp_b.pt --cmp gpurun_out/winp_a.pt
Cases: the unit-test geometries plus the real ResNet-50 / HRNet-W48 layers at reduced batch."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from torchok_amd import _C  # noqa: E402

CASES = [(2, 16, 16, 64, 64), (1, 7, 7, 512, 512), (40, 4, 4, 64, 64), (2, 30, 26, 96, 48), (3, 9, 33, 48, 48),
         (5, 14, 14, 256, 256), (3, 28, 28, 128, 128), (2, 56, 56, 64, 128), (2, 56, 56, 64, 64), (1, 40, 72, 48, 48),
         (64, 14, 14, 256, 256), (32, 7, 7, 512, 512), (3, 128, 256, 48, 48), (3, 64, 128, 96, 96), (3, 32, 64, 192, 192),
         (6, 16, 32, 384, 384), (7, 5, 3, 64, 64), (33, 1, 9, 64, 64), (256, 14, 14, 256, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--save', required=True)
    ap.add_argument('--cmp', default=None)
    args = ap.parse_args()
    lib = _C.load_library()
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for (n, h, w, c, k) in CASES:
        d = _C.ConvDesc(n, h, w, c, k, 3, 3, h, w, 1, 1, 3)
        g = torch.Generator(device='cuda').manual_seed(n * 131 + h * 17 + c)
        x = torch.randn(n, h, w, c, device='cuda', generator=g).to(torch.bfloat16)
        dy = torch.randn(n, h, w, k, device='cuda', generator=g).to(torch.bfloat16)
        dw = torch.zeros(k, 3, 3, c, device='cuda')
        wsb = lib.tok_conv_wgrad_ws_bytes(ctypes.byref(d))
        ws = torch.full((max(wsb // 4, 16),), float('nan'), device='cuda')
        rc = lib.tok_conv_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), k, c, ws.data_ptr(), wsb, 0, st)
        assert rc == 0, lib.tok_last_error()
        torch.cuda.synchronize()
        # fp32 restatement on the device (conv weight gradient of the bf16-rounded operands)
        xr = x.float().permute(0, 3, 1, 2)
        dyr = dy.float().permute(0, 3, 1, 2)
        ref = torch.nn.grad.conv2d_weight(xr, (k, c, 3, 3), dyr, stride=1, padding=1).permute(0, 2, 3, 1)
        err = float((dw - ref).norm() / ref.norm())
        out[(n, h, w, c, k)] = dw.cpu()
        print(f'{(n, h, w, c, k)}: rel err vs fp32 {err:.2e}', flush=True)
        assert err < 2e-3
    torch.save(out, args.save)
    if args.cmp:
        other = torch.load(args.cmp)
        bad = [key for key in out if not torch.equal(out[key], other[key])]
        print('bit-identical to', args.cmp, ':', 'ALL' if not bad else f'NO — differ: {bad}')
        sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()

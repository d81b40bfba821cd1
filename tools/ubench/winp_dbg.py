#!/usr/bin/env python
"""Localise an error of the window weight-gradient kernel: per (tap, 16-channel tile) relative error vs fp32."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torchok_amd import _C
lib = _C.load_library()
st = torch.cuda.current_stream().cuda_stream
for (n, h, w, c, k) in [(2, 30, 26, 96, 48), (1, 32, 32, 48, 48), (1, 8, 32, 48, 48), (3, 9, 33, 48, 48)]:
    d = _C.ConvDesc(n, h, w, c, k, 3, 3, h, w, 1, 1, 3)
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(n, h, w, c, device='cuda', generator=g).to(torch.bfloat16)
    dy = torch.randn(n, h, w, k, device='cuda', generator=g).to(torch.bfloat16)
    dw = torch.zeros(k, 3, 3, c, device='cuda')
    wsb = lib.tok_conv_wgrad_ws_bytes(ctypes.byref(d))
    ws = torch.zeros(max(wsb // 4, 16), device='cuda')
    assert lib.tok_conv_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), k, c, ws.data_ptr(), wsb, 0, st) == 0
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (k, c, 3, 3), dy.float().permute(0, 3, 1, 2), stride=1, padding=1).permute(0, 2, 3, 1)
    print((n, h, w, c, k), 'total', float((dw - ref).norm() / ref.norm()))
    for kr in range(3):
        for ks in range(3):
            row = []
            for ct in range(c // 16):
                a, b = dw[:, kr, ks, ct * 16:(ct + 1) * 16], ref[:, kr, ks, ct * 16:(ct + 1) * 16]
                row.append(f'{float((a - b).norm() / b.norm()):.1e}')
            print('  tap', kr, ks, ' '.join(row))
    # per output-channel tile
    print('  by out tile', [f'{float((dw[i*16:(i+1)*16]-ref[i*16:(i+1)*16]).norm()/ref[i*16:(i+1)*16].norm()):.1e}' for i in range(k // 16)])

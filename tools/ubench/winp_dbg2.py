#!/usr/bin/env python
"""Which reduction rows does the window weight-gradient kernel get wrong?  x = 1, dy = one-hot pixel: dW[:, tap, :] must be the
validity (0 / 1) of (pixel, tap)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torchok_amd import _C
lib = _C.load_library()
st = torch.cuda.current_stream().cuda_stream
n, h, w, c, k = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (1, 8, 32, 48, 48))]
d = _C.ConvDesc(n, h, w, c, k, 3, 3, h, w, 1, 1, 3)
x = (torch.arange(n * h * w, device='cuda').float() + 1).view(n, h, w, 1).expand(n, h, w, c).contiguous().to(torch.bfloat16)   # pixel index + 1 (exact in bf16 up to 256)
wsb = lib.tok_conv_wgrad_ws_bytes(ctypes.byref(d))
ws = torch.zeros(max(wsb // 4, 16), device='cuda')
M = n * h * w
bad = {}
for m in range(M):
    dy = torch.zeros(M, k, device='cuda', dtype=torch.bfloat16)
    dy[m] = 1
    dw = torch.zeros(k, 3, 3, c, device='cuda')
    assert lib.tok_conv_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), k, c, ws.data_ptr(), wsb, 0, st) == 0
    p, q = (m // w) % h, m % w
    for kr in range(3):
        for ks in range(3):
            want = float(m + (kr - 1) * w + (ks - 1) + 1) if 0 <= p + kr - 1 < h and 0 <= q + ks - 1 < w else 0.0
            got = dw[:, kr, ks, :]
            if not bool((got == want).all()):
                bad.setdefault((kr, ks), []).append((m, p, q, want, float(got.min()), float(got.max())))
for t, lst in sorted(bad.items()):
    print('tap', t, len(lst), 'bad pixels; first:', lst[:12])
print('done', M, 'pixels')

"""Isolated per-call times of the token GEMMs of one SwinV2-T block at batch 256, per stage, with the achieved byte and FLOP
rates (algorithmic bytes: read x + write y for forward, etc.).    python tools/ubench/swin_budget.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchok_amd import _C  # noqa: E402

lib = _C.load_library()
st = torch.cuda.current_stream().cuda_stream
BF = torch.bfloat16


def timeit(f, n=10):
    for _ in range(2):
        assert f() == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
for stage, (rows, c, blocks) in enumerate(((802816, 96, 2), (200704, 192, 2), (50176, 384, 6), (12544, 768, 2))):
    shapes = [('qkv', c, 3 * c), ('proj', c, c)]
    if c == 768:
        shapes += [('fc1', c, 4 * c), ('fc2', 4 * c, c)]
    for name, k_in, n_out in shapes:
        d = _C.ConvDesc(rows, 1, 1, k_in, n_out, 1, 1, 1, 1, 1, 0, 1)
        x = torch.randn(rows, k_in, device='cuda').to(BF)
        y = torch.randn(rows, n_out, device='cuda').to(BF)
        wf = (torch.randn(n_out, k_in, device='cuda') * 0.05).to(BF)
        wd = wf.t().contiguous()
        b = torch.randn(n_out, device='cuda')
        dw = torch.empty(n_out, k_in, device='cuda')
        db = torch.empty(n_out, device='cuda')
        wsb = int(lib.tok_conv_wgrad_bias_ws_bytes(d))
        ws = torch.empty(max(wsb // 4, 16), device='cuda')
        f = timeit(lambda: lib.tok_conv_fwd(d, x.data_ptr(), wf.data_ptr(), b.data_ptr(), y.data_ptr(), None, st))
        g = timeit(lambda: lib.tok_conv_dgrad(d, y.data_ptr(), wd.data_ptr(), x.data_ptr(), 0, st))
        w = timeit(lambda: lib.tok_conv_wgrad_bias(d, x.data_ptr(), y.data_ptr(), dw.data_ptr(), n_out, k_in, ws.data_ptr(), wsb, 0,
                                                   db.data_ptr(), 0, st))
        byt = rows * (k_in + n_out) * 2
        fl = 2.0 * rows * k_in * n_out
        print(f'stage {stage + 1} {name:5s} M={rows} K={k_in} N={n_out}: fwd {f:6.1f} us ({byt / f / 1e6:5.2f} TB/s, {fl / f / 1e6:5.0f} TF/s)  '
              f'dgrad {g:6.1f} us ({byt / g / 1e6:5.2f} TB/s)  wgrad {w:6.1f} us ({byt / w / 1e6:5.2f} TB/s, {fl / w / 1e6:5.0f} TF/s)', flush=True)
        tot['fwd'] += f * blocks
        tot['dgrad'] += g * blocks
        tot['wgrad'] += w * blocks
    # LayerNorm + residual forward / backward at this stage
    xs = torch.randn(rows, c, device='cuda').to(BF)
    del x, y
print('per step (12 blocks): ' + ', '.join(f'{k} {v / 1e3:.2f} ms' for k, v in tot.items()))

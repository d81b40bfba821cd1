"""Per-call time of the Mlp launches at the SwinV2-T B=256 stage shapes (isolated, HIP events):
tok_mlp_fwd (no save / pre only / pre + act), tok_mlp_bwd_dx (with / without d(pre) rows), tok_mlp_bwd_dw, and the unfused pair of
weight-gradient launches tok_mlp_bwd_dw replaces.    python tools/ubench/mlp_dw_time.py [lib]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchok_amd import _C  # noqa: E402

lib = _C.load_library()
BF = torch.bfloat16
P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
st = torch.cuda.current_stream().cuda_stream


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


for rows, c in ((802816, 96), (200704, 192), (50176, 384)):
    h = 4 * c
    x = torch.randn(rows, c, device='cuda').to(BF)
    dy = (torch.randn(rows, c, device='cuda') * 0.5).to(BF)
    w1 = (torch.randn(h, c, device='cuda') * c ** -0.5).to(BF)
    w2 = (torch.randn(c, h, device='cuda') * h ** -0.5).to(BF)
    w2d, w1d = w2.t().contiguous(), w1.t().contiguous()
    b1, b2 = torch.randn(h, device='cuda') * .1, torch.randn(c, device='cuda') * .1
    y, dx = torch.empty_like(x), torch.empty_like(x)
    pre, act, dpre = (torch.empty(rows, h, dtype=BF, device='cuda') for _ in range(3))
    ws_b = lib.tok_mlp_bwd_dw_ws_bytes(rows, c, h)
    ws = torch.empty(ws_b // 4, dtype=torch.float32, device='cuda')
    g = [torch.empty(h, c, device='cuda'), torch.empty(h, device='cuda'), torch.empty(c, h, device='cuda'), torch.empty(c, device='cuda')]
    t = {}
    t['fwd'] = timeit(lambda: lib.tok_mlp_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(y), None, None, rows, c, h, st))
    t['fwd+pre'] = timeit(lambda: lib.tok_mlp_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(y), P(pre), None, rows, c, h, st))
    t['fwd+pre+act'] = timeit(lambda: lib.tok_mlp_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(y), P(pre), P(act), rows, c, h, st))
    t['dx'] = timeit(lambda: lib.tok_mlp_bwd_dx(P(dy), P(w2d), P(pre), P(w1d), P(dx), 0, None, rows, c, h, st))
    t['dx+dpre'] = timeit(lambda: lib.tok_mlp_bwd_dx(P(dy), P(w2d), P(pre), P(w1d), P(dx), 0, P(dpre), rows, c, h, st))
    t['dw(recompute)'] = timeit(lambda: lib.tok_mlp_bwd_dw(P(x), P(dy), P(w1), P(b1), P(w2d), P(g[0]), 0, P(g[1]), 0, P(g[2]), 0,
                                                         P(g[3]), 0, P(ws), ws_b, rows, c, h, st))
    d1 = _C.ConvDesc(rows, 1, 1, c, h, 1, 1, 1, 1, 1, 0, 1)
    d2 = _C.ConvDesc(rows, 1, 1, h, c, 1, 1, 1, 1, 1, 0, 1)
    wb1, wb2 = int(lib.tok_conv_wgrad_bias_ws_bytes(d1)), int(lib.tok_conv_wgrad_bias_ws_bytes(d2))
    s1, s2 = torch.empty(max(wb1 // 4, 1), device='cuda'), torch.empty(max(wb2 // 4, 1), device='cuda')
    t['wgrad fc1 (x, dpre)'] = timeit(lambda: lib.tok_conv_wgrad_bias(d1, P(x), P(dpre), P(g[0]), h, c, P(s1), wb1, 0, P(g[1]), 0, st))
    t['wgrad fc2 (act, dy)'] = timeit(lambda: lib.tok_conv_wgrad_bias(d2, P(act), P(dy), P(g[2]), c, h, P(s2), wb2, 0, P(g[3]), 0, st))
    flop = 2.0 * rows * c * h
    print(f'rows {rows} c {c}: ' + '  '.join(f'{k} {v:.0f} us' for k, v in t.items()) +
          f'   | dw: {4 * flop / t["dw(recompute)"] / 1e6:.0f} TF/s over its 4 products', flush=True)
    del x, dy, pre, act, dpre, ws
    torch.cuda.empty_cache()

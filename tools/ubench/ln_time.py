"""Per-call time of tok_layernorm_fwd / _bwd at the SwinV2-T B=256 stage shapes (isolated):  [TOK_LIB=...] python tools/ubench/ln_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchok_amd import _C  # noqa: E402

lib = _C.load_library()
st = torch.cuda.current_stream().cuda_stream
BF = torch.bfloat16
P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731


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


tot_f = tot_b = 0.0
for rows, c, blocks in ((802816, 96, 2), (200704, 192, 2), (50176, 384, 6), (12544, 768, 2)):
    x = torch.randn(rows, c, device='cuda').to(BF)
    sh = torch.randn(rows, c, device='cuda').to(BF)
    g = torch.randn(rows, c, device='cuda').to(BF)
    out, dx = torch.empty_like(x), torch.empty_like(x)
    gamma, beta = torch.randn(c, device='cuda'), torch.randn(c, device='cuda')
    mean, rstd = torch.empty(rows, device='cuda'), torch.empty(rows, device='cuda')
    sc = torch.rand(256, device='cuda')
    nrows = lib.tok_layernorm_bwd_rows(rows, c)
    part = torch.empty(2 * nrows * c, device='cuda')
    f = timeit(lambda: lib.tok_layernorm_fwd(P(x), P(sh), P(sc), rows // 256, P(gamma), P(beta), P(out), P(mean), P(rstd), rows, c, c, 1e-5, st))
    b = timeit(lambda: lib.tok_layernorm_bwd(P(g), P(x), P(mean), P(rstd), P(gamma), P(sc), rows // 256, P(dx), 0, P(part), rows, c, c, st))
    byt = rows * c * 2 * 3
    print(f'rows {rows} c {c}: fwd {f:6.1f} us ({byt / f / 1e6:4.2f} TB/s)  bwd {b:6.1f} us ({byt / b / 1e6:4.2f} TB/s)  partial rows {nrows}', flush=True)
    tot_f += 2 * blocks * f
    tot_b += 2 * blocks * b
print(f'per step (two LayerNorms per block): fwd {tot_f / 1e3:.2f} ms, bwd {tot_b / 1e3:.2f} ms   [{os.environ.get("TOK_LIB", "in-tree library")}]')

"""Isolated timing of the BatchNorm finalize + apply pair against the folded launch (round 5), ResNet-50 B=256 shapes.
   python tools/ubench/bn_fused_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchok_amd import _C  # noqa: E402

SHAPES = [(802816, 64), (200704, 128), (200704, 512), (50176, 256), (50176, 1024), (12544, 512), (12544, 2048)]


def timeit(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    lib = _C.load_library()
    st = torch.cuda.current_stream().cuda_stream
    BF = torch.bfloat16
    P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    for m, c in SHAPES:
        y = torch.randn(m, c, device='cuda').to(BF)
        dout = torch.randn(m, c, device='cuda').to(BF)
        out = torch.empty_like(y)
        dy = torch.empty_like(y)
        mask = torch.empty(m, c // 8, dtype=torch.uint8, device='cuda')
        rows = 384 if c <= 128 else 192
        stats = torch.rand(2, rows, c, device='cuda') * m / rows
        gamma, beta = torch.ones(c, device='cuda'), torch.zeros(c, device='cuda')
        vec = torch.zeros(4, c, device='cuda')
        coef = torch.zeros(3, c, device='cuda')
        dg, db = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
        sync = torch.zeros(_C.PHASE_SLOT_BYTES // 4 if hasattr(_C, 'PHASE_SLOT_BYTES') else 65536, dtype=torch.int32, device='cuda')
        tot = [0]
        fin = (P(stats), rows, m, c, c, P(gamma), P(beta), None, None, None, 0.1, 1e-5, P(vec[2]), P(vec[3]), P(vec[0]), P(vec[1]))

        def two():
            lib.tok_bn_finalize(*fin, st)
            lib.tok_bn_act_fwd(P(y), P(vec[0]), P(vec[1]), None, 1, P(out), P(mask), m, c, st)

        def act_only():
            lib.tok_bn_act_fwd(P(y), P(vec[0]), P(vec[1]), None, 1, P(out), P(mask), m, c, st)

        def one():
            tot[0] += lib.tok_bn_fused_producers(c)
            lib.tok_bn_finalize_act_fwd(*fin, P(y), None, 1, P(out), P(mask), m, None, P(sync), tot[0], st)
        rows_b = lib.tok_bn_bwd_rows(m, c)
        part = torch.rand(2, rows_b, c, device='cuda')
        finb = (P(part), rows_b, m, c, c, P(gamma), P(vec[2]), P(vec[3]), P(dg), P(db), P(coef), 0, 0)
        app = (P(dout), P(y), P(mask), P(vec[0]), P(vec[1]))

        def two_b():
            lib.tok_bn_bwd_finalize(*finb, st)
            lib.tok_bn_bwd_apply(*app, P(coef), 1, P(dy), None, 0, m, c, st)

        def app_only():
            lib.tok_bn_bwd_apply(*app, P(coef), 1, P(dy), None, 0, m, c, st)

        def one_b():
            tot[0] += lib.tok_bn_fused_producers(c)
            lib.tok_bn_bwd_finalize_apply(*finb, *app, 1, P(dy), None, 0, P(sync), tot[0], st)
        mb = m * c * 2 / 1e6
        print(f'M={m} C={c} ({mb:.0f} MB/tensor): fwd  apply {timeit(act_only):6.1f}  fin+apply {timeit(two):6.1f}  folded {timeit(one):6.1f} us | '
              f'bwd  apply {timeit(app_only):6.1f}  fin+apply {timeit(two_b):6.1f}  folded {timeit(one_b):6.1f} us', flush=True)
        assert int(sync[2]) == 0


if __name__ == '__main__':
    main()

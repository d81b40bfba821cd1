"""Where the pointwise ring's time goes on a streaming layer: the kernel with parts switched off (TOK_PW_DBG bits: 1 no fragments /
MFMAs, 2 no output stores, 4 no DMA loads; results invalid, timing only).   python tools/ubench/pw_probe.py"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
SHAPES = [(802816, 64, 256), (802816, 256, 64), (802816, 64, 64), (200704, 128, 512), (200704, 512, 128)]


def run(tag):
    from torchok_amd import _C
    lib = _C.load_library()
    st = torch.cuda.current_stream().cuda_stream
    BF = torch.bfloat16
    P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731

    def timeit(f, n=20):
        for _ in range(3):
            assert f() == 0, lib.tok_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for m, k, n in SHAPES:
        g = torch.Generator(device='cuda').manual_seed(m + k + n)
        d = _C.ConvDesc(m // 64, 8, 8, k, n, 1, 1, 8, 8, 1, 0, 1)      # (an image map, not token rows: the rules differ)
        x = torch.randn(m, k, device='cuda', generator=g).to(BF)
        w = (torch.randn(n, k, device='cuda', generator=g) * k ** -0.5).to(BF)
        y = torch.empty(m, n, dtype=BF, device='cuda')
        rows = lib.tok_conv_fwd_stat_rows(d)
        stats = torch.zeros(2, rows, n, device='cuda')
        tf = timeit(lambda: lib.tok_conv_fwd(d, P(x), P(w), None, P(y), P(stats), st))
        tp = timeit(lambda: lib.tok_conv_fwd(d, P(x), P(w), None, P(y), None, st))
        print(f'[{tag}] M={m} C={k} N={n}: fwd+stats {tf:7.1f} us   fwd {tp:7.1f} us   ({(m * k + m * n) * 2 / 1e6:.0f} MB)', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run(sys.argv[1])
        sys.exit(0)
    names = {1: 'no mfma', 2: 'no stores', 4: 'no loads (out-of-range offsets)', 8: 'no epilogue', 16: 'no barrier', 32: 'no DMA instructions'}
    for v in (0, 8, 9, 8 + 32, 1 + 8 + 32, 1 + 8 + 16 + 32, 32, 16):
        tag = ' + '.join(n for b, n in names.items() if v & b) or 'all on'
        subprocess.run([sys.executable, __file__, tag], env=dict(os.environ, TOK_PW_DBG=str(v)), check=True)

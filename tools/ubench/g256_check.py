"""gemm256.hip against conv_igemm.hip on the layers it serves: results (forward + batch statistics, data gradient fresh / accumulated /
with BatchNorm-backward sums / ReLU-masked store) and per-call time.   python tools/ubench/g256_check.py        (runs itself twice: TOK_GEMM256=1 / 0)"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
SHAPES = [(50176, 1024, 256), (50176, 256, 1024), (50176, 512, 1024), (12544, 2048, 512), (12544, 512, 2048), (12544, 1024, 2048),
          (50176, 384, 1152), (50176, 384, 384), (12544, 768, 2304), (12544, 768, 768), (12544, 768, 3072), (12544, 3072, 768), (12544, 2304, 768), (5000, 512, 520), (786432, 720, 720),
          (200704, 712, 512)]


def run(tag):
    from torchok_amd import _C
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
    out = {}
    for m, k, n in SHAPES:
        g = torch.Generator(device='cuda').manual_seed(m + k + n)
        d = _C.ConvDesc(m, 1, 1, k, n, 1, 1, 1, 1, 1, 0, 1)
        x = torch.randn(m, k, device='cuda', generator=g).to(BF)
        w = (torch.randn(n, k, device='cuda', generator=g) * k ** -0.5).to(BF)
        dy = torch.randn(m, n, device='cuda', generator=g).to(BF)
        wd = w.t().contiguous()                      # dgrad pack [c = k][K = n]
        y = torch.empty(m, n, dtype=BF, device='cuda')
        rows = lib.tok_conv_fwd_stat_rows(d)
        stats = torch.zeros(2, rows, n, device='cuda')
        tf = timeit(lambda: lib.tok_conv_fwd(d, P(x), P(w), None, P(y), P(stats), st))
        dx = torch.empty(m, k, dtype=BF, device='cuda')
        td = timeit(lambda: lib.tok_conv_dgrad(d, P(dy), P(wd), P(dx), 0, st))
        base = torch.randn(m, k, device='cuda', generator=g).to(BF)
        dxa = base.clone()
        assert lib.tok_conv_dgrad(d, P(dy), P(wd), P(dxa), 1, st) == 0
        rows_d = lib.tok_conv_dgrad_stat_rows(d)
        part = torch.zeros(2, rows_d, k, device='cuda')
        bny = torch.randn(m, k, device='cuda', generator=g).to(BF)
        mask = torch.randint(0, 256, (m, k // 8), device='cuda', generator=g, dtype=torch.uint8)
        dxs = torch.empty(m, k, dtype=BF, device='cuda')
        assert lib.tok_conv_dgrad_bnstats(d, P(dy), P(wd), P(dxs), 0, P(bny), P(mask), P(part), st) == 0, lib.tok_last_error()
        bias = torch.randn(n, device='cuda', generator=g)
        ya, yact = torch.empty(m, n, dtype=BF, device='cuda'), torch.empty(m, n, dtype=BF, device='cuda')     # fc1 + GELU
        tfa = timeit(lambda: lib.tok_conv_fwd_act(d, P(x), P(w), P(bias), P(ya), P(yact), 1, st), 5)
        dxg = torch.empty(m, k, dtype=BF, device='cuda')                                                       # fc2's dgrad * GELU'
        tda = timeit(lambda: lib.tok_conv_dgrad_act(d, P(dy), P(wd), P(bny), 1, P(dxg), st), 5)
        partm = torch.zeros(2, rows_d, k, device='cuda')
        dxm = base.clone()                            # mask-store on top of an accumulated gradient (the residual unit's use)
        assert lib.tok_conv_dgrad_maskstore(d, P(dy), P(wd), P(dxm), 1, P(mask), P(partm), st) == 0, lib.tok_last_error()
        torch.cuda.synchronize()
        fl = 2.0 * m * k * n
        print(f'[{tag}] M={m} K={k} N={n}: fwd+stats {tf:6.1f} us ({fl / tf / 1e6:5.0f} TF/s)  dgrad {td:6.1f} us ({fl / td / 1e6:5.0f} TF/s)  fwd+gelu {tfa:6.1f} us  dgrad*dgelu {tda:6.1f} us  rows {rows}/{rows_d}', flush=True)
        out[(m, k, n)] = dict(y=y.float().cpu(), s=stats.sum(1).cpu(), dx=dx.float().cpu(), dxa=dxa.float().cpu(), dxs=dxs.float().cpu(),
                              p=part.sum(1).cpu(), dxm=dxm.float().cpu(), pm=partm.sum(1).cpu(), ya=ya.float().cpu(), yact=yact.float().cpu(), dxg=dxg.float().cpu(), ref_y=(x.float() @ w.float().t()).cpu() if m * n < 1.2e8 else None)
    torch.save(out, f'/tmp/g256_{tag}.pt')


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run(sys.argv[1])
        sys.exit(0)
    for tag, v in (('new', '2'), ('old', '0')):
        subprocess.run([sys.executable, __file__, tag], env=dict(os.environ, TOK_GEMM256=v, TOK_GEMM256_ACT='1'), check=True)    # (fused activation on the 256 tiles is off by default: switched on for the comparison)
    a, b = torch.load('/tmp/g256_new.pt'), torch.load('/tmp/g256_old.pt')

    def rel(u, v):
        return float((u.double() - v.double()).norm() / (v.double().norm() + 1e-30))
    for key in a:
        ra, rb = a[key], b[key]
        line = ' '.join(f'{nm} {rel(ra[nm], rb[nm]):.1e}' for nm in ('y', 's', 'dx', 'dxa', 'dxs', 'p', 'dxm', 'pm', 'ya', 'yact', 'dxg'))
        ref = f" y-vs-fp32 new {rel(ra['y'], ra['ref_y']):.1e} old {rel(rb['y'], rb['ref_y']):.1e}" if ra['ref_y'] is not None else ''
        print(f'M,K,N={key}: new vs old  {line}{ref}')

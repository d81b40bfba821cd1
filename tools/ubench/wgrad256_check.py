"""conv_wgrad_ring8_kernel<256, 256> against the 128 x 128 ring plan on the layers it serves: dW (+ bias column sums), per-call time.
   python tools/ubench/wgrad256_check.py        (runs itself twice: TOK_WGRAD_256=1 / 0)"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
# (rows, in features = Ktot, out features = k)
SHAPES = [(50176, 1024, 256), (50176, 256, 1024), (50176, 512, 1024), (12544, 2048, 512), (12544, 512, 2048), (12544, 1024, 2048),
          (50176, 384, 1152), (50176, 384, 1536), (50176, 1536, 384), (12544, 768, 2304), (12544, 768, 768), (12544, 768, 3072),
          (12544, 3072, 768), (786432, 720, 720), (9000, 520, 264)]


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
    for m, c, k in SHAPES:
        g = torch.Generator(device='cuda').manual_seed(m + c + k)
        d = _C.ConvDesc(m, 1, 1, c, k, 1, 1, 1, 1, 1, 0, 1)
        x = torch.randn(m, c, device='cuda', generator=g).to(BF)
        dy = torch.randn(m, k, device='cuda', generator=g).to(BF)
        dw = torch.empty(k, c, device='cuda')
        db = torch.empty(k, device='cuda')
        wsb = int(lib.tok_conv_wgrad_bias_ws_bytes(d))
        ws = torch.empty(max(wsb // 4, 16), device='cuda')
        t = timeit(lambda: lib.tok_conv_wgrad_bias(d, P(x), P(dy), P(dw), k, c, P(ws), wsb, 0, P(db), 0, st))
        base = torch.randn(k, c, device='cuda', generator=g)
        dwa = base.clone()
        assert lib.tok_conv_wgrad(d, P(x), P(dy), P(dwa), k, c, P(ws), wsb, 1, st) == 0, lib.tok_last_error()
        torch.cuda.synchronize()
        fl = 2.0 * m * c * k
        print(f'[{tag}] M={m} Ktot={c} k={k}: wgrad+bias {t:7.1f} us ({fl / t / 1e6:5.0f} TF/s)  workspace {wsb / 1e6:6.1f} MB', flush=True)
        ref = (dy.float().t() @ x.float()).cpu() if m <= 60000 else None
        out[(m, c, k)] = dict(dw=dw.cpu(), db=db.cpu(), dwa=(dwa - base).cpu(), ref=ref, refb=dy.float().sum(0).cpu())
    torch.save(out, f'/tmp/w256_{tag}.pt')


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run(sys.argv[1])
        sys.exit(0)
    for tag, v in (('new', '2'), ('old', '0')):
        subprocess.run([sys.executable, __file__, tag], env=dict(os.environ, TOK_WGRAD_256=v), check=True)
    a, b = torch.load('/tmp/w256_new.pt'), torch.load('/tmp/w256_old.pt')

    def rel(u, v):
        return float((u.double() - v.double()).norm() / (v.double().norm() + 1e-30))
    for key in a:
        ra, rb = a[key], b[key]
        extra = f" dW-vs-fp32 new {rel(ra['dw'], ra['ref']):.1e} old {rel(rb['dw'], rb['ref']):.1e}" if ra['ref'] is not None else ''
        print(f"M,Ktot,k={key}: new vs old  dw {rel(ra['dw'], rb['dw']):.1e} db {rel(ra['db'], rb['db']):.1e} dwa {rel(ra['dwa'], rb['dwa']):.1e}"
              f" db-vs-fp32 {rel(ra['db'], ra['refb']):.1e}{extra}")

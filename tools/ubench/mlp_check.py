"""Fused-MLP forward/backward against the unfused launches and a torch restatement; timings.  python mlp_check.py [rows c]..."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchok_amd import _C
lib = _C.load_library()
st = torch.cuda.current_stream().cuda_stream
BF = torch.bfloat16
P = lambda t: t.data_ptr() if t is not None else None


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(rows, c):
    h = 4 * c
    g = torch.Generator(device='cuda').manual_seed(rows + c)
    x = torch.randn(rows, c, device='cuda', generator=g).to(BF)
    w1 = (torch.randn(h, c, device='cuda', generator=g) * c ** -0.5).to(BF)
    w2 = (torch.randn(c, h, device='cuda', generator=g) * h ** -0.5).to(BF)
    b1 = torch.randn(h, device='cuda', generator=g) * 0.1
    b2 = torch.randn(c, device='cuda', generator=g) * 0.1
    y = torch.empty(rows, c, device='cuda', dtype=BF)
    assert lib.tok_mlp_serves(rows, c, h) == 1
    _C.check(lib.tok_mlp_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(y), None, None, rows, c, h, st), 'mlp_fwd')
    ys = torch.empty_like(y); pre_s = torch.empty(rows, h, device='cuda', dtype=BF); act_s = torch.empty_like(pre_s)
    _C.check(lib.tok_mlp_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(ys), P(pre_s), P(act_s), rows, c, h, st), 'mlp_fwd save')
    torch.cuda.synchronize()
    # unfused launches
    d1 = _C.ConvDesc(rows, 1, 1, c, h, 1, 1, 1, 1, 1, 0, 1)
    d2 = _C.ConvDesc(rows, 1, 1, h, c, 1, 1, 1, 1, 1, 0, 1)
    pre = torch.empty(rows, h, device='cuda', dtype=BF); act = torch.empty_like(pre); y2 = torch.empty_like(y)
    def unfused():
        _C.check(lib.tok_conv_fwd_act(ctypes.byref(d1), P(x), P(w1), P(b1), P(pre), P(act), 1, st), 'fwd_act')
        _C.check(lib.tok_conv_fwd(ctypes.byref(d2), P(act), P(w2), P(b2), P(y2), None, st), 'fwd')
    unfused(); torch.cuda.synchronize()
    n = min(rows, 4096)
    pr = (x[:n].float() @ w1.float().t() + b1).to(BF)
    hr = torch.nn.functional.gelu(pr.float()).to(BF)
    yr = (hr.float() @ w2.float().t() + b2)
    e_f = (y[:n].float() - yr).abs().max().item(); e_u = (y2[:n].float() - yr).abs().max().item()
    d_fu = (y.float() - y2.float()).abs().max().item(); neq = (y != y2).float().mean().item()
    save_ok = bool((ys == y).all() and (pre_s == pre).all() and (act_s == act).all())
    t_f = timeit(lambda: lib.tok_mlp_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(y), None, None, rows, c, h, st))
    t_s = timeit(lambda: lib.tok_mlp_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(ys), P(pre_s), P(act_s), rows, c, h, st))
    t_u = timeit(unfused)
    # backward to the input
    dy = torch.randn(rows, c, device='cuda', generator=g).to(BF)
    w2d = w2.t().contiguous(); w1d = w1.t().contiguous()          # dgrad packs: [H][C] and [C][H]
    base = torch.randn(rows, c, device='cuda', generator=g).to(BF)
    res = {}
    for accf in (0, 1):
        dpre_u = torch.empty(rows, h, device='cuda', dtype=BF); dx_u = base.clone()
        _C.check(lib.tok_conv_dgrad_act(ctypes.byref(d2), P(dy), P(w2d), P(pre), 1, P(dpre_u), st), 'dgrad_act')
        _C.check(lib.tok_conv_dgrad(ctypes.byref(d1), P(dpre_u), P(w1d), P(dx_u), accf, st), 'dgrad')
        dpre_f = torch.empty_like(dpre_u); dx_f = base.clone(); dx_n = base.clone()
        _C.check(lib.tok_mlp_bwd_dx(P(dy), P(w2d), P(pre), P(w1d), P(dx_f), accf, P(dpre_f), rows, c, h, st), 'mlp_bwd_dx')
        _C.check(lib.tok_mlp_bwd_dx(P(dy), P(w2d), P(pre), P(w1d), P(dx_n), accf, None, rows, c, h, st), 'mlp_bwd_dx')
        torch.cuda.synchronize()
        res[accf] = (bool((dx_f == dx_u).all()), bool((dpre_f == dpre_u).all()), bool((dx_n == dx_u).all()),
                     (dx_f.float() - dx_u.float()).abs().max().item())
    dpre_u = torch.empty(rows, h, device='cuda', dtype=BF); dx_u = torch.empty_like(y); dpre_f = torch.empty_like(dpre_u); dx_f = torch.empty_like(y)
    def unfused_b():
        lib.tok_conv_dgrad_act(ctypes.byref(d2), P(dy), P(w2d), P(pre), 1, P(dpre_u), st)
        lib.tok_conv_dgrad(ctypes.byref(d1), P(dpre_u), P(w1d), P(dx_u), 0, st)
    tb_u = timeit(unfused_b)
    tb_s = timeit(lambda: lib.tok_mlp_bwd_dx(P(dy), P(w2d), P(pre), P(w1d), P(dx_f), 0, P(dpre_f), rows, c, h, st))
    tb_n = timeit(lambda: lib.tok_mlp_bwd_dx(P(dy), P(w2d), P(pre), P(w1d), P(dx_f), 0, None, rows, c, h, st))
    print(f'   bwd_dx identical (dx, dpre, dx no-save | max diff) acc0 {res[0]} acc1 {res[1]} | fused+dpre {tb_s:.1f} us fused {tb_n:.1f} us '
          f'unfused {tb_u:.1f} us', flush=True)
    flops = 2 * 2 * rows * c * h
    print(f'rows {rows} c {c}: |fused-ref| {e_f:.4f} |unfused-ref| {e_u:.4f} |fused-unfused| {d_fu:.4f} (differ {neq:.2e}) '
          f'save-mode identical {save_ok} | fused {t_f:.1f} us ({flops / t_f * 1e-6:.0f} TF/s) fused+save {t_s:.1f} us unfused {t_u:.1f} us', flush=True)


if __name__ == '__main__':
    args = list(map(int, sys.argv[1:]))
    shapes = list(zip(args[0::2], args[1::2])) or [(802816, 96), (200704, 192), (50176, 384), (1000, 96), (77, 192)]
    for rows, c in shapes:
        run(rows, c)

"""Time tok_window_attn_fwd / _bwd at the SwinV2-T stage geometries (TOK_LIB selects the build)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchok_amd import _C
lib = _C.load_library()
BF = torch.bfloat16
P = lambda t: None if t is None else t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
tot_f = tot_b = 0.0
for (b, h, w, heads, ws, shift, mult) in ((256, 56, 56, 3, 7, 3, 2), (256, 28, 28, 6, 7, 3, 2), (256, 14, 14, 12, 7, 3, 6), (256, 7, 7, 24, 7, 0, 2)):
    c, n, nw = heads * 32, ws * ws, (h // ws) * (w // ws)
    qkv = torch.randn(b * h * w, 3 * c, device='cuda').to(BF); g = torch.randn(b * h * w, c, device='cuda').to(BF)
    ls = torch.full((heads,), 2.3, device='cuda'); bias = torch.randn(heads, n, n, device='cuda')
    mask = torch.zeros(nw, n, n, device='cuda') if shift else None
    out = torch.empty(b * h * w, c, dtype=BF, device='cuda'); lse = torch.empty(b * nw * heads * n, device='cuda')
    rows = lib.tok_window_attn_bwd_rows(b, h, w, heads, ws)
    dq = torch.empty(b * h * w, 3 * c, dtype=BF, device='cuda'); scr = torch.empty(rows, heads * n * n, device='cuda'); dsp = torch.empty(rows, heads, device='cuda')
    f = lambda: lib.tok_window_attn_fwd(P(qkv), b, h, w, c, heads, ws, shift, 3 * c, P(ls), P(bias), P(mask), P(out), P(lse), st)
    bw = lambda: lib.tok_window_attn_bwd(P(qkv), P(g), b, h, w, c, heads, ws, shift, 3 * c, P(ls), P(bias), P(mask), P(lse), P(dq), P(scr), P(dsp), st)
    res = []
    for fn in (f, bw):
        for _ in range(3): assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    tot_f += res[0] * mult; tot_b += res[1] * mult
    print(f'{h}x{w} heads {heads}: fwd {res[0]:.1f} us  bwd {res[1]:.1f} us', flush=True)
print(f'per SwinV2-T step: fwd {tot_f / 1e3:.3f} ms  bwd {tot_b / 1e3:.3f} ms')

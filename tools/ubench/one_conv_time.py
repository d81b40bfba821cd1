"""us per call of one conv shape:  python one_conv_time.py n h c k r stride fwd|dgrad|wgrad   (square maps; TOK_LIB selects the library)"""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchok_amd import _C
lib = _C.load_library()
st = torch.cuda.current_stream().cuda_stream
n, h, c, k, r, stride = map(int, sys.argv[1:7]); what = sys.argv[7]
pad = (r - 1) // 2; p = (h + 2 * pad - r) // stride + 1
d = _C.ConvDesc(n, h, h, c, k, r, r, p, p, stride, pad, r)
BF = torch.bfloat16
x = torch.randn(n, h, h, c, device='cuda').to(BF); y = torch.randn(n, p, p, k, device='cuda').to(BF)
wf = (torch.randn(k, r, r, c, device='cuda') * .05).to(BF); wd = (torch.randn(c, r, r, k, device='cuda') * .05).to(BF)
rows = lib.tok_conv_fwd_stat_rows(ctypes.byref(d)); stats = torch.empty(2, rows, k, device='cuda')
dw = torch.empty(k, r, r, c, device='cuda'); wsb = lib.tok_conv_wgrad_ws_bytes(ctypes.byref(d)); ws = torch.empty(max(wsb // 4, 16), device='cuda')
def run():
    if what == 'fwd': lib.tok_conv_fwd(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), None, y.data_ptr(), stats.data_ptr(), st)
    elif what == 'dgrad': lib.tok_conv_dgrad(ctypes.byref(d), y.data_ptr(), wd.data_ptr(), x.data_ptr(), 0, st)
    else: lib.tok_conv_wgrad(ctypes.byref(d), x.data_ptr(), y.data_ptr(), dw.data_ptr(), k, c, ws.data_ptr(), wsb, 0, st)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f'{e0.elapsed_time(e1) * 50:.1f} us')

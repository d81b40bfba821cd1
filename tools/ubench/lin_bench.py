import ctypes, sys, torch
sys.path.insert(0, '/root/repo')
from torchok_amd import _C
from tools.bench_conv import timeit
lib = _C.load_library()
st = torch.cuda.current_stream().cuda_stream
BF16 = torch.bfloat16
for (n, h, c, k) in [(256, 14, 384, 1536), (50176, 1, 384, 1536), (256, 14, 1536, 384), (50176, 1, 1536, 384), (256, 56, 96, 384), (802816, 1, 96, 384)]:
    d = _C.ConvDesc(n, h, h, c, k, 1, 1, h, h, 1, 0, 1)
    x = torch.randn(n, h, h, c, device='cuda').to(BF16)
    y = torch.randn(n, h, h, k, device='cuda').to(BF16)
    wf = (torch.randn(k, 1, 1, c, device='cuda') * 0.05).to(BF16)
    wd = (torch.randn(c, 1, 1, k, device='cuda') * 0.05).to(BF16)
    dw = torch.empty(k, c, device='cuda')
    wsb = lib.tok_conv_wgrad_ws_bytes(ctypes.byref(d))
    ws = torch.empty(max(wsb // 4, 16), device='cuda')
    f = timeit(lambda: lib.tok_conv_fwd(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), None, y.data_ptr(), None, st))
    g = timeit(lambda: lib.tok_conv_dgrad(ctypes.byref(d), y.data_ptr(), wd.data_ptr(), x.data_ptr(), 0, st))
    w = timeit(lambda: lib.tok_conv_wgrad(ctypes.byref(d), x.data_ptr(), y.data_ptr(), dw.data_ptr(), k, c, ws.data_ptr(), wsb, 0, st))
    print((n, h, c, k), 'fwd %.1f dgrad %.1f wgrad %.1f us' % (f, g, w))

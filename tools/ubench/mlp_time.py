"""Time tok_mlp_fwd out of a given library: python mlp_time.py lib rows c"""
import ctypes, sys, torch
lib = ctypes.CDLL(sys.argv[1]); rows, c = int(sys.argv[2]), int(sys.argv[3]); h = 4 * c
BF = torch.bfloat16; P = lambda t: ctypes.c_void_p(t.data_ptr())
x = torch.randn(rows, c, device='cuda').to(BF); w1 = (torch.randn(h, c, device='cuda') * c ** -0.5).to(BF)
w2 = (torch.randn(c, h, device='cuda') * h ** -0.5).to(BF); b1 = torch.randn(h, device='cuda') * .1; b2 = torch.randn(c, device='cuda') * .1
y = torch.empty_like(x); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
f = lambda: lib.tok_mlp_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(y), None, None, ctypes.c_int64(rows), c, h, st)
for _ in range(3): assert f() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print(sys.argv[1].split('/')[-1], rows, c, f'{e0.elapsed_time(e1) / 20 * 1e3:.1f} us', flush=True)

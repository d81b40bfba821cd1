"""Phase timing of the fused MLP (library built with -DTOK_MLP_TIMING): python mlp_timing.py lib rows c [bwd]"""
import ctypes, sys, torch
lib = ctypes.CDLL(sys.argv[1]); rows, c = int(sys.argv[2]), int(sys.argv[3]); h = 4 * c
BF = torch.bfloat16; P = lambda t: ctypes.c_void_p(t.data_ptr())
x = torch.randn(rows, c, device='cuda').to(BF); w1 = (torch.randn(h, c, device='cuda') * c ** -0.5).to(BF)
w2 = (torch.randn(c, h, device='cuda') * h ** -0.5).to(BF); b1 = torch.randn(h, device='cuda') * .1; b2 = torch.randn(c, device='cuda') * .1
y = torch.empty_like(x); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    rc = lib.tok_mlp_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(y), ctypes.c_int64(rows), c, h, st)
    assert rc == 0, rc
torch.cuda.synchronize()

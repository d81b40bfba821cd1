import ctypes, sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from torchok_amd import _C
lib = _C.load_library()
st = torch.cuda.current_stream().cuda_stream
n, h, w, c, k = 8, 128, 128, 48, 48
d = _C.ConvDesc(n, h, w, c, k, 3, 3, h, w, 1, 1, 3)
g = torch.Generator().manual_seed(0)
x = torch.randn(n, h, w, c, generator=g).to(torch.bfloat16)
wt = (torch.randn(k, 3, 3, c, generator=g) * 0.05).to(torch.bfloat16)
y = torch.zeros(n, h, w, k, dtype=torch.bfloat16, device='cuda')
assert lib.tok_conv_fwd(ctypes.byref(d), x.cuda().data_ptr(), wt.cuda().data_ptr(), None, y.data_ptr(), None, st) == 0
torch.cuda.synchronize()
ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wt.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
err = (y.float().cpu() - ref).abs()
print('rel', float(err.norm() / ref.norm()))
bad = err > 0.05
print('bad fraction', float(bad.float().mean()))
print('by row%8   ', [round(float(bad[:, r::8].float().mean()), 4) for r in range(8)])
print('by image row (first 10, last 4)', [round(float(bad[:, r].float().mean()), 3) for r in list(range(10)) + [124, 125, 126, 127]])
print('by col%32  ', [round(float(bad[:, :, cc::32].float().mean()), 3) for cc in range(32)])
print('by channel ', [round(float(bad[..., ch].float().mean()), 3) for ch in range(48)])
print('by image   ', [round(float(bad[i].float().mean()), 3) for i in range(n)])

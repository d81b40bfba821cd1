import ctypes, sys, torch
sys.path.insert(0, '/root/repo')
from torchok_amd import _C
lib = _C.load_library('/root/repo/torchok_amd/lib/libtok_timing.so')
st = torch.cuda.current_stream().cuda_stream
BF16 = torch.bfloat16
def run(n,h,c,k,r,stride,what):
    pad=(r-1)//2; p=(h+2*pad-r)//stride+1
    d=_C.ConvDesc(n,h,h,c,k,r,r,p,p,stride,pad,r)
    x=torch.randn(n,h,h,c,device='cuda').to(BF16); y=torch.randn(n,p,p,k,device='cuda').to(BF16)
    wf=(torch.randn(k,r,r,c,device='cuda')*0.05).to(BF16); wd=(torch.randn(c,r,r,k,device='cuda')*0.05).to(BF16)
    rows=lib.tok_conv_fwd_stat_rows(ctypes.byref(d)); stats=torch.empty(2,rows,k,device='cuda')
    for _ in range(3):
        if what=='fwd': lib.tok_conv_fwd(ctypes.byref(d),x.data_ptr(),wf.data_ptr(),None,y.data_ptr(),stats.data_ptr(),st)
        else: lib.tok_conv_dgrad(ctypes.byref(d),y.data_ptr(),wd.data_ptr(),x.data_ptr(),0,st)
    torch.cuda.synchronize()
for cfg in [(256,56,64,256,1,1,'fwd'),(256,56,256,64,1,1,'fwd'),(256,56,256,64,1,1,'dgrad'),(256,56,64,64,3,1,'fwd'),(256,28,128,512,1,1,'fwd'),(256,14,256,256,3,1,'fwd')]:
    print(cfg, file=sys.stderr); run(*cfg)

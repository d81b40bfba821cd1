#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/dbg
cat > /tmp/dbg.py <<'PY'
import sys, torch
sys.path.insert(0, '/root/repo')
from torchok_amd import _C
lib = _C.load_library()
st = torch.cuda.current_stream().cuda_stream
BF = torch.bfloat16
P = lambda t: t.data_ptr() if t is not None else None
which = sys.argv[1]
m, k, n = 12544, 768, 3072
g = torch.Generator(device='cuda').manual_seed(1)
d = _C.ConvDesc(m, 1, 1, k, n, 1, 1, 1, 1, 1, 0, 1)
x = torch.randn(m, k, device='cuda', generator=g).to(BF)
w = (torch.randn(n, k, device='cuda', generator=g) * k ** -0.5).to(BF)
dy = torch.randn(m, n, device='cuda', generator=g).to(BF)
wd = w.t().contiguous()
bias = torch.randn(n, device='cuda', generator=g)
pre = torch.randn(m, k, device='cuda', generator=g).to(BF)
if which == 'fwd':
    ya, yact = torch.empty(m, n, dtype=BF, device='cuda'), torch.empty(m, n, dtype=BF, device='cuda')
    print('rc', lib.tok_conv_fwd_act(d, P(x), P(w), P(bias), P(ya), P(yact), 1, st), lib.tok_last_error())
    torch.cuda.synchronize()
    ref = x.float() @ w.float().t() + bias
    print('fwd ok', float((ya.float() - ref).norm() / ref.norm()), float((yact.float() - torch.nn.functional.gelu(ya.float())).norm() / yact.float().norm()))
else:
    dx = torch.empty(m, k, dtype=BF, device='cuda')
    print('rc', lib.tok_conv_dgrad_act(d, P(dy), P(wd), P(pre), 1, P(dx), st), lib.tok_last_error())
    torch.cuda.synchronize()
    print('dgrad ok', float(dx.float().norm()))
PY
for w in fwd dgrad; do timeout 120 python /tmp/dbg.py $w > gpurun_out/dbg/$w.txt 2>&1; echo "$w rc=$?"; tail -3 gpurun_out/dbg/$w.txt; done
( timeout 900 python tools/ubench/g256_check.py ) > gpurun_out/dbg/g256_check.txt 2>&1; echo "g256_check rc=$?"; tail -5 gpurun_out/dbg/g256_check.txt

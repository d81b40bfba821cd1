"""Which Python lines of torchok_amd issue ATen copy / fill / cat kernels in one training step?
   python tools/ubench/copy_sites.py [swinv2_custom|resnet50|hrnet_w48]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
which = sys.argv[1] if len(sys.argv) > 1 else 'swinv2_custom'
if which.startswith('swin'):
    task = bench.build_swin_task(1000, 224).cuda().train(); shape = (64, 3, 224, 224); tgt = lambda g: torch.randint(0, 1000, (64,), generator=g, device='cuda')
elif which.startswith('hrnet'):
    task = bench.build_seg_task('hrnet_w48', 19, 512, 1024).cuda().train(); shape = (2, 3, 512, 1024); tgt = lambda g: torch.randint(0, 19, (2, 512, 1024), generator=g, device='cuda')
else:
    task = bench.build_task('resnet50', 1000).cuda().train(); shape = (64, 3, 224, 224); tgt = lambda g: torch.randint(0, 1000, (64,), generator=g, device='cuda')
opt = task.configure_optimizers()[0]['optimizer']
g = torch.Generator(device='cuda').manual_seed(1)
batch = {'image': torch.randn(*shape, generator=g, device='cuda').to(torch.bfloat16), 'target': tgt(g)}
def step(i):
    out = task.training_step(batch, i); opt.zero_grad(set_to_none=True); out['loss'].backward(); opt.step()
for i in range(3): step(i)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
cnt = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in ('empty', 'as_strided', 'view', 'detach', 'permute', 'alias', 'slice', 'select', 'reshape',
                                       'expand', 'unsqueeze', 'squeeze', 't.default', 'transpose', '_local_scalar')):
            st = traceback.extract_stack()
            site = next((f'{os.path.basename(f.filename)}:{f.lineno} {f.line}' for f in reversed(st)
                         if 'torchok_amd' in f.filename or f.filename.endswith('bench.py')), '?')
            cnt[(name, site[:150])] += 1
        return func(*args, **(kwargs or {}))
torch.autograd.set_multithreading_enabled(False)
with Spy():
    step(3)
torch.cuda.synchronize()
for (n, s_), c in cnt.most_common(45):
    print(f'{c:5d} {n:28s} {s_}')

"""Host time of the backward walk per node class (TOK_HOST_PROF=1):  python tools/ubench/host_prof.py <backbone> [steps]"""
import os, sys
os.environ['TOK_HOST_PROF'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torchok_amd.engine import core
bb = sys.argv[1] if len(sys.argv) > 1 else 'swinv2_custom'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
task = (bench.build_swin_task(1000, 224, bb) if bb in ('swinv2_custom', 'davit_t') else bench.build_task(bb, 1000)).cuda().train()
opt = task.configure_optimizers()[0]['optimizer']
g = torch.Generator(device='cuda').manual_seed(1)
batch = {'image': torch.randn(256, 3, 224, 224, generator=g, device='cuda').to(torch.bfloat16),
         'target': torch.randint(0, 1000, (256,), generator=g, device='cuda')}
for i in range(steps + 5):
    if i == 5:
        core._host_prof.clear()
    out = task.training_step(batch, i)
    opt.zero_grad(set_to_none=True)
    out['loss'].backward()
    opt.step()
torch.cuda.synchronize()
for k, n, t in core.host_prof_report():
    print(f'{k:28s} {n / steps:7.1f} calls/step  {t / steps * 1e3:8.3f} ms/step  {t / n * 1e6:8.1f} us/call')

"""Launch-thread time of every phase of train_step at a batch the GPU finishes instantly (pure host cost per phase):
   python tools/ubench/host_phases.py hrnet_w48|resnet50|swinv2_custom [batch] [--profile-opt]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402

bb = sys.argv[1] if len(sys.argv) > 1 else 'swinv2_custom'
B = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else 2
g = torch.Generator(device='cuda').manual_seed(1)
if bb.startswith('hrnet'):
    task = bench.build_seg_task(bb, 19, 512, 1024).cuda().train()
    batch = {'image': torch.randn(B, 3, 512, 1024, generator=g, device='cuda').to(torch.bfloat16),
             'target': torch.randint(0, 19, (B, 512, 1024), generator=g, device='cuda')}
elif bb in ('swinv2_custom', 'davit_t'):
    task = bench.build_swin_task(1000, 224, bb).cuda().train()
    batch = {'image': torch.randn(B, 3, 224, 224, generator=g, device='cuda').to(torch.bfloat16),
             'target': torch.randint(0, 1000, (B,), generator=g, device='cuda')}
else:
    task = bench.build_task(bb, 1000).cuda().train()
    batch = {'image': torch.randn(B, 3, 224, 224, generator=g, device='cuda').to(torch.bfloat16),
             'target': torch.randint(0, 1000, (B,), generator=g, device='cuda')}
opt = task.configure_optimizers()[0]['optimizer']
from torchok_amd.engine.step import train_step  # noqa: E402
for i in range(5):
    train_step(task, opt, batch, i)
torch.cuda.synchronize()
acc = {}
N = 20
pr = cProfile.Profile() if '--profile-opt' in sys.argv else None
for i in range(N):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    out = task.training_step(batch, i); t.append(time.perf_counter())
    opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    out['loss'].backward(); t.append(time.perf_counter())
    if pr:
        pr.enable()
    opt.step()
    if pr:
        pr.disable()
    t.append(time.perf_counter())
    task.on_train_batch_end(out, batch, i); t.append(time.perf_counter())
    for k, a, b in zip(('training_step', 'zero_grad', 'backward', 'optimizer.step', 'on_train_batch_end'), t[:-1], t[1:]):
        acc[k] = acc.get(k, 0.0) + (b - a)
print(f'{bb} B={B} host ms/step: ' + ', '.join(f'{k} {v / N * 1e3:.3f}' for k, v in acc.items()) +
      f' | total {sum(acc.values()) / N * 1e3:.2f}')
if pr:
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)

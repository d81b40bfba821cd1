"""cProfile of the launch thread for one workload at a batch the GPU finishes instantly:
   python tools/ubench/host_cprofile.py hrnet_w48|resnet50|swinv2_custom [batch]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
bb = sys.argv[1] if len(sys.argv) > 1 else 'hrnet_w48'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
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
from torchok_amd.engine.step import train_step
for i in range(5):
    train_step(task, opt, batch, i, batch_end_hook=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    train_step(task, opt, batch, i, batch_end_hook=False)
torch.cuda.synchronize()
print(f'{bb} B={B}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step (host-bound)')
torch.autograd.set_multithreading_enabled(False)      # the backward walk on this thread, so that the profile sees it
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    train_step(task, opt, batch, i, batch_end_hook=False)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)

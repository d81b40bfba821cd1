"""cProfile of the launch thread with the gradient reducer active on one rank (SwinV2-T, tiny batch)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
import torch, torch.distributed as dist
import bench
from torchok_amd.dist.ddp import GradientAllReducer
from torchok_amd.engine.step import train_step
dist.init_process_group('nccl', rank=0, world_size=1)
bb = sys.argv[1] if len(sys.argv) > 1 else 'swinv2_custom'
B = 2
g = torch.Generator(device='cuda').manual_seed(1)
task = (bench.build_swin_task(1000, 224, bb) if bb in ('swinv2_custom', 'davit_t') else bench.build_task(bb, 1000)).cuda().train()
batch = {'image': torch.randn(B, 3, 224, 224, generator=g, device='cuda').to(torch.bfloat16),
         'target': torch.randint(0, 1000, (B,), generator=g, device='cuda')}
opt = task.configure_optimizers()[0]['optimizer']
red = GradientAllReducer(opt, module=task, find_unused_parameters=bb != 'resnet50')
for i in range(5):
    train_step(task, opt, batch, i, reducer=red, batch_end_hook=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    train_step(task, opt, batch, i, reducer=red, batch_end_hook=False)
torch.cuda.synchronize()
print(f'{bb} B={B} with reducer: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step (host-bound)')
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile(); pr.enable()
for i in range(10):
    train_step(task, opt, batch, i, reducer=red, batch_end_hook=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
dist.destroy_process_group()

"""Unprofiled: how long the main stream takes from "backward fully enqueued" to "optimizer step done", and from there to the first
forward kernel of the next step having run (HIP events on the step stream; bench-size batch).  If the launch thread is ahead of the
GPU these are the kernels' own durations (join of the side stream + optimizer + repack); a larger number is launch-thread latency
the GPU sits idle through.    python tools/ubench/tail_gap.py swinv2_custom|resnet50|hrnet_w48"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402

bb = sys.argv[1] if len(sys.argv) > 1 else 'swinv2_custom'
g = torch.Generator(device='cuda').manual_seed(1)
if bb.startswith('hrnet'):
    B = 24
    task = bench.build_seg_task(bb, 19, 512, 1024).cuda().train()
    batch = {'image': torch.randn(B, 3, 512, 1024, generator=g, device='cuda').to(torch.bfloat16),
             'target': torch.randint(0, 19, (B, 512, 1024), generator=g, device='cuda')}
elif bb in ('swinv2_custom', 'davit_t'):
    B = 256
    task = bench.build_swin_task(1000, 224, bb).cuda().train()
    batch = {'image': torch.randn(B, 3, 224, 224, generator=g, device='cuda').to(torch.bfloat16),
             'target': torch.randint(0, 1000, (B,), generator=g, device='cuda')}
else:
    B = 256
    task = bench.build_task(bb, 1000).cuda().train()
    batch = {'image': torch.randn(B, 3, 224, 224, generator=g, device='cuda').to(torch.bfloat16),
             'target': torch.randint(0, 1000, (B,), generator=g, device='cuda')}
opt = task.configure_optimizers()[0]['optimizer']
N = 14
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(N)]
for i in range(N):
    ev[i][0].record()
    out = task.training_step(batch, i)
    opt.zero_grad(set_to_none=True)
    out['loss'].backward()
    ev[i][1].record()          # everything of the backward is enqueued in front of this
    opt.step()
    ev[i][2].record()
    task.on_train_batch_end(out, batch, i)
    ev[i][3].record()
torch.cuda.synchronize()
step = [ev[i][0].elapsed_time(ev[i + 1][0]) for i in range(4, N - 1)]
tail = [ev[i][1].elapsed_time(ev[i][2]) for i in range(4, N - 1)]
fb = [ev[i][0].elapsed_time(ev[i][1]) for i in range(4, N - 1)]
print(f'{bb} B={B}: step {statistics.median(step):.3f} ms = forward+backward {statistics.median(fb):.3f} + '
      f'[backward enqueued -> optimizer done] {statistics.median(tail):.3f} ms (min {min(tail):.3f}, max {max(tail):.3f})')

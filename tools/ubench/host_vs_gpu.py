"""How far ahead of the GPU does the Python launch thread run?  Per training step: host time to ENQUEUE the step (no sync),
split into forward / backward / optimizer, next to the GPU time of the step (HIP events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

task = bench.build_task('resnet50', 1000).cuda().train()
opt = task.configure_optimizers()[0]['optimizer']
g = torch.Generator(device='cuda').manual_seed(1)
image = torch.randn(256, 3, 224, 224, generator=g, device='cuda').to(torch.bfloat16)
target = torch.randint(0, 1000, (256,), generator=g, device='cuda')
batch = {'image': image, 'target': target}
def step(i, t):
    t0 = time.perf_counter()
    out = task.training_step(batch, i)
    t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out['loss'].backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    t.append((t1 - t0, t2 - t1, t3 - t2))
for i in range(10):
    step(i, [])
torch.cuda.synchronize()
ts = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
w0 = time.perf_counter()
e0.record()
for i in range(30):
    step(i, ts)
e1.record()
host_total = time.perf_counter() - w0
torch.cuda.synchronize()
f = sum(t[0] for t in ts) / len(ts) * 1e3
b = sum(t[1] for t in ts) / len(ts) * 1e3
o = sum(t[2] for t in ts) / len(ts) * 1e3
print(f'host enqueue per step: forward {f:.2f} ms, backward {b:.2f} ms, optimizer {o:.2f} ms, total {host_total / 30 * 1e3:.2f} ms; '
      f'GPU per step {e0.elapsed_time(e1) / 30:.2f} ms')
# the same with the GPU drained before every step: pure host cost of enqueueing one step
ts = []
for i in range(10):
    torch.cuda.synchronize()
    step(i, ts)
f = sum(t[0] for t in ts) / len(ts) * 1e3
b = sum(t[1] for t in ts) / len(ts) * 1e3
o = sum(t[2] for t in ts) / len(ts) * 1e3
print(f'host enqueue per step on an idle GPU: forward {f:.2f} ms, backward {b:.2f} ms, optimizer {o:.2f} ms')

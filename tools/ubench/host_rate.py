#!/usr/bin/env python
"""How far ahead of the GPU does the host run?  Enqueues K eager ResNet-50 training steps (bench.py's default workload)
and reports the host time spent enqueueing per step next to the GPU time per step:  python tools/ubench/host_rate.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    torch.manual_seed(1234)
    task = bench.build_task('resnet50', 1000).cuda().train()
    opt = task.configure_optimizers()[0]['optimizer']
    x = torch.randn(256, 3, 224, 224, device='cuda').to(torch.bfloat16)
    y = torch.randint(0, 1000, (256,), device='cuda')
    batch = {'image': x, 'target': y}

    def step(i):
        out = task.training_step(batch, i)
        opt.zero_grad(set_to_none=True)
        out['loss'].backward()
        opt.step()
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    k = 20
    t0 = time.perf_counter()
    fwd = 0.0
    for i in range(k):
        a = time.perf_counter()
        out = task.training_step(batch, i)
        fwd += time.perf_counter() - a
        opt.zero_grad(set_to_none=True)
        out['loss'].backward()
        opt.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'host enqueue {1e3 * (t1 - t0) / k:.2f} ms/step (forward part {1e3 * fwd / k:.2f}), '
          f'GPU drain after the last enqueue {1e3 * (t2 - t1):.2f} ms, wall {1e3 * (t2 - t0) / k:.2f} ms/step')


if __name__ == '__main__':
    main()

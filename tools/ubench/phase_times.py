#!/usr/bin/env python
"""Un-profiled GPU time of the phases of one training step on the MAIN stream (HIP events between the calls of
engine/step.py: train_step): forward | backward (autograd + the side-stream joins it ends with) | optimizer (+ repack).
    python tools/ubench/phase_times.py [bench.py workload args]
Answers what rocprofv3 timelines cannot: whether the idle stretch they show before the optimizer kernel exists without
the profiler."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--backbone', default='resnet50')
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--res', type=int, default=224)
    ap.add_argument('--width', type=int, default=0)
    ap.add_argument('--classes', type=int, default=1000)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=8)
    a = ap.parse_args()
    width = a.width or a.res
    seg = a.backbone.startswith('hrnet')
    swin = a.backbone in ('swinv2_custom', 'davit_t')
    task = (bench.build_seg_task(a.backbone, a.classes, a.res, width) if seg else
            bench.build_swin_task(a.classes, a.res, a.backbone) if swin else bench.build_task(a.backbone, a.classes)).cuda().train()
    opt = task.configure_optimizers()[0]['optimizer']
    g = torch.Generator(device='cuda').manual_seed(1234)
    image = torch.randn(a.batch, 3, a.res, width, generator=g, device='cuda').to(torch.bfloat16)
    target = torch.randint(0, a.classes, (a.batch, a.res, width) if seg else (a.batch,), generator=g, device='cuda')
    batch = {'image': image, 'target': target}
    n = a.warmup + a.steps
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n)]
    for i in range(n):
        ev[i][0].record()
        out = task.training_step(batch, i)
        opt.zero_grad(set_to_none=True)
        ev[i][1].record()
        out['loss'].backward()
        ev[i][2].record()
        opt.step()
        task.on_train_batch_end(out, batch, i)
        ev[i][3].record()
    torch.cuda.synchronize()
    rows = [[ev[i][k].elapsed_time(ev[i][k + 1]) for k in range(3)] + [ev[i][0].elapsed_time(ev[i][3])] for i in range(a.warmup, n)]
    m = [sum(r[k] for r in rows) / len(rows) for k in range(4)]
    print(f'{a.backbone} B={a.batch}: forward {m[0]:.3f} ms | backward {m[1]:.3f} ms | optimizer + repack {m[2]:.3f} ms | step {m[3]:.3f} ms '
          f'(main-stream HIP events, mean of {len(rows)} steps)')


if __name__ == '__main__':
    main()

"""Probe: SwinV2-T step time eager vs hipGraph replay (SGD stand-in for the optimizer) — is the eager step host-bound?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torchok_amd.engine.graph import GraphedTrainingStep
import torchok_amd as T
from torchok_amd.constructor.config import apply_schema

def task_sgd(res=224):
    t = bench.build_swin_task(1000, res)
    cfg = t._hparams
    cfg.optimization[0].optimizer.name = 'SGD'
    cfg.optimization[0].optimizer.params = {'lr': 0.01, 'momentum': 0.9}
    return T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for graphed in (False, True):
    torch.manual_seed(0)
    task = task_sgd().cuda().train()
    opt = task.configure_optimizers()[0]['optimizer']
    batch = {'image': torch.randn(B, 3, 224, 224, device='cuda').to(torch.bfloat16), 'target': torch.randint(0, 1000, (B,), device='cuda')}
    if graphed:
        step = GraphedTrainingStep(task, opt, batch, warmup=3)
        fn = lambda: step(batch)
    else:
        def fn():
            out = task.training_step(batch, 0); opt.zero_grad(set_to_none=True); out['loss'].backward(); opt.step()
        for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): fn()
    torch.cuda.synchronize()
    print('graphed' if graphed else 'eager  ', 'B', B, '%.2f ms/step' % ((time.perf_counter() - t0) / 8 * 1e3))

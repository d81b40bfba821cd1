"""YAML -> task -> HIP step on the GPU box.  The reference starts a run with
`python -m torchok -cp <dir> -cn <recipe>` (/root/reference/torchok/__main__.py:21-41: Hydra loads the YAML, the schema of
constructor/config_structure.py is applied, TASKS.get(cfg.task.name)(cfg, **cfg.task.params) builds the task,
Constructor.create_optimizer builds the optimizer).  Here the same chain — `load_config` (anchors, ${oc.env:}, ${a.b}),
schema, the TASKS / BACKBONES / POOLINGS / HEADS / LOSSES / OPTIMIZERS / SCHEDULERS / METRICS registries — runs on cuda:0 from
a build-authored recipe (tests/recipes/classification_resnet18_golden.yaml) and the step it produces is held against the
committed golden step of the reference's own ResNet-18 (tests/golden/resnet18_cls_step.npz, gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

import torchok_amd as T
from helpers import deterministic_state

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)
RECIPE = os.path.join(HERE, 'recipes', 'classification_resnet18_golden.yaml')


def _task_from_yaml(overrides=None):
    os.environ.setdefault('HOME', '/root')
    cfg = T.load_config(RECIPE, overrides=overrides)
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    return cfg, task


def test_recipe_resolves_like_a_reference_recipe():
    cfg, task = _task_from_yaml()
    assert cfg.task.params.inputs[0]['shape'] == [3, 96, 96]                       # anchors
    assert cfg.data['TRAIN'][0]['dataset']['params']['num_classes'] == 10          # ${a.b} interpolation
    assert cfg.logger['log_dir'].startswith(os.environ['HOME'])                    # ${oc.env:HOME}
    assert type(task.backbone).__name__ == 'ResNet' and type(task.head).__name__ == 'ClassificationHead'
    conf = task.configure_optimizers()[0]
    assert type(conf['optimizer']).__name__ == 'SGD' and type(conf['lr_scheduler']['scheduler']).__name__ == 'ExponentialLR'
    assert len(task.metrics_manager.phase2metrics['TRAIN']) == 1


def test_yaml_built_task_reproduces_the_reference_golden_step_on_hip():
    g = np.load(os.path.join(HERE, 'golden', 'resnet18_cls_step.npz'))
    assert str(g['backbone']) == 'resnet18' and int(g['num_classes']) == 10
    cfg, task = _task_from_yaml()
    sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, int(g['seed']))
    task.load_state_dict(sd, strict=False)
    task.cuda().train()
    conf = task.configure_optimizers()[0]
    opt, sched = conf['optimizer'], conf['lr_scheduler']['scheduler']
    x, y = torch.from_numpy(g['x'].astype(np.float32)).cuda(), torch.from_numpy(g['y']).cuda()
    batch = {'image': x, 'target': y, 'index': torch.arange(len(y), device='cuda')}
    from torchok_amd.engine.step import train_step
    out = train_step(task, opt, batch, 0)
    sched.step()
    torch.cuda.synchronize()
    # loss and logits of the YAML-built task == the reference's golden step (same gates as test_golden_gpu.py)
    assert abs(float(out['loss'].detach()) - float(g['loss'])) < 5e-3 * abs(float(g['loss']))
    pred = out['prediction'].detach().float().cpu().numpy() if 'prediction' in out else None
    if pred is not None:
        assert np.linalg.norm(pred - g['prediction']) < 0.025 * np.linalg.norm(g['prediction'])
    # gradients reached every parameter and have the reference's norms
    names = [n for n, _ in task.named_parameters()]
    assert names == [str(n) for n in g['param_names']]
    gn = np.array([float(p.grad.detach().double().norm()) for _, p in task.named_parameters()])
    dev = np.abs(gn / g['grad_norm'] - 1)
    assert np.median(dev) < 0.02 and np.percentile(dev, 90) < 0.08 and dev.max() < 0.25
    # the optimizer of the recipe (SGD 0.1 / 0.9 / 1e-4 = the golden step's) moved the parameters to the reference's norms
    pn = np.array([float(p.detach().double().norm()) for _, p in task.named_parameters()])
    assert np.abs(pn / g['post_step_norm'] - 1).max() < 2e-2
    assert abs(opt.param_groups[0]['lr'] - 0.09) < 1e-9                            # the scheduler section was honoured
    task.on_train_epoch_end()
    assert any(k.startswith('train/') for k in task.logged)


def test_yaml_recipe_through_the_fit_loop_on_hip():
    """The same recipe through torchok_amd.run.fit (what `python -m torchok_amd.run -cp tests/recipes -cn ...` does)."""
    from torchok_amd.run import fit
    cfg, _ = _task_from_yaml({'trainer.max_steps': 2})
    g = torch.Generator().manual_seed(0)
    batches = [{'image': torch.randn(8, 3, 96, 96, generator=g).cuda(), 'target': torch.randint(0, 10, (8,), generator=g).cuda()} for _ in range(2)]
    seen = []
    res = fit(cfg, batches=batches, max_steps=2, device='cuda:0', on_step=lambda i, out: seen.append(float(out['loss'])))
    assert res['steps'] == 2 and len(seen) == 2 and all(np.isfinite(seen))
    assert next(res['task'].parameters()).is_cuda

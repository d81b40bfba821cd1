"""HIP path vs the committed golden vectors (generated from the reference's own files by
tests/golden/gen_golden.py).  bf16 tolerances: logits/loss tight-ish, gradient norms loose (ReLU-mask /
argmax flips under bf16 rounding; see test_resnet_gpu.py for the autocast yardstick)."""
import os

import numpy as np
import pytest
import torch

import torchok_amd as T
from helpers import cls_config, deterministic_state

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('name', ['resnet18_cls_step', 'resnet50_cls_step'])
def test_hip_path_vs_reference_golden(name):
    g = np.load(os.path.join(GOLD, name + '.npz'))
    backbone, classes, seed = str(g['backbone']), int(g['num_classes']), int(g['seed'])
    cfg = cls_config(backbone, classes)
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, seed)
    task.load_state_dict(sd, strict=False)
    task.cuda().train()
    x, y = torch.from_numpy(g['x'].astype(np.float32)).cuda(), torch.from_numpy(g['y']).cuda()
    feats = task.backbone.forward_features(x)
    assert [list(f.shape) for f in feats] == g['feat_shapes'].tolist()
    for f, ss in zip(feats[1:], g['feat_sumsq'][1:]):
        assert abs(float((f.detach().double() ** 2).sum().item()) / float(ss) - 1) < 3e-2
    out = task.training_step({'image': x, 'target': y}, 0)
    fw = task.forward_with_gt({'image': x, 'target': y})
    pred = fw['prediction'].detach().float().cpu().numpy()
    assert np.linalg.norm(pred - g['prediction']) < 0.05 * np.linalg.norm(g['prediction'])
    assert abs(float(out['loss'].detach().item()) - float(g['loss'])) < 0.05 * abs(float(g['loss']))
    out['loss'].backward()
    names = [str(n) for n in g['param_names']]
    assert names == [n for n, _ in task.named_parameters()]
    gn = np.array([float(p.grad.detach().double().norm().item()) for _, p in task.named_parameters()])
    assert np.median(np.abs(gn / g['grad_norm'] - 1)) < 0.1
    fcb = task.head.fc.bias.grad.detach().float().cpu().numpy()
    assert np.abs(fcb - g['grad__head.fc.bias']).max() < 0.05 * np.abs(g['grad__head.fc.bias']).max() + 1e-3
